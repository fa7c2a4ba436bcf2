#!/bin/bash
# Stage times of every sampling layout, both pixel layouts (1024 x 1080p): at DRI 120 (what the sweep has always used: one MCU row
# for 4:2:0 / 4:2:2, half a row for 4:4:4 / 4:4:0 / greyscale, two rows for 4:1:1) and at one MCU row per restart interval; 4:2:0
# also at two rows (DRI 240: x-major output fuses it since round 6, profiles/r06_xmajor_intervals.txt).
#   gpurun -- 'bash tools/layout_sweep.sh'      -> gpurun_out/layout_sweep.txt   (step = what mj_plan_execute does: one fused
#   launch where the plan allows it; stage0+1 / stage2 = the same plan's stages launched separately)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/layout_sweep.txt
: > "$O"
for ss in 420 422 440 444 411 grey; do
  case $ss in 420) intervals="120 240";; 422) intervals="120";; 411) intervals="120 60";; *) intervals="120 240";; esac
  for ri in $intervals; do
    for lay in xmajor rowmajor; do
      echo "== $ss $lay DRI=$ri" >> "$O"
      timeout 300 python3 "$R/tools/stage_probe.py" --subsampling $ss --layout $lay --batch 1024 --iters 10 --ri $ri >> "$O" 2>&1
    done
  done
done
cat "$O"
