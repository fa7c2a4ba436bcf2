#!/bin/bash
# Stage times of every sampling layout, both pixel layouts (1024 x 1080p, DRI 120): gpurun -- 'bash tools/layout_sweep.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/layout_sweep.txt
: > "$O"
for ss in 420 422 440 444 411 grey; do
  for lay in xmajor rowmajor; do
    echo "== $ss $lay" >> "$O"
    timeout 300 python3 "$R/tools/stage_probe.py" --subsampling $ss --layout $lay --batch 1024 --iters 10 >> "$O" 2>&1
  done
done
cat "$O"
