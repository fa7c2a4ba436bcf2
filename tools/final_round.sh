#!/bin/bash
# Everything the round's committed evidence comes from, in one gpurun call:  gpurun --timeout 2700 -- 'bash tools/final_round.sh r06e'
TAG=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
( time bash "$R/tools/profile_round.sh" "$TAG" > /dev/null 2>&1 ) 2> "$O/profile_round_time_$TAG.txt"
mkdir -p "$O/prog_$TAG"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prog_$TAG/stats" -- python3 "$R/tools/prog_batch_probe.py" 1024 > "$O/prog_$TAG/stats.log" 2>&1 )
( cd "$R" && timeout 600 python3 bench.py --total-images 1250 --no-cpu-baseline --no-progressive > "$O/config4_1gpu_share_$TAG.json" 2> "$O/config4_1gpu_share_$TAG.err" )
( cd "$R" && timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --total-images 2500 --no-cpu-baseline --no-progressive > "$O/config4_2rank_sharegpu_$TAG.json" 2> "$O/config4_2rank_sharegpu_$TAG.err" )
bash "$R/tools/layout_sweep.sh" > /dev/null 2>&1
( cd "$R" && timeout 300 python3 tools/stage_probe.py --mixed --distinct 256 "" MJ_SEG_ORDER=blob MJ_SEG_ORDER=binned MJ_SEG_ORDER=striped MJ_HUFFMAN=lanes11 > "$O/mixed_orders_$TAG.txt" 2>&1 )
( cd "$R" && timeout 300 python3 tools/generic_layout_probe.py > "$O/generic_layouts_$TAG.txt" 2>&1 )
( cd "$R" && timeout 300 python3 tools/prog_batch_probe.py 16 256 512 768 1024 2048 > "$O/prog_sweep_$TAG.txt" 2>&1 )
( cd "$R" && timeout 300 python3 tools/e2e_probe.py > "$O/e2e_$TAG.txt" 2>&1 )
( cd "$R" && timeout 600 python3 bench.py --gpus 2 --share-gpu --steps 20 --warmup 3 --no-cpu-baseline --no-progressive > "$O/bench_gpus2_sharegpu_$TAG.json" 2> "$O/bench_gpus2_sharegpu_$TAG.err" )
( cd "$R" && bash tools/nodri_stats.sh 256 > "$O/nodri_stats_$TAG.txt" 2>&1 )
( cd "$R" && timeout 300 python3 tools/stage_probe.py --ri 0 --batch 1 --distinct 1 "" > "$O/nodri_sizes_$TAG.txt" 2>&1; for n in 16 64 256 1024; do timeout 300 python3 tools/stage_probe.py --ri 0 --batch $n --distinct 16 "" MJ_SYNC_COUNT=classic 2>&1 | tail -2 | sed "s/^/$n files: /" >> "$O/nodri_sizes_$TAG.txt"; done )
( cd "$R" && timeout 200 python3 tools/single_file_probe.py 120 > "$O/single_file_$TAG.txt" 2>&1 )
( cd "$R" && timeout 900 python3 tools/stress_fused.py 120 61 > "$O/stress_fused_$TAG.txt" 2>&1 )
( cd "$R" && timeout 900 python3 tools/stress_parity.py 600 62 > "$O/stress_parity_$TAG.txt" 2>&1 )
( cd "$R" && timeout 900 python3 tools/stress_progressive.py > "$O/stress_progressive_$TAG.txt" 2>&1 )
( cd "$R" && timeout 900 python3 tools/stress_crafted.py 600 63 > "$O/stress_crafted_$TAG.txt" 2>&1 )
( cd "$R" && { [ -x tools/step_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm tools/step_probe.hip -o tools/step_probe.bin; } && timeout 300 ./tools/step_probe.bin > "$O/step_probe_$TAG.txt" 2>&1 )
ls "$O" | tail -20
