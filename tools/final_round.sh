#!/bin/bash
# Everything the round's committed evidence comes from, in one gpurun call:  gpurun --timeout 2400 -- 'bash tools/final_round.sh r02c'
TAG=${1:-r02c}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
bash "$R/tools/profile_round.sh" "$TAG" > /dev/null 2>&1
bash "$R/tools/prog_profile.sh" "$TAG" 1024 > /dev/null 2>&1
( cd "$R" && timeout 600 python3 tools/prog_batch_probe.py 16 256 1024 2048 8192 > "$O/prog_sweep_$TAG.txt" 2>&1 )
( cd "$R" && timeout 600 python3 bench.py --total-images 1250 --no-cpu-baseline --no-progressive > "$O/config4_1gpu_share_$TAG.json" 2> "$O/config4_1gpu_share_$TAG.err" )
( cd "$R" && timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --total-images 2500 --no-cpu-baseline --no-progressive > "$O/config4_2rank_sharegpu_$TAG.json" 2> "$O/config4_2rank_sharegpu_$TAG.err" )
bash "$R/tools/layout_sweep.sh" > /dev/null 2>&1
ls "$O" | tail -20
