#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01c'
# then, back in the container:  python tools/profile_summary.py r01c
# One rocprofv3 pass per counter set (never combined with trace domains other than the kernel trace).
TAG=${1:-r01x}
LAYOUT=${2:-xmajor}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
timeout 1200 python3 "$R/bench.py" --layout "$LAYOUT" > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
# the kernel trace is taken over a bench.py process that prints its own line: the rocprof average of the dominant kernel and the
# line's ms_per_step / HIP-event launch time / shader clock then come from ONE process on ONE box (round 5's did not: 6.10 ms in
# the committed stats against a 5.74 ms step in the driver's line).  A second of warm-up first (200 steps), then 30 timed steps.
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- \
    python3 "$R/bench.py" --steps 30 --warmup 200 --no-cpu-baseline --layout "$LAYOUT" > "$O/stats_bench_line.json" 2> "$O/stats.log"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
    n=$(echo "$set" | cut -c1-14 | tr " " "_")
    timeout 400 rocprofv3 --pmc $set --output-format csv -d "$O/pmc_$n" -- \
        python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --layout "$LAYOUT" > "$O/pmc_$n.log" 2>&1
done
ls "$O"
