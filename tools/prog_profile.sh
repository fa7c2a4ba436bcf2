#!/bin/bash
# Progressive (config 5) kernel trace + SQ counters of the scan kernels:  gpurun -- 'bash tools/prog_profile.sh r02p 1024'
TAG=${1:-r02p}; N=${2:-1024}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prog_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/tools/prog_batch_probe.py" $N > "$O/stats.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA --output-format csv -d "$O/pmc_sq" -- python3 "$R/tools/prog_batch_probe.py" $N > "$O/pmc_sq.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY --output-format csv -d "$O/pmc_sq2" -- python3 "$R/tools/prog_batch_probe.py" $N > "$O/pmc_sq2.log" 2>&1
cat "$O/stats.log" | tail -3
