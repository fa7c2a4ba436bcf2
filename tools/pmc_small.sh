#!/bin/bash
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d "$O/set$i" -- python3 "$R/tools/stage_probe.py" --warm 1 --iters 1 "$@" > "$O/set$i.log" 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES
TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
SETS
