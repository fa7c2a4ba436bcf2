#!/bin/bash
# PMC passes over tools/stage_probe.py on the GPU box (one rocprofv3 run per counter set, kernel trace only):
#   gpurun --timeout 900 -- 'bash tools/pmc_probe.sh <tag> [stage_probe args...]'
# then: python tools/pmc_summary.py gpurun_out/pmc_<tag>
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d "$O/set$i" -- \
        python3 "$R/tools/stage_probe.py" --warm 2 --iters 2 "$@" > "$O/set$i.log" 2>&1
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_SALU
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_TAG_STALL_sum TCC_IB_STALL_sum
SETS
ls "$O"
