#!/bin/bash
# One rocprofv3 --pmc pass over tools/stage_probe.py:  bash tools/pmc_one.sh <tag> "<counters>" [stage_probe args]
TAG=$1; SET=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $SET --output-format csv -d "$O/set1" -- python3 "$R/tools/stage_probe.py" --warm 1 --iters 1 "$@" > "$O/set1.log" 2>&1
python3 "$R/tools/pmc_summary.py" "$O" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if '${PMC_KERNEL:-reconstruct}' in k: print('$TAG', k[:40], {c:int(x) for c,x in v.items()})
"
