#!/bin/bash
# rocprofv3 --pmc passes over tools/prog_batch_probe.py (config 5):  bash tools/pmc_prog.sh <tag> <n_files> "<counters>" ["<counters>" ...]
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d "$O/set$i" -- python3 "$R/tools/prog_batch_probe.py" $N > "$O/set$i.log" 2>&1
done
python3 "$R/tools/pmc_summary.py" "$O" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'progressive_fast' in k: print('$TAG', k[:48], {c:int(x) for c,x in v.items()})
"
