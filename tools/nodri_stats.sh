#!/bin/bash
# per-kernel totals of one no-DRI batch (files without restart markers: the synchronisation form):  bash tools/nodri_stats.sh [batch] [lib] [NAME=VALUE,...]
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-256}
LIB=${2:-$R/pyjpegdecoder_amd/libmijpeg.so}
EXP=${3:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_nd
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nd -- python3 $R/tools/stage_probe.py --lib $LIB --ri 0 --batch $N --warm 2 --iters 4 "$EXP" > /tmp/prof_nd.log 2>&1
grep stage0 /tmp/prof_nd.log
python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/prof_nd/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    n=[int(r['Calls']) for r in rows if 'reconstruct' in r['Name']][0]
    for r in rows:
        print('  %-44s calls/exec %5.1f  per exec %8.1f us  avg %8.1f us' % (r['Name'][:44], int(r['Calls'])/n, float(r['TotalDurationNs'])/1e3/n, float(r['AverageNs'])/1e3))
PY
