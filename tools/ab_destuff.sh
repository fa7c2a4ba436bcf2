#!/bin/bash
# per-kernel average durations of stage 0 and stage 1 for two builds:  bash tools/ab_destuff.sh [other.so]   (gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
OTHER=${1:-$R/pyjpegdecoder_amd/libmijpeg_base.so}
cd /tmp && export TMPDIR=/tmp
for LIB in $OTHER $R/pyjpegdecoder_amd/libmijpeg.so; do
  rm -rf /tmp/prof_ab
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python3 $R/tools/stage_probe.py --lib $LIB "" > /tmp/prof_ab.log 2>&1
  echo "== $LIB"
  python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/prof_ab/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('destuff','lanes13','reconstruct')): print('  %-40s calls %4s avg %9.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
