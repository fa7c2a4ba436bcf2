#!/bin/bash
# Same-box A/B of stage 2: every library given (paths relative to the repo) through tools/stage_probe.py, twice, interleaved.
#   gpurun -- 'bash tools/ab_stage2.sh pyjpegdecoder_amd/libmijpeg_prev.so pyjpegdecoder_amd/libmijpeg.so [-- stage_probe args]'
LIBS=(); ARGS=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; ARGS=("$@"); break; fi; LIBS+=("$1"); shift; done
for rep in 1 2; do
  for l in "${LIBS[@]}"; do
    echo -n "$(basename $l)  "; python tools/stage_probe.py --lib "$l" "${ARGS[@]}" "" 2>&1 | grep stage2
  done
done
