#!/bin/bash
# Round 6, the review's item 1 (end-of-block-truncated coefficient store): upper bounds before building it.  Variant builds
#   libmijpeg_xld.so    make XFLAGS=-DMJ_X_SPARSE_LD    the strip worker does not fetch rows 4..7 of ANY chroma block
#   libmijpeg_xst.so    make XFLAGS=-DMJ_X_SPARSE_ST    the lane walk does not store them
#   libmijpeg_xldst.so  both
# (wrong pixels: the real thing would skip 95.8 % of the benchmark's chroma blocks, the probes skip all of them, at no cost in
# instructions) against the product build, same box, interleaved: whole steps (fused), the two launches, and the HBM counters
# of the fused launch.   gpurun --timeout 1500 -- 'bash tools/sparse_probe.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/sparse_probe
mkdir -p "$O"
cd "$R"
for rep in 1 2; do
  for l in libmijpeg libmijpeg_xld libmijpeg_xst libmijpeg_xldst; do
    echo "== $l (rep $rep)"
    timeout 300 python3 tools/fused_probe.py --lib pyjpegdecoder_amd/$l.so --distinct 64 "" 2>&1 | grep -v "^   images"
  done
done > "$O/steps.txt" 2>&1
for l in libmijpeg libmijpeg_xldst; do
  for set in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    n=$(echo "$set" | cut -c1-14 | tr " " "_")
    PMC_KERNEL=k_ bash tools/pmc_one.sh "sp_${l}_$n" "$set" --lib "$R/pyjpegdecoder_amd/$l.so" --distinct 64
  done
done > "$O/pmc.txt" 2>&1
cat "$O/steps.txt" "$O/pmc.txt"
# where the fused launch's time goes now (diagnostic build: per-workgroup stamps)
MJ_NO_GRAPH=1 MJ_DEBUG_FUSED=1 timeout 300 python3 tools/fused_probe.py --lib pyjpegdecoder_amd/libmijpeg_diag.so --distinct 64 --reps 3 "" > "$O/diag_fused.txt" 2>&1
tail -12 "$O/diag_fused.txt"
