"""Stage-1 time of the three forms for a batch: python tools/form_probe.py n_images restart_interval"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from pyjpegdecoder_amd import BatchDecoder, _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
n, ri = int(sys.argv[1]), int(sys.argv[2])
blob, offs = synth.synth_batch(min(n, 16), 0, 1920, 1080, 85, "420", ri)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(min(n, 16))]
files = [raws[i % len(raws)] for i in range(n)]
dec = BatchDecoder(0)
for mode in ("wave", "lanes", "sync", None):
    if mode: B.set_option("MJ_HUFFMAN", mode)
    else: B.set_option("MJ_HUFFMAN", None)
    prep = prepare_batch(files)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": n})
    plan.execute(); plan.sync()
    s1, s2 = plan.time_stages(3)
    print(f"{n} x 1080p RI={ri} {mode or 'auto'}: stage1 {s1:.2f} ms, stage2 {s2:.2f} ms")
    plan.close()
