"""How fast are files WITHOUT restart markers (one segment per image)?  usage: python tools/nodri_probe.py [n_images [modes]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import synth
from pyjpegdecoder_amd import BatchDecoder, _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blob, offs = synth.synth_batch(16, 0, 1920, 1080, 85, "420", 0)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(16)]
files = [raws[i % 16] for i in range(n)]
dec = BatchDecoder(0)
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ("wave", "lanes", "sync")
for mode in modes:
    B.set_option("MJ_HUFFMAN", mode)
    prep = prepare_batch(files)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": n})
    plan.execute(); plan.sync()
    s1, s2 = plan.time_stages(5)
    print(f"{mode}: {n} x 1080p without DRI: stage1 {s1:.1f} ms, stage2 {s2:.1f} ms -> {n * 2.0736 / ((s1 + s2) * 1e-3):.0f} MP/s")
    plan.close()
