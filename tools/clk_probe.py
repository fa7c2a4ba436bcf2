import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["MJ_DEBUG_STAGE2"] = "4"
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
from tools import synth
blob, offs = synth.synth_batch(32, 1, 1920, 1080, 85, "420", 120)
raws = [blob[int(offs[i]):int(offs[i+1])].tobytes() for i in range(32)]
files = [raws[i % 32] for i in range(512)]
prep = prepare_batch(files)
ctx = B.Context(0)
plan = B.Plan(ctx, prep.to_c(), {"prep": prep, "n_images": 512})
plan.execute_stage1(); plan.sync()
for it in range(3):
    plan.execute_stage2(); plan.sync()
    out = plan.read(rgb=True)["rgb"]
    v = out[:16].view(np.uint64)
    print("shader cycles", v[0], "realtime ticks(100MHz)", v[1], "=> clock GHz", v[0] / (v[1] * 10.0))
