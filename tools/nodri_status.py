import sys, numpy as np, torch
sys.path.insert(0, '.')
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
from tools import synth
N, ND = 256, 64
blob, offs = synth.synth_batch(ND, 0, 1920, 1080, 85, "420", 0)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(ND)]
files = [raws[i % ND] for i in range(N)]
dev = torch.device("cuda", 0)
ctx = B.Context(0)
for rounds in (None, "32", "64", "8"):
    B.set_option("MJ_SYNC_ROUNDS", rounds)
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": N})
    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
    plan.execute(0, d_rgb.data_ptr()); plan.sync()
    st = plan.read(rgb=False)["status"]
    s1, s2 = plan.time_stages(5, d_rgb.data_ptr())
    print("rounds", rounds, "form", plan.stage1_form() & 15, "nonzero statuses", int((st != 0).sum()), "values", np.unique(st), "stage0+1 %.3f ms" % s1, flush=True)
    plan.close()
