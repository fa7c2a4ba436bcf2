import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pathlib import Path
from pyjpegdecoder_amd import _binding as B
B.LIB_PATH = Path(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")) / "pyjpegdecoder_amd" / "libmijpeg_x1.so"
from pyjpegdecoder_amd.batch import prepare_batch
from tools import synth
from oracle import oracle
n = 16
raws = [synth.synth_jpeg(i, 1920, 1080, 85, "420", 120) for i in range(n)]
tot = 0
for r in raws[:2]:
    c = oracle.decode(r)["coef"].astype(np.int32)
    for b in range(c.shape[0]):
        pos = np.nonzero(c[b, 1:])[0] + 1
        prev = 0; k = 0
        for q in pos:
            k += (q - prev - 1) // 16 + 1; prev = q
        if prev != 63: k += 1
        tot += k
print("expected AC symbols per segment (host count, 2 images):", tot / (2 * 68))
os.environ["MJ_HUFFMAN"] = "lanes"
prep = prepare_batch(raws, B.MJ_LAYOUT_XMAJOR, 0)
ctx = B.Context(0)
dev = torch.device("cuda", 0)
d_blob = torch.from_numpy(prep.blob).to(dev)
plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
plan.execute(torch.cuda.current_stream().cuda_stream, d_rgb.data_ptr())
torch.cuda.synchronize()
per = 1920 * 1080 * 3
got = d_rgb[:per].cpu().numpy().reshape(1920, 1080, 3)
ref = oracle.decode(raws[0])["rgb"]
print("parity image 0:", np.array_equal(got, ref), "status", plan.read(rgb=False)["status"][:4])
