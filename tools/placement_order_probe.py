"""Output buffer allocated BEFORE or AFTER the plan (whose coefficient store and stream buffer are hipMalloc'ed at creation): the step
of the config-3 batch, one fresh process per order (tools/placement_probe.py found the relation, this pins the rule).
    python tools/placement_order_probe.py before|after|spacer [gib]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
from tools import synth

order = sys.argv[1] if len(sys.argv) > 1 else "after"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
dev = torch.device("cuda", 0)
blob, offs = synth.synth_batch(64, 0, 1920, 1080, 85, "420", 120)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(64)]
files = [raws[i % 64] for i in range(1024)]
prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
ctx = B.Context(0)
d_blob = torch.from_numpy(prep.blob).to(dev)
n = 1024 * 1920 * 1080 * 3
out = torch.empty(n, dtype=torch.uint8, device=dev) if order == "before" else None
plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": 1024})
spacer = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device=dev) if order == "spacer" else None
if out is None:
    out = torch.empty(n, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    plan.execute(stream, out.data_ptr())
torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        plan.execute(stream, out.data_ptr())
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 50 * 1e3)
c = plan.device_buffers()["coef"]
print(f"{order:7s} coef {c:#x} out {out.data_ptr():#x} (out - coef {(out.data_ptr() - c) / (1 << 30):+.2f} GiB): " + " ".join(f"{r:.3f}" for r in res) + " ms per step", flush=True)
plan.close()
ctx.close()
