"""Can stage 1 of one part of a batch run under stage 2 of another?  The config-3 batch as P plans of 1024 / P images each, stage 1 on
one stream, stage 2 on a second one behind an event per part, against the one-plan execute.
    python tools/overlap_probe.py [2 4 8]            OVERLAP_RI=0 (files without restart markers), OVERLAP_N=1024 in the environment"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch                                                      # noqa: E402
from pyjpegdecoder_amd import _binding as B                       # noqa: E402
from pyjpegdecoder_amd.batch import prepare_batch                 # noqa: E402
from tools import synth                                           # noqa: E402

N, ND, W, H = int(os.environ.get("OVERLAP_N", "1024")), 64, 1920, 1080
RI = int(os.environ.get("OVERLAP_RI", "120"))
dev = torch.device("cuda", 0)
blob, offs = synth.synth_batch(ND, 0, W, H, 85, "420", RI)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(ND)]
files = [raws[i % ND] for i in range(N)]
ctx = B.Context(0)


def make(fs):
    prep = prepare_batch(fs, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(fs), "blob": d_blob})
    return plan


whole = make(files)
d_rgb = torch.empty(whole.info.rgb_bytes, dtype=torch.uint8, device=dev)
s_main = torch.cuda.current_stream().cuda_stream
for _ in range(15):
    whole.execute(s_main, d_rgb.data_ptr())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    whole.execute(s_main, d_rgb.data_ptr())
torch.cuda.synchronize()
base = (time.perf_counter() - t) / 20
print(f"DRI={RI}: one plan of {N}: {base * 1e3:.3f} ms per batch", flush=True)
ref = d_rgb.clone()

for parts in [int(a) for a in sys.argv[1:]] or [2, 4]:
    per = N // parts
    plans = [make(files[i * per:(i + 1) * per]) for i in range(parts)]
    rgb_per = whole.info.rgb_bytes // parts
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(parts)]
    d_rgb.zero_()

    def step():
        for i, p in enumerate(plans):
            p.execute_stage1(sa.cuda_stream)
            evs[i].record(sa)
            sb.wait_event(evs[i])
            p.execute_stage2(sb.cuda_stream, d_rgb.data_ptr() + i * rgb_per)
        sa.wait_stream(sb)                                         # the next batch's stage 1 must not overwrite coefficients stage 2 still reads

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    ok = torch.equal(d_rgb, ref)
    print(f"{parts} plans of {per}, stage 2 of part i under stage 1 of part i + 1: {dt * 1e3:.3f} ms per batch ({base / dt:.2f} x), output identical: {ok}", flush=True)
    for p in plans:
        p.close()
whole.close()
ctx.close()
