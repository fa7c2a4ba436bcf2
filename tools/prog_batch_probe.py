"""Progressive (config 5) throughput against batch size: the scan walk is one wavefront per image and scan, latency bound,
so a batch's time is one image's serial chain until the chip's wave slots are full.  python tools/prog_batch_probe.py 1024 2048 4096"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from tools import synth
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
W, H, nd = 1920, 1080, 8
if os.environ.get("MJ_PROBE_LIB"):      # e.g. pyjpegdecoder_amd/libmijpeg_diag.so: its stage timing prints the refining walk's phase shares
    from pathlib import Path
    B.LIB_PATH = Path(os.environ["MJ_PROBE_LIB"]).resolve()
for kv in filter(None, os.environ.get("MJ_OPTS", "").split(",")):      # e.g. MJ_OPTS=MJ_PROG_SPLIT=0,MJ_PROG_ROWS=1
    B.set_option(*kv.split("=", 1))
raws = []
for i in range(nd):
    b = io.BytesIO(); Image.fromarray(synth.synth_rgb(500000 + i, W, H)).save(b, "JPEG", quality=85, subsampling=2, progressive=True); raws.append(b.getvalue())
ctx = B.Context(0); dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [1024]:
    files = [raws[i % nd] for i in range(n)]
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
    plan.execute(0, d_rgb.data_ptr()); plan.sync()
    t = time.perf_counter()
    for _ in range(2): plan.execute(0, d_rgb.data_ptr())
    plan.sync(); dt = (time.perf_counter() - t) / 2
    ok = not plan.read(rgb=False)["status"].any()
    if os.environ.get("MJ_PROBE_LIB"):
        print("stage times", plan.time_stages(1, d_rgb.data_ptr()), flush=True)
    print(f"{n} x 1080p progressive: {dt*1e3:.1f} ms per batch = {n*W*H/1e6/dt:.0f} MP/s  status ok {ok}", flush=True)
    plan.close(); del d_rgb, d_blob
    torch.cuda.empty_cache()
