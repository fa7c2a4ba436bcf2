"""Stage timing probe (GPU box): builds the config-3 plan once and prints stage-1 / stage-2 HIP-event times under a list of
environment settings, one line each.  Usage:
    python tools/stage_probe.py [--lib pyjpegdecoder_amd/libmijpeg_diag.so] [--batch 1024] [--distinct 64] \
        [--ri 120] [--layout xmajor] [--subsampling 420] NAME=VALUE[,NAME=VALUE]... (one experiment per argument; "" = defaults)
Library switches (MJ_HUFFMAN, MJ_LANES_PER_WAVE, ...: mj_set_option) are set through the API, everything else (MJ_DEBUG_* of the
diagnostic build) as environment; every experiment gets a fresh plan."""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--ri", type=int, default=120)
    ap.add_argument("--layout", default="xmajor")
    ap.add_argument("--subsampling", default="420")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm", type=int, default=15, help="untimed executes before the timed ones (clock ramp)")
    ap.add_argument("--tune", type=int, default=4, help="coefficient stores a fused plan tries before it is timed (mj_plan_tune_placement; 1 = none)")
    ap.add_argument("--mixed", action="store_true", help="the heterogeneous family of bench.py's mixed_content instead of the homogeneous one")
    ap.add_argument("exps", nargs="*", default=[""])
    args = ap.parse_args()
    import numpy as np
    import torch
    from pyjpegdecoder_amd import _binding as B
    if args.lib:
        B.LIB_PATH = Path(args.lib).resolve()
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    dev = torch.device("cuda", 0)
    if args.mixed:
        blob, offs = synth.synth_mixed_batch(args.distinct, 900000, args.width, args.height, args.subsampling, args.ri)
    else:
        blob, offs = synth.synth_batch(args.distinct, 0, args.width, args.height, 85, args.subsampling, args.ri)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(args.distinct)]
    files = [raws[i % args.distinct] for i in range(args.batch)]
    layout = B.MJ_LAYOUT_XMAJOR if args.layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    prep = prepare_batch(files, layout, 0)
    ctx = B.Context(0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    d_rgb = None
    for exp in args.exps:
        sets = [kv.split("=", 1) for kv in exp.split(",") if kv]
        old = {k: os.environ.get(k) for k, _ in sets}
        api_opts = []
        for k, v in sets:          # the library's own switches go through mj_set_option; MJ_DEBUG_* etc. are the diagnostic build's environment
            try:
                B.set_option(k, v)
                api_opts.append(k)
            except B.UnknownOption:
                os.environ[k] = v
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
        if d_rgb is None:
            d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        for _ in range(args.warm):
            plan.execute(torch.cuda.current_stream().cuda_stream, d_rgb.data_ptr())
        torch.cuda.synchronize()
        import time
        st = torch.cuda.current_stream().cuda_stream
        if args.tune > 1 and plan.stage1_form() & B.MJ_FORM_FUSED:      # (the fast placement class for every line of a sweep: profiles/r06_placement.txt)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.7:
                plan.execute(st, d_rgb.data_ptr())
                torch.cuda.synchronize()
            plan.tune_placement(st, d_rgb.data_ptr(), args.tune)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            plan.execute(st, d_rgb.data_ptr())
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / args.iters * 1e3
        s1, s2 = plan.time_stages(args.iters, d_rgb.data_ptr())
        print(f"{exp or 'default':50s} stage0+1 {s1:7.3f} ms   stage2 {s2:7.3f} ms   step {step:7.3f} ms", flush=True)
        plan.close()
        for k in api_opts:
            B.set_option(k, None)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.close()


if __name__ == "__main__":
    main()
