"""Fused launch (fused.hip) against the two launches on the GPU box: the config-3 batch (or another shape) decoded with
MJ_FUSED=0 and with the fused kernel under a list of settings; every fused output is compared byte for byte with the
two-launch output (which bench.py and the tests hold to the oracle), every image's status is looked at, and whole steps
are timed.  Usage:
    python tools/fused_probe.py [--batch 1024] [--distinct 64] [--subsampling 420] [--width 1920 --height 1080] [--segment gpu] \
        NAME=VALUE[,NAME=VALUE]... (one experiment per argument; "" = defaults)"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--subsampling", default="420")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--ri", type=int, default=0, help="restart interval in MCUs (0 = one MCU row)")
    ap.add_argument("--segment", default="host", choices=["host", "gpu"])
    ap.add_argument("--layout", default="xmajor", choices=["xmajor", "rowmajor"])
    ap.add_argument("--mixed", action="store_true", help="files of mixed content (bench.py's mixed_content family)")
    ap.add_argument("--tune", type=int, default=4, help="coefficient stores a fused plan tries before it is timed (1 = none)")
    ap.add_argument("--lib", default=None, help="another build of the library (pyjpegdecoder_amd/libmijpeg_diag.so + MJ_DEBUG_FUSED=1: phase times)")
    ap.add_argument("exps", nargs="*", default=[""])
    args = ap.parse_args()
    import numpy as np
    import torch
    from pyjpegdecoder_amd import _binding as B
    if args.lib:
        B.LIB_PATH = Path(args.lib).resolve()
    from pyjpegdecoder_amd import parse_jpeg
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    dev = torch.device("cuda", 0)
    mcus_per_row = (args.width + (15 if args.subsampling in ("420", "422") else 7)) // (16 if args.subsampling in ("420", "422") else 8)
    if args.mixed:
        blob, offs = synth.synth_mixed_batch(args.distinct, 900000, args.width, args.height, args.subsampling, args.ri or mcus_per_row)
    else:
        blob, offs = synth.synth_batch(args.distinct, 0, args.width, args.height, 85, args.subsampling, args.ri or mcus_per_row)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(args.distinct)]
    files = [raws[i % args.distinct] for i in range(args.batch)]
    layout = B.MJ_LAYOUT_XMAJOR if args.layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    if args.segment == "gpu":
        prep = prepare_batch(files, layout, 0, [parse_jpeg(f, headers_only=True) for f in files])
    else:
        prep = prepare_batch(files, layout, 0)
    ctx = B.Context(0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    # ONE output buffer for every experiment (round 6: which block the allocator hands out decides between two classes of step,
    # 5-9 % apart — profiles/r06_placement.txt), and every fused plan picks its coefficient store against it before it is timed
    keep = {}

    def run(opts):
        for k, v in opts:
            B.set_option(k, v)
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
        if "out" not in keep:
            keep["out"] = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        out = keep["out"]
        out.zero_()
        try:
            form = plan.stage1_form()
            for _ in range(10):
                plan.execute(stream, out.data_ptr())
            torch.cuda.synchronize()
            if args.tune > 1:
                plan.tune_placement(stream, out.data_ptr(), args.tune)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                plan.execute(stream, out.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            st = plan.read(rgb=False)["status"]
            parts = plan.time_execute(5, out.data_ptr())
        finally:
            plan.close()
            for k, _ in opts:
                B.set_option(k, None)
        return out, ms, form, st, parts
    ref, ms0, form0, st0, parts0 = run([("MJ_FUSED", "0")])
    ref = ref.clone()
    print(f"{'two launches (MJ_FUSED=0)':50s} form {form0:3d}  {ms0:7.3f} ms per step ({parts0[0]:.3f} + {parts0[1]:.3f})   statuses not ok: {int((st0 != 0).sum())}", flush=True)
    for exp in args.exps:
        opts = [kv.split("=", 1) for kv in exp.split(",") if kv]
        out, ms, form, st, parts = run(opts)
        same = bool(torch.equal(out, ref))
        print(f"{exp or 'default':50s} form {form:3d}  {ms:7.3f} ms per step ({parts[0]:.3f} + {parts[1]:.3f})   statuses not ok: {int((st != 0).sum())}   "
              f"{'identical to the two launches' if same else 'DIFFERS from the two launches'}", flush=True)
        if not same:
            d = (out != ref).view(args.batch, -1).any(dim=1).nonzero().flatten()
            print("   images that differ:", d[:16].tolist(), "of", int(d.numel()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
