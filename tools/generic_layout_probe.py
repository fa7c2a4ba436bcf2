"""Sampling layouts outside the common five (DESIGN.md "every other combination of sampling factors"): stage
times of 256 x 1080p per layout through the path such files take — the wave form of stage 1 (one restart segment per
wavefront) and the generic stage 2 (k_reconstruct_generic: one MCU per wavefront pass, exact-order IDCT) — with the first and
the last image of the batch held to the oracle.  The files are written symbol by symbol (tools/craft_jpeg.py: no encoder at
hand produces these layouts).   gpurun -- 'python tools/generic_layout_probe.py [--batch 256]'"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

LAYOUTS = {                      # name: factors (h, v) of Y, Cb, Cr
    "4:1:0 (Y 4x2)": [(4, 2), (1, 1), (1, 1)],
    "Y 2x2, Cb 2x1, Cr 1x1": [(2, 2), (2, 1), (1, 1)],
    "Y 1x1, chroma 2x2": [(1, 1), (2, 2), (2, 2)],
    "Y 3x1": [(3, 1), (1, 1), (1, 1)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import craft_jpeg
    W, H = 1920, 1080
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    for name, factors in LAYOUTS.items():
        hmax = max(h for h, _ in factors)
        t0 = time.perf_counter()
        raws = [craft_jpeg.craft_baseline(W, H, factors, seed=900 + i, restart_interval=-(-W // (8 * hmax))) for i in range(args.distinct)]
        files = [raws[i % args.distinct] for i in range(args.batch)]
        prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
        d_blob = torch.from_numpy(prep.blob).to(dev)
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        plan.execute(0, d_rgb.data_ptr())
        plan.sync()
        bad = int(plan.read(rgb=False)["status"].any())
        s1, s2 = plan.time_stages(args.iters, d_rgb.data_ptr())
        per = W * H * 3
        ok = all(np.array_equal(d_rgb[i * per:(i + 1) * per].cpu().numpy().reshape(W, H, 3), oracle.decode(files[i])["rgb"]) for i in (0, args.batch - 1))
        mp = args.batch * W * H / 1e6
        print(f"{name:24s} {args.batch} x 1080p ({sum(map(len, raws)) // len(raws) // 1024} KB each): stage 1 form {plan.stage1_form() & 15} {s1:8.2f} ms, "
              f"stage 2 {s2:8.2f} ms = {mp / ((s1 + s2) * 1e-3):8.0f} MP/s; status {'BAD' if bad else 'ok'}, first and last image vs oracle "
              f"{'equal' if ok else 'DIFFER'}  (files written in {time.perf_counter() - t0:.0f} s)", flush=True)
        plan.close()
        del d_rgb, d_blob
    ctx.close()


if __name__ == "__main__":
    main()
