#!/usr/bin/env python3
"""Pin the oracle wider: random small files decoded by THE REFERENCE (imported here, in this container only) and by
oracle/ — final image and zig-zag coefficients must be identical.  tests/golden/ holds a handful of captured fixtures;
this is the same comparison on as many random files as one cares to wait for (the reference does ~0.04 MP/s).
Not part of the test suite and of no use on the GPU box (/root/reference does not exist there).

    python tools/crosscheck_reference.py [n_files] [seed] [processes] [--crafted]

--crafted: the files of tools/craft_jpeg.py instead — baseline files with any sampling factors 1..4 per component,
progressive files with random scan scripts (DC scans over subsets of the components, random bands and refinement levels) in
every layout the reference can finish.
"""
import contextlib
import io
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


CRAFTED = "--crafted" in sys.argv


def one(args):
    i, seed = args
    from tools import make_goldens as mg          # imports the reference
    from tools import synth
    from oracle import oracle
    rng = np.random.default_rng([seed, i])
    if CRAFTED:
        from tools.craft_jpeg import random_baseline, random_progressive
        prog = bool(rng.integers(0, 2))
        raw = (random_progressive if prog else random_baseline)(rng, int(rng.integers(0, 1 << 30)))
        desc = f"crafted {'progressive' if prog else 'baseline'} #{i}, {len(raw)} bytes"
        return compare(i, desc, raw, prog, mg, oracle)
    w, h = int(rng.integers(1, 97)), int(rng.integers(1, 97))
    q = int(rng.choice([10, 40, 75, 85, 95, 100]))
    noise = float(rng.choice([0.0, 8.0, 30.0, 90.0]))
    kind = int(rng.integers(0, 10))
    if kind < 7:
        lay = ("420", "444", "422", "440", "grey", "411")[int(rng.integers(0, 6))]
        ri = int(rng.choice([0, 0, 1, 2, 5, 9]))
        raw = synth.synth_jpeg(int(rng.integers(0, 1 << 30)), w, h, q, lay, ri, noise)
        desc = f"baseline {w}x{h} q{q} {lay} ri{ri} noise{noise}"
        prog = False
    else:
        from PIL import Image
        b = io.BytesIO()
        sub = int(rng.integers(0, 3))
        w, h = max(w, 8), max(h, 8)
        Image.fromarray(synth.synth_rgb(int(rng.integers(0, 1 << 30)), w, h, noise)).save(
            b, "JPEG", quality=q, subsampling=sub, progressive=True)
        raw = b.getvalue()
        desc = f"progressive {w}x{h} q{q} sub{sub} noise{noise}"
        prog = True
    return compare(i, desc, raw, prog, mg, oracle)


def compare(i, desc, raw, prog, mg, oracle):
    with tempfile.TemporaryDirectory() as d:
        path = Path(d) / "f.jpg"
        path.write_bytes(raw)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                if prog:
                    dec = mg.jd.JpegDecoder(path)
                    ref_rgb, ref_coef = np.asarray(dec.image_array), None
                else:
                    dec, cap = mg.run_reference(path)
                    ref_rgb, ref_coef = np.asarray(dec.image_array), np.stack(cap["coef"])
        except Exception as exc:                      # the reference's own limits (e.g. 1-pixel-wide progressive)
            return i, desc, "reference raised " + type(exc).__name__
    out = oracle.decode(raw)
    if ref_rgb.shape != out["rgb"].shape or not np.array_equal(ref_rgb, out["rgb"]):
        return i, desc, "IMAGE DIFFERS"
    if ref_coef is not None and not np.array_equal(ref_coef, out["coef"]):
        return i, desc, "COEFFICIENTS DIFFER"
    return i, desc, "ok"


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if len(argv) > 0 else 200
    seed = int(argv[1]) if len(argv) > 1 else 1
    procs = int(argv[2]) if len(argv) > 2 else min(8, os.cpu_count() or 1)
    from multiprocessing import Pool
    with Pool(procs) as pool:
        res = pool.map(one, [(i, seed) for i in range(n)], chunksize=4)
    ok = sum(r[2] == "ok" for r in res)
    raised = [r for r in res if r[2].startswith("reference raised")]
    bad = [r for r in res if r[2] not in ("ok",) and not r[2].startswith("reference raised")]
    for r in bad[:20]:
        print("MISMATCH", r)
    for r in raised[:5]:
        print("note:", r)
    print(f"{n} files: {ok} identical, {len(raised)} the reference itself could not decode, {len(bad)} MISMATCHES")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
