#!/usr/bin/env python3
"""One-off stress on the GPU box: thousands of files from tools/craft_jpeg.py — baseline files with any sampling factors 1..4
per component, progressive files with random scan scripts — through the public API in batches, against the oracle (which
tools/crosscheck_reference.py --crafted holds to the reference itself on the same generators).

    python tools/stress_crafted.py [n_files] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle                                      # noqa: E402
from pyjpegdecoder_amd import BatchDecoder                     # noqa: E402
from tools.craft_jpeg import random_baseline, random_progressive   # noqa: E402


def main():
    from pyjpegdecoder_amd import _binding as B
    for kv in filter(None, os.environ.get("MJ_OPTS", "").split(",")):      # e.g. MJ_OPTS=MJ_PROG_SPLIT=2: every refining AC scan as scout + parts
        B.set_option(*kv.split("=", 1))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dec = {"host": BatchDecoder(0, segment="host"), "gpu": BatchDecoder(0, segment="gpu"), "rowmajor": BatchDecoder(0, layout="rowmajor")}
    bad = done = 0
    t0 = time.time()
    while done < n:
        k = min(200, n - done)
        files = [(random_progressive if rng.integers(0, 2) else random_baseline)(rng, int(rng.integers(0, 1 << 30))) for _ in range(k)]
        want = [oracle.decode(f) for f in files]
        for name, d in dec.items():
            if name == "host":
                got, seams = d.decode(files, return_seams=True)
            else:
                got, seams = d.decode(files), None
            for i in range(k):
                ref = want[i]["rgb"] if name != "rowmajor" else np.ascontiguousarray(np.swapaxes(want[i]["rgb"], 0, 1))
                ok = got[i].shape == ref.shape and np.array_equal(got[i], ref)
                if ok and seams is not None:
                    ok = np.array_equal(seams[i]["coef"], want[i]["coef"])
                if not ok:
                    bad += 1
                    print(f"MISMATCH ({name}) file {done + i}: {len(files[i])} bytes", flush=True)
        done += k
        print(f"{done} files, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    for d in dec.values():
        d.close()
    print(f"{n} crafted files x 3 decoders: {bad} MISMATCHES")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
