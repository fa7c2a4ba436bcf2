"""Time the reference itself (/root/reference/jpeg_decoder.py, imported, never copied) on this container's cores:
BASELINE.md plan step 1.  One process, then eight at once, one 1080p 4:2:0 DRI=120 q85 synthetic file each (the bench
workload's family; ~50 s per file).  Writes profiles/reference_python_timing.json, which bench.py quotes as
cpu_baseline.reference_python (the reference cannot travel to the GPU box; these numbers are from the build container).

    python tools/time_reference.py [--procs 1 8] [--out profiles/reference_python_timing.json]
"""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import platform
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
W, H = 1920, 1080


def _decode(path):
    sys.path.insert(0, "/root/reference")
    import jpeg_decoder as jd
    jd.JpegDecoder.show = lambda self: None          # no viewer
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):   # the reference prints per MCU
        d = jd.JpegDecoder(Path(path))
    dt = time.perf_counter() - t0
    assert d.image_array.shape == (W, H, 3)
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--out", default=str(ROOT / "profiles" / "reference_python_timing.json"))
    args = ap.parse_args()
    from tools import synth
    tmp = Path(tempfile.mkdtemp(prefix="mjref_"))
    nmax = max(args.procs)
    for i in range(nmax):
        (tmp / f"{i}.jpg").write_bytes(synth.synth_jpeg(100000 * 0 + i, W, H, 85, "420", 120))
    runs = []
    for n in args.procs:
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(n) as pool:
            per = pool.map(_decode, [str(tmp / f"{i}.jpg") for i in range(n)])
        wall = time.perf_counter() - t0
        runs.append({"processes": n, "images": n, "wall_s": round(wall, 2), "per_image_s": [round(x, 2) for x in per],
                     "value": round(n * W * H / 1e6 / max(per), 4), "unit": "MP/s",
                     "note": "n files decoded concurrently, one per process; value = n x 2.0736 MP / slowest decode"})
        print(runs[-1], flush=True)
    try:
        commit = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = None
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out = {"what": "the reference (/root/reference/jpeg_decoder.py) timed in the build container by tools/time_reference.py",
           "workload": "1920x1080 4:2:0 baseline JPEG, q85, DRI=120 (tools/synth.py seeds 0..n-1)",
           "host": {"cpu": cpu, "cores": os.cpu_count(), "python": platform.python_version()},
           "script_commit": commit, "runs": runs}
    Path(args.out).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
