"""One-off stress run on the GPU box for the progressive path: random Pillow-written progressive files (libjpeg's scan
script with successive approximation; colour in three samplings and greyscale; restart rows on some; smooth to very noisy)
through the stream walks with band pipelining (the default), with one-row bands, one launch per dependency level, and the
general walk — both pixel layouts — each compared with the CPU oracle bit for bit.
    python tools/stress_progressive.py [n_files] [seed]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from tools import synth
from oracle import oracle
from pyjpegdecoder_amd import BatchDecoder, _binding as _B

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
files, want = [], []
t0 = time.time()
for i in range(n_files):
    w, h = int(rng.integers(1, 520)), int(rng.integers(1, 400))
    rgb = synth.synth_rgb(int(rng.integers(0, 1 << 30)), w, h, float(rng.choice([0.0, 4.0, 20.0, 60.0])))
    kw = dict(quality=int(rng.choice([10, 35, 60, 85, 95, 100])), progressive=True)
    grey = rng.integers(0, 6) == 0
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.integers(0, 3) == 0:
        kw["restart_marker_rows"] = int(rng.integers(1, 5))
    if rng.integers(0, 4) == 0:
        kw["optimize"] = True
    b = io.BytesIO()
    try:
        Image.fromarray(rgb[..., 1] if grey else rgb).save(b, "JPEG", **kw)
    except OSError:             # the writer gives up on some tiny restart-row combinations
        continue
    files.append(b.getvalue())
    want.append(oracle.decode(b.getvalue())["rgb"])
print("made %d files + oracle answers in %.1f s" % (len(files), time.time() - t0), flush=True)
bad = 0
forms = {"bands": {}, "two-row bands": {"MJ_PROG_ROWS": "2"}, "levels": {"MJ_PROG_BANDS": "0"}, "general walk": {"MJ_PROG_FAST": "0"},
         "split scans": {"MJ_PROG_SPLIT": "2"}, "split, 3 rows, 3 parts": {"MJ_PROG_SPLIT": "2", "MJ_PROG_ROWS": "3", "MJ_PROG_PARTS": "3"},
         "split, 7 parts": {"MJ_PROG_SPLIT": "2", "MJ_PROG_PARTS": "7"},
         "chunks": {"MJ_PROG_CHUNKS": "2"}, "chunks of 128 B, split": {"MJ_PROG_CHUNKS": "2", "MJ_PROG_CHUNK": "128", "MJ_PROG_SPLIT": "2"}}
for name, env in forms.items():
    for k, v in env.items():
        _B.set_option(k, v)
    for layout in ("xmajor", "rowmajor"):
        dec = BatchDecoder(device=0, layout=layout)
        t0 = time.time()
        n_bad = 0
        for lo in range(0, len(files), 97):           # batches of mixed geometry
            outs = dec.decode(files[lo:lo + 97])
            for i, img in enumerate(outs):
                got = np.swapaxes(img, 0, 1) if layout == "rowmajor" else img
                if not np.array_equal(got, want[lo + i]):
                    n_bad += 1
        dec.close()
        print(f"{name:22s} {layout:9s}: {len(files)} files, {n_bad} mismatches, {time.time() - t0:.1f} s", flush=True)
        bad += n_bad
    for k in env:
        _B.set_option(k, None)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
