"""One-off stress run on the GPU box: many random baseline / progressive files through every route of the public
API, each compared with the CPU oracle bit for bit.  Not part of the test suite (minutes, not seconds):
    python tools/stress_parity.py [n_files] [seed]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools import synth
from oracle import oracle
from pyjpegdecoder_amd import BatchDecoder, _binding as B

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
layouts = ("420", "444", "422", "440", "grey", "411")
files, want = [], []
t0 = time.time()
for i in range(n_files):
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    q = int(rng.choice([5, 20, 50, 75, 85, 95, 100]))
    lay = layouts[int(rng.integers(0, len(layouts)))]
    ri = int(rng.choice([0, 0, 1, 2, 3, 7, 16, 50]))
    noise = float(rng.choice([0.0, 5.0, 25.0, 80.0]))
    f = synth.synth_jpeg(int(rng.integers(0, 1 << 30)), w, h, q, lay, ri, noise)
    files.append(f)
    want.append(oracle.decode(f)["rgb"])
try:
    from PIL import Image
    for i in range(n_files // 10):
        w, h = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(int(rng.integers(0, 1 << 30)), w, h)).save(
            b, "JPEG", quality=int(rng.choice([30, 75, 92])), subsampling=int(rng.integers(0, 3)), progressive=True)
        files.append(b.getvalue())
        want.append(oracle.decode(b.getvalue())["rgb"])
except ImportError:
    pass
print("made %d files + oracle answers in %.1f s" % (len(files), time.time() - t0))

def check(label, outs, transpose=False):
    bad = 0
    for i, (o, w) in enumerate(zip(outs, want)):
        a = o.cpu().numpy() if hasattr(o, "cpu") else o
        if transpose:
            a = np.swapaxes(a, 0, 1)
        if a.shape != w.shape or not np.array_equal(a, w):
            bad += 1
            if bad < 4:
                print("   MISMATCH", label, "file", i, a.shape, w.shape)
    print("%-44s %s" % (label, "ok" if bad == 0 else "%d MISMATCHES" % bad))
    return bad

total = 0
for seg in ("host", "gpu"):
    for layout in ("xmajor", "rowmajor"):
        dec = BatchDecoder(0, layout=layout, segment=seg)
        total += check(f"decode        segment={seg} layout={layout}", dec.decode(files), layout == "rowmajor")
        total += check(f"decode_device segment={seg} layout={layout}", dec.decode_device(files), layout == "rowmajor")
        dec.close()
for form in ("wave", "lanes", "lanes11", "sync"):
    B.set_option("MJ_HUFFMAN", form)
    dec = BatchDecoder(0, segment="host")
    total += check(f"decode        stage-1 form forced: {form}", dec.decode(files))
    dec.close()
B.set_option("MJ_HUFFMAN", None)
dec = BatchDecoder(0, segment="gpu")
chunks = [files[i:i + 97] for i in range(0, len(files), 97)]
outs = [t for batch in dec.decode_device_iter(chunks) for t in batch]
total += check("decode_device_iter, batches of 97", outs)
dec.close()
print("TOTAL MISMATCHES", total)
sys.exit(1 if total else 0)
