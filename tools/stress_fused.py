"""One-off stress run on the GPU box for the fused launch (fused.hip) after round 6 generalised it: random UNIFORM batches — sampling
layout, image size, image count, restart interval (one row, fractions of a row, several rows, unrelated to the row), pixel layout,
who finds the markers, number of consumers — each decoded by the plan's own execute (one fused launch where form_select.h allows
it) and by the two launches (MJ_FUSED=0), coefficient store poisoned before every execute; outputs compared byte for byte, every
distinct file against the oracle.  With `damage` as third argument a few files of every batch get entropy-coded bytes overwritten
(no marker made or unmade): the statuses must be the two launches', no MJ_ST_INTERNAL (a consumer that gave up waiting), the
images that still decode identical, and no execute may take as long as the consumers' guard.  Not part of the test suite (minutes):
    python tools/stress_fused.py [n_trials] [seed] [damage] [mixed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd import parse_jpeg
from pyjpegdecoder_amd.batch import prepare_batch
from tools import synth

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
damage = "damage" in sys.argv[3:]
mixed_too = "mixed" in sys.argv[3:]          # every third trial on files of mixed content
slowest = 0.0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
ctx = B.Context(0)


def decode(prep, n, opts):
    for k, v in opts:
        B.set_option(k, v)
    try:
        d_blob = torch.from_numpy(prep.blob).to(dev)
        torch.cuda.synchronize()
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
        try:
            out = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
            keep = None
            for poison in (0x5A, 0xC3):
                out.zero_()
                torch.cuda.synchronize()
                plan.fill_coef(poison)
                t_ex = time.perf_counter()
                plan.execute(0, out.data_ptr())
                plan.sync()
                global slowest
                slowest = max(slowest, time.perf_counter() - t_ex)
                if keep is None:
                    keep = out.clone()
                elif not torch.equal(out, keep):
                    return None, None, plan.stage1_form()
            return keep, plan.read(rgb=False)["status"], plan.stage1_form()
        finally:
            plan.close()
    finally:
        for k, _ in opts:
            B.set_option(k, None)


bad = fused_n = 0
t0 = time.time()
for trial in range(n_trials):
    ss = str(rng.choice(["420", "422", "440", "444", "411"]))
    mw = 32 if ss == "411" else (16 if ss in ("420", "422") else 8)
    mh = 16 if ss in ("420", "440") else 8
    W, H = int(rng.integers(64, 1500)), int(rng.integers(48, 1200))
    mpr, mcv = -(-W // mw), -(-H // mh)
    kind = int(rng.integers(0, 6))
    divs = [d for d in range(2, 9) if mpr % d == 0]
    if kind == 0 or (kind == 1 and not divs):
        ri = mpr
    elif kind == 1:
        ri = mpr // int(rng.choice(divs))
    elif kind == 2:
        ri = mpr * int(rng.integers(2, 4))
    else:
        ri = int(rng.integers(max(4, mpr // 6), 3 * mpr))
    spi = -(-(mpr * mcv) // ri)
    # enough segments for the lane form, not more pixels than ~3 GB of output
    n_lo = max(8, -(-1100 // spi))
    n_hi = max(n_lo + 1, min(4000, int(3e9 // (W * H * 3))))
    n = int(rng.integers(n_lo, n_hi))
    distinct = int(rng.integers(2, 5))
    layout = str(rng.choice(["xmajor", "rowmajor"]))
    if layout == "xmajor" and mpr % ri != 0 and ri != 2 * mpr and rng.integers(0, 3) > 0:
        layout = "rowmajor"                     # (x-major plans only fuse intervals that divide the row, or of two rows: keep most trials on fused ground)
    lay = B.MJ_LAYOUT_XMAJOR if layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    gpu_seg = bool(rng.integers(0, 2))
    cons = rng.choice([None, "1", "3", "8"])
    fseed, q = int(rng.integers(0, 1 << 30)), int(rng.choice([50, 75, 85, 92]))
    mixed = mixed_too and trial % 3 == 2       # files of mixed content: the segments are dealt out by length, the hand-off crosses workgroups
    if mixed:
        distinct = int(rng.integers(8, 33))
        blob, offs = synth.synth_mixed_batch(distinct, fseed, W, H, ss, ri)
    else:
        try:
            blob, offs = synth.synth_batch(distinct, fseed, W, H, q, ss, ri)
        except RuntimeError:                    # (the writer's output buffer is one byte per pixel: 4:4:4 at high quality can exceed it)
            blob, offs = synth.synth_batch(distinct, fseed, W, H, 60, ss, ri)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    files = [raws[(3 * i + i // distinct) % distinct] for i in range(n)]
    hurt = []
    if damage:
        hurt = sorted(set(int(x) for x in rng.integers(0, n, size=max(1, n // 60))))
        for i in hurt:
            b = bytearray(files[i])
            lo, done, tries = len(b) // 3, 0, 0
            while done < int(rng.integers(1, 9)) and tries < 200:
                tries += 1
                pos = int(rng.integers(lo, len(b) - 4))
                v = int(rng.integers(0, 255))
                if 0xFF in (b[pos - 1], b[pos], b[pos + 1]) or v == b[pos]:
                    continue                    # (no marker made, none unmade: the segmentation stands, on the host and on the GPU)
                b[pos] = v
                done += 1
            files[i] = bytes(b)
    parsed = [parse_jpeg(f, headers_only=True) for f in files] if gpu_seg else None
    prep = prepare_batch(files, lay, 0, parsed)
    base = [("MJ_HUFFMAN", "lanes")]
    two, st2, form2 = decode(prep, n, base + [("MJ_FUSED", "0")])
    one, st1, form1 = decode(prep, n, base + ([("MJ_FUSED_CONSUMERS", cons)] if cons else []))
    per = W * H * 3
    if damage:
        ok = two is not None and one is not None and np.array_equal(st1 != 0, st2 != 0) and set(np.flatnonzero(st1)) <= set(hurt) and \
            B.MJ_ST_INTERNAL not in st1 and B.MJ_ST_INTERNAL not in st2
        if ok:
            good = torch.tensor([i for i in range(n) if st1[i] == 0], device=dev)
            ok = bool(torch.equal(one.view(n, per)[good], two.view(n, per)[good]))
    else:
        ok = two is not None and one is not None and not st2.any() and not st1.any() and bool(torch.equal(one, two))
    if ok:
        imgs = two.view(n, per)
        for d in range(distinct):
            i = next((k for k in range(n) if (3 * k + k // distinct) % distinct == d and k not in hurt), None)
            if i is None:                       # (every instance of this file was damaged)
                continue
            got = imgs[i].cpu().numpy()
            got = got.reshape(W, H, 3) if layout == "xmajor" else np.swapaxes(got.reshape(H, W, 3), 0, 1)
            ok = ok and np.array_equal(got, oracle.decode(raws[d])["rgb"])
    fused = bool(form1 & B.MJ_FORM_FUSED)
    fused_n += fused
    if trial == 0:
        slowest = 0.0                           # (the first executes load the code object)
    bad += not ok
    print(f"trial {trial:3d}: {ss} {W}x{H} x{n} ri={ri} ({mpr} MCUs per row, {spi} segments per image) {'mixed content ' if mixed else ''}{layout} markers by {'gpu' if gpu_seg else 'host'} "
          f"consumers {cons}: {'fused' if fused else 'two launches'} (form {form1}) {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_trials} trials{' with damaged files' if damage else ''}, {fused_n} of them through a fused launch, {bad} mismatches, "
      f"slowest execute {slowest * 1e3:.0f} ms, {time.time() - t0:.0f} s")
bad += damage and slowest > 1.5
ctx.close()
sys.exit(1 if bad else 0)
