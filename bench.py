#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric (megapixels/s decoded on 1080p 4:2:0 baseline batches) on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          N > 1: starts the N ranks itself (torch.distributed.run as a child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (the same ranks, launcher outside)

Default workload = BASELINE configs[2]: per GPU a batch of 1024 synthetic 1920x1080 4:2:0 baseline JPEGs with DRI
restart markers, on-GPU Huffman + IDCT.  A "step" = one pass of the hot path (stage 0+1 Huffman decode, stage 2
dequant/IDCT/upsample/colour) over the rank's whole batch, compressed input already resident in HBM, RGB left in HBM.
Images are the units; they are sharded over ranks with no collective on the data path — weak scaling:
every rank decodes its own `--batch` images; the only cross-rank traffic is a gloo barrier and the MAX of the timings.

    python bench.py --total-images 10000 [--gpus N]      BASELINE configs[3]: ONE job of that many images, sharded over
                                                         the ranks (1 250 per GPU at N = 8), each rank feeding its share
                                                         through a per-GPU image queue (strong scaling)

Rank 0 prints ONE JSON line.  At N = 1 it also carries: `roofline` (+ `roofline_other_stage`; `two_launches` when the step is
one fused launch), `cpu_baseline` (the C oracle on this host's cores + the reference's own timing from the build
container), `idct_only` (BASELINE configs[1]), `progressive` (BASELINE configs[4]: 1024 x 1080p progressive),
`gpu_segmented`, `stage2_exact_only_ms`, `single_file_latency`, `mixed_content`, `without_restart_markers`,
`host_bytes_to_device_pixels`.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W, H = 1920, 1080
BLOCKS_PER_IMAGE = 48960                      # 8160 MCUs x 6 blocks
STAGE2_BYTES_PER_IMAGE = BLOCKS_PER_IMAGE * 128 + W * H * 3       # 12 487 680
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def sources_sha16() -> str:
    """Fingerprint of every source of libmijpeg.so (kernels, launch geometry, headers): PMC summaries under profiles/ are
    only quoted for the build they came from."""
    h = hashlib.sha256()
    d = ROOT / "pyjpegdecoder_amd" / "csrc"
    for f in sorted(list(d.glob("*.hip")) + list(d.glob("*.h")) + list(d.glob("*.cpp")) + [ROOT / "include" / "mijpeg.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def images_not_matching_the_oracle(raws, fetch, layout_name: str = "xmajor", w: int = W, h: int = H, workers=None):
    """Every file of `raws` decoded by the oracle on a pool of host threads (the C oracle runs outside the GIL) and compared with
    what the GPU left: fetch(i) -> that image's bytes as a flat uint8 array.  Returns the indices that differ.  Checker only:
    called after the timed regions."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    if not raws:
        return []
    oracle.decode(raws[0])                            # (builds the oracle's tables before the threads start)

    def one(i):
        ref = oracle.decode(raws[i])["rgb"]
        got = fetch(i)
        got = got.reshape(ref.shape) if layout_name == "xmajor" else np.swapaxes(got.reshape((h, w) + ref.shape[2:]), 0, 1)
        return None if np.array_equal(got, ref) else i
    with ThreadPoolExecutor(max_workers=workers or max(1, min(os.cpu_count() or 1, 32))) as ex:
        return sorted(i for i in ex.map(one, range(len(raws))) if i is not None)


def timed_executes(torch, plan, stream, out_ptr, reps: int, warm_s: float = 1.0, warm_min: int = 3, tune: int = 4):
    """Seconds per plan.execute over `reps` back-to-back executes, after at least `warm_s` seconds of them: a plan that has just
    been created starts on a chip that idled through its host-side preparation, and launches timed in the first second after
    such a pause came out long (the clock ramps; round 5's gpu_segmented object: 7.13 ms per step against 6.04 of kernels)."""
    t0 = time.perf_counter()
    n = 0
    while n < warm_min or time.perf_counter() - t0 < warm_s:
        plan.execute(stream, out_ptr)
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if tune > 1:            # (fused plans only, a no-op for the others: the fast placement class, as for the headline — profiles/r06_placement.txt)
        plan.tune_placement(stream, out_ptr, tune)
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(stream, out_ptr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cpu_baseline(raws, budget_s: float = 20.0):
    """The CPU oracle (bit-exact restatement of the reference path) timed on this host, one thread, then all cores."""
    import ctypes
    from oracle import oracle
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
    except OSError:
        pass
    oracle.decode(raws[0])                    # warm (builds tables)
    t0 = time.perf_counter()
    n = 0
    for r in raws:
        oracle.decode(r)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": round(n * W * H / 1e6 / dt, 3), "unit": "MP/s", "cores": 1, "kind": "port",
           "sample": f"{n} of the batch's 1080p images, {dt:.1f} s, oracle/jpeg_oracle.c single-threaded"}
    try:
        from concurrent.futures import ThreadPoolExecutor
        nc = max(1, min(os.cpu_count() or 1, 64))
        work = [raws[i % len(raws)] for i in range(min(4 * nc, 256))]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nc) as ex:
            list(ex.map(oracle.decode, work))
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round(len(work) * W * H / 1e6 / dt2, 3), "unit": "MP/s", "cores": nc,
                            "sample": f"{len(work)} images on {nc} threads, {dt2:.1f} s"}
    except Exception as exc:                     # never let the baseline break the bench line
        out["all_cores"] = {"error": str(exc)}
    # the reference itself cannot travel to the GPU box: its timing comes from the build container (tools/time_reference.py)
    ref = ROOT / "profiles" / "reference_python_timing.json"
    if ref.exists():
        try:
            d = json.loads(ref.read_text())
            out["reference_python"] = {"runs": [{k: r[k] for k in ("processes", "value", "unit", "wall_s")} for r in d["runs"]],
                                       "host": d["host"], "script": "tools/time_reference.py", "script_commit": d.get("script_commit"),
                                       "note": "the reference's pure-Python path, timed in the build container, not on this host"}
        except Exception as exc:
            out["reference_python"] = {"error": str(exc)}
    return out


def pipelined_side(ctx, dev, torch, prep, d_blob, plan, d_rgb, depth: int = 3, rounds: int = 20):
    """The same step with `depth` plans of the batch in flight on their own streams (each with its own coefficient store and
    output buffer): stage 1 is latency bound and leaves issue slots that another plan's stage 2 takes.  Beside the headline —
    `value` stays one step at a time."""
    import numpy as np
    from pyjpegdecoder_amd import _binding as B
    plans, outs = [plan], [d_rgb]
    for _ in range(depth - 1):
        plans.append(B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(prep.parsed)}))
        outs.append(torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev))
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    try:
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 1.0:                     # (a second of it first: see timed_executes)
            for q, o, st in zip(plans, outs, streams):
                q.execute(st.cuda_stream, o.data_ptr())
            torch.cuda.synchronize()
        for q, o, st in zip(plans[1:], outs[1:], streams[1:]):      # (the headline plan has picked its store already)
            q.tune_placement(st.cuda_stream, o.data_ptr(), 4)
        t0 = time.perf_counter()
        for _ in range(rounds):
            for q, o, st in zip(plans, outs, streams):
                q.execute(st.cuda_stream, o.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (rounds * depth)
        ok = all(not q.read(rgb=False)["status"].any() for q in plans) and all(bool(torch.equal(o, d_rgb)) for o in outs[1:])
    finally:
        for q in plans[1:]:
            q.close()
    n = plan.info.total_pixels
    return {"value": round(n / 1e6 / dt, 1), "unit": "MP/s", "ms_per_step": round(dt * 1e3, 3), "plans_in_flight": depth,
            "steps": rounds * depth, "parity": "every plan's output identical to the headline plan's" if ok else "MISMATCH",
            "note": "the headline step issued round-robin on separate plans and streams, inputs shared, outputs separate; not `value`. "
                    "A fused launch takes every CU's whole LDS, so a context's fused launches take turns (an event chain, api.hip): "
                    "plans in flight cost nothing and gain nothing on the kernels — what overlaps is the host side (plan creation, uploads)"}


def progressive_side(ctx, dev, torch, n_images: int = 1024, n_distinct: int = 8):
    """BASELINE configs[4]: a batch of 1080p 4:2:0 progressive files (libjpeg's default 10-scan script, written by Pillow),
    resident in HBM, decoded scan by scan; parity of two images against the oracle."""
    import io
    import numpy as np
    from PIL import Image
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    raws = []
    for i in range(n_distinct):
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(500000 + i, W, H)).save(b, "JPEG", quality=85, subsampling=2, progressive=True)
        raws.append(b.getvalue())
    files = [raws[i % n_distinct] for i in range(n_images)]
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n_images})
    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    per = W * H * 3

    def bad_of(m):          # every distinct file among the first m images against the oracle
        nd = min(m, n_distinct)
        return images_not_matching_the_oracle(raws[:nd], lambda i: d_rgb[i * per:(i + 1) * per].cpu().numpy())
    try:
        dt = timed_executes(torch, plan, stream, d_rgb.data_ptr(), 3, warm_s=0.3, warm_min=1)
        s1, s2 = plan.time_stages(1, d_rgb.data_ptr())
        plan.execute(stream, d_rgb.data_ptr())
        torch.cuda.synchronize()
        bad = bad_of(n_images)
        ok = not plan.read(rgb=False)["status"].any() and not bad
    finally:
        plan.close()
    # the same files in smaller batches: below ~900 files the plan walks the luma refinements as scout + parts
    smaller = []
    for m in (16, 256):
        if m >= n_images:
            continue
        prep_m = prepare_batch(files[:m], B.MJ_LAYOUT_XMAJOR, 0)
        d_blob_m = torch.from_numpy(prep_m.blob).to(dev)
        plan_m = B.Plan(ctx, prep_m.to_c(d_blob_m.data_ptr()), {"prep": prep_m, "n_images": m})
        try:
            dt_m = timed_executes(torch, plan_m, stream, d_rgb.data_ptr(), 3, warm_s=0.3, warm_min=1)
            bad_m = bad_of(m)
            ok_m = not plan_m.read(rgb=False)["status"].any() and not bad_m
            smaller.append({"images": m, "ms_per_step": round(dt_m * 1e3, 2), "value": round(m * W * H / 1e6 / dt_m, 1),
                            "parity": f"bit-exact vs oracle (all {min(m, n_distinct)} distinct files)" if ok_m else f"MISMATCH (files {bad_m[:8]})"})
        finally:
            plan_m.close()
    # algorithmic bytes of the scan walks: the entropy-coded bytes once, plus for every scan the coefficients it covers —
    # 2 B x (Se - Ss + 1) per block of its components, written by a first scan, read and written by a refining one
    from pyjpegdecoder_amd import parse_jpeg
    scan_bytes = 0
    p0 = parse_jpeg(raws[0])
    comp_blocks = {cid: (c.horizontal_sampling * c.vertical_sampling) * (BLOCKS_PER_IMAGE // 6) for cid, c in p0.color_components.items()}
    for sc in p0.scans:
        per_block = 2 * (sc.spectral_end - sc.spectral_start + 1) * (2 if sc.bit_high else 1)
        scan_bytes += per_block * sum(comp_blocks[c] for c in sc.component_ids)
    ent = int(sum(map(len, raws)) // n_distinct)
    s1_bytes = n_images * (ent + scan_bytes)
    roof = {"kernel": "k_destuff_pieces + k_progressive_fast + k_progressive_scan (all launches of stage 1)", "bound": "hbm",
            "achieved": round(s1_bytes / (s1 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(s1_bytes / (s1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_step": int(s1_bytes), "stage1_ms": round(s1, 3),
            "note": "entropy bytes + per scan 2 B x (Se-Ss+1) per covered block (x2 for refining scans: read-modify-write); "
                    "serial-walk (instruction issue) bound, quoted against HBM as SURVEY 8d asks; launches per step and their "
                    "average durations: profiles/r06*_progressive_kernel_stats.csv"}
    return {"value": round(n_images * W * H / 1e6 / dt, 1), "unit": "MP/s", "ms_per_step": round(dt * 1e3, 2),
            "stage1_ms": round(s1, 2), "stage2_ms": round(s2, 3), "roofline": roof,
            "workload": f"{n_images} x 1920x1080 4:2:0 progressive JPEG (Pillow/libjpeg default scan script, q85, {n_distinct} distinct), "
                        "scan-by-scan entropy decode + the ordinary stage 2 (BASELINE configs[4])",
            "entropy_bytes_per_image": int(sum(map(len, raws)) // n_distinct),
            "smaller_batches": smaller,
            "parity": f"bit-exact vs oracle (all {n_distinct} distinct files), every image's status ok" if ok else f"MISMATCH (files {bad[:8]})"}


def mixed_content_side(ctx, dev, torch, layout, headline_ms, n_images: int = 1024, n_distinct: int = 256):
    """The headline workload on files of MIXED content (tools/jpegenc.c mjenc_synth_mixed_batch: quality 50..95, noise 0..80
    above / below a random split row: restart segments differ several-fold in bits): one lane walks one restart segment, so
    the stage-1 launch lasts as long as the longest segment's walk.  Parity of the smallest and the largest file vs the oracle."""
    import numpy as np
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    blob, offs = synth.synth_mixed_batch(n_distinct, 900000, W, H, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n_distinct)]
    files = [raws[i % n_distinct] for i in range(n_images)]
    prep = prepare_batch(files, layout, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n_images})
    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        dt = timed_executes(torch, plan, stream, d_rgb.data_ptr(), 20)
        s1, s2 = plan.time_stages(5, d_rgb.data_ptr())
        fused = bool(plan.stage1_form() & B.MJ_FORM_FUSED)
        front, main = plan.time_execute(5, d_rgb.data_ptr()) if fused else (None, None)
        plan.execute(stream, d_rgb.data_ptr())
        torch.cuda.synchronize()
        per = W * H * 3
        sizes = [len(r) for r in raws]
        bad = images_not_matching_the_oracle(raws, lambda i: d_rgb[i * per:(i + 1) * per].cpu().numpy(),
                                             "xmajor" if layout == B.MJ_LAYOUT_XMAJOR else "rowmajor")
        imgs = d_rgb[:n_images * per].view(n_images, per)
        same = all(bool(torch.equal(imgs[k * n_distinct:min(n_images, (k + 1) * n_distinct)], imgs[:min(n_distinct, n_images - k * n_distinct)]))
                   for k in range(1, (n_images + n_distinct - 1) // n_distinct))
        ok = not plan.read(rgb=False)["status"].any() and not bad and same
        seg_len = np.asarray(prep.seg_end, dtype=np.int64) - np.asarray(prep.seg_begin, dtype=np.int64)
    finally:
        plan.close()
    return {"value": round(n_images * W * H / 1e6 / dt, 1), "unit": "MP/s", "ms_per_step": round(dt * 1e3, 3),
            "stage01_ms": round(s1, 3), "stage2_ms": round(s2, 3), "vs_headline_ms_per_step": round(dt * 1e3 / headline_ms, 3),
            "fused_launch": fused, "stage0_ms": None if front is None else round(front, 3), "fused_ms": None if main is None else round(main, 3),
            "workload": f"{n_images} x 1920x1080 4:2:0 baseline JPEG, DRI=120, {n_distinct} distinct files of mixed content "
                        "(quality 50..95, noise sigma 0..80 above / below a random split row)",
            "file_bytes": {"min": int(min(sizes)), "mean": int(sum(sizes) // len(sizes)), "max": int(max(sizes))},
            "restart_segment_bytes": {"mean": int(seg_len.mean()), "max": int(seg_len.max())},
            "parity": (f"bit-exact vs oracle: all {n_distinct} distinct files, every replica identical to its first instance, every image's status ok"
                       if ok else f"MISMATCH (files {bad[:8]})"),
            "note": "stage 1 = one restart segment per lane: its launch lasts as long as the longest segment's serial walk, whatever "
                    "the others hold; segments are dealt out by length so that long ones sit in different waves (DESIGN.md section 3); "
                    "the step is ONE fused launch whose consumers take jobs from one pool across workgroups (stage01_ms / stage2_ms: the "
                    "same plan's stages launched separately)"}


def idct_only_side(ctx, dev, torch, n_images: int = 256):
    """BASELINE configs[1]: 256 synthetic 512x512 4:2:0 files, entropy decode on the HOST (here: the oracle's, outside any timing),
    stage 2 alone on the GPU through the plan path of mj_idct_batch (mj_plan_write_coef + mj_plan_execute_stage2); all 256
    images against the oracle."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    w = h = 512
    blob, offs = synth.synth_batch(n_images, 300000, w, h, 85, "420", 0)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n_images)]
    oracle.decode(raws[0])
    with ThreadPoolExecutor(max_workers=max(1, min(os.cpu_count() or 1, 32))) as ex:
        dec = list(ex.map(oracle.decode, raws))
    coef = np.concatenate([d["coef"] for d in dec])
    prep = prepare_batch(raws, B.MJ_LAYOUT_XMAJOR, 0)
    bc = prep.to_c()
    bc.blob, bc.blob_mem = None, B.MJ_MEM_NONE
    plan = B.Plan(ctx, bc, {"prep": prep, "n_images": n_images})
    try:
        plan.write_coef(coef)
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        for _ in range(30):
            plan.execute_stage2(0, d_rgb.data_ptr())
        plan.sync()
        _, s2 = plan.time_stages(50, d_rgb.data_ptr())
        per = w * h * 3
        host = d_rgb.cpu().numpy()
        ok = all(np.array_equal(host[i * per:(i + 1) * per].reshape(w, h, 3), dec[i]["rgb"]) for i in range(n_images))
    finally:
        plan.close()
    nbytes = n_images * ((w // 16) * (h // 16) * 6 * 128 + per)          # 1 572 864 B per image
    gbs = nbytes / (s2 * 1e-3) / 1e9
    return {"value": round(n_images * w * h / 1e6 / (s2 * 1e-3), 1), "unit": "MP/s", "stage2_ms": round(s2, 4),
            "workload": f"{n_images} x 512x512 4:2:0 baseline JPEG, q85, Huffman decode on the host, dequant+IDCT+upsample+RGB on the GPU (BASELINE configs[1])",
            "roofline": {"kernel": "k_reconstruct_fast (stage 2 alone)", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(s2, 4),
                         "note": "128 B/block read + 3 B/pixel written = 1 572 864 B per 512x512 image; 0.4 GB per launch: a batch this small "
                                 "ends before the chip's 3 072 wavefronts have each had two jobs, and its coefficients fit the 256 MB Infinity Cache"},
            "parity": f"bit-exact vs oracle (all {n_images} images)" if ok else "MISMATCH"}


def gpu_segmented_side(ctx, dev, torch, files, layout, d_ref):
    """The headline step with the restart markers found ON THE GPU (k_scan_markers inside the timed step: the host reads headers
    only — BatchDecoder's default route); output compared with the headline plan's on the device."""
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd import parse_jpeg
    from pyjpegdecoder_amd.batch import prepare_batch
    t0 = time.perf_counter()
    prep = prepare_batch(files, layout, 0, [parse_jpeg(f, headers_only=True) for f in files])
    host_s = time.perf_counter() - t0
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(files)})
    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        dt = timed_executes(torch, plan, stream, d_rgb.data_ptr(), 60)
        front, main = plan.time_execute(5, d_rgb.data_ptr())
        ok = not plan.read(rgb=False)["status"].any() and bool(torch.equal(d_rgb, d_ref))
        form = plan.stage1_form()
    finally:
        plan.close()
    n = len(files)
    return {"value": round(n * W * H / 1e6 / dt, 1), "unit": "MP/s", "ms_per_step": round(dt * 1e3, 3),
            "front_ms": round(front, 4), "main_ms": round(main, 4), "fused_launch": bool(form & B.MJ_FORM_FUSED),
            "host_headers_only_parse_assemble_s": round(host_s, 2),
            "parity": "identical to the headline plan's output (device-side compare), every image's status ok" if ok else "MISMATCH",
            "note": "front = k_scan_markers + k_destuff (or stage 0+1 when the plan is not fused), main = the fused launch (or stage 2)"}


def single_file_latency_side(raw, tmpdir):
    """The reference's own call: JpegDecoder(path) on ONE 1080p file (jpeg_decoder.py:56-110) — parse, plan, decode, pixels on
    the host; median of 7 after a warm-up."""
    import numpy as np
    from oracle import oracle
    from pyjpegdecoder_amd import JpegDecoder
    path = Path(tmpdir) / "one.jpg"
    path.write_bytes(raw)
    JpegDecoder(path)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        d = JpegDecoder(path)
        ts.append(time.perf_counter() - t0)
    ok = bool(np.array_equal(d.image_array, oracle.decode(raw)["rgb"]))
    med = sorted(ts)[len(ts) // 2]
    return {"value": round(med * 1e3, 2), "unit": "ms", "higher_is_better": False, "min_ms": round(min(ts) * 1e3, 2),
            "workload": "JpegDecoder(Path) on one 1920x1080 4:2:0 baseline file with DRI=120: file read, marker loop, plan, GPU decode, pixels to the host",
            "parity": "bit-exact vs oracle" if ok else "MISMATCH"}


def visible_gpus() -> int:
    """GPUs of this node, counted without initialising HIP (the launcher must stay a process that never touched the GPU): the
    KFD topology lists one node per agent, CPUs with simd_count 0.  Honours a ROCR/HIP_VISIBLE_DEVICES list.  -1 = unknown."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    base = Path("/sys/class/kfd/kfd/topology/nodes")
    if not base.is_dir():
        return -1
    n = 0
    for node in base.iterdir():
        try:
            props = dict(line.split(None, 1) for line in (node / "properties").read_text().splitlines() if " " in line)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            return -1
    return n


def launch_ranks(n_ranks: int, argv, share_gpu: bool, script=None) -> int:
    """`python bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <same arguments>` as a child (c10d rendezvous on 127.0.0.1, port chosen by the store itself), pass
    its stdout — rank 0's one JSON line — through, return its exit code."""
    import subprocess
    import uuid
    have = visible_gpus()
    if not share_gpu and 0 <= have < n_ranks:
        print(f"bench.py: --gpus {n_ranks} but this node shows {have} GPU(s); --share-gpu runs every rank on cuda:0 (self-test)",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the rendezvous store binds port 0 itself (c10d backend) and the agent hands the ranks an address and a port of its own
    # choosing: no port is picked here by bind-and-close, which another process could take before the store binds it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id={uuid.uuid4()}", "--local-addr", "127.0.0.1",
           str(script or Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default: ~3 s of GPU time at the default workload)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU (weak-scaling workload)")
    ap.add_argument("--distinct", type=int, default=256, help="distinct synthetic images per GPU (tiled to the workload's size)")
    ap.add_argument("--total-images", type=int, default=0,
                    help="BASELINE configs[3]: ONE job of this many images sharded over the ranks, per-GPU image queue (strong scaling)")
    ap.add_argument("--queue-batch", type=int, default=1250,
                    help="images per plan in the per-GPU queue (--total-images): stage 1 lasts as long as one restart segment's walk "
                         "whatever the plan holds, up to ~1400 1080p images, so plans are as large as the share allows")
    ap.add_argument("--queue-depth", type=int, default=2, help="plans in flight at once in the per-GPU queue, one stream each")
    ap.add_argument("--queue-collect-first", action="store_true",
                    help="A/B: the queue collects a slot's previous plan before it creates the next one (the new plan then takes over its "
                         "buffers: a fixed pairing of coefficient store and output slot; measured 3 %% slower)")
    ap.add_argument("--layout", default="xmajor", choices=["xmajor", "rowmajor"])
    ap.add_argument("--segment", default="host", choices=["host", "gpu"],
                    help="who finds the restart markers: the host parser (default) or stage 0 on the GPU (then inside the timed step)")
    ap.add_argument("--restart-interval", type=int, default=120,
                    help="MCUs per restart segment of the synthetic files (120 = one MCU row = BASELINE configs[2]; 0 = no DRI: "
                         "one segment per image, decoded through the synchronisation passes)")
    ap.add_argument("--parity-images", type=int, default=0, help="distinct images held to the oracle after the timed region (0 = all of them)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline and the side measurements")
    ap.add_argument("--no-progressive", action="store_true")
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"],
                    help="process group for the barrier / MAX of timings (gloo: nothing of this job needs RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="self-test: every rank uses cuda:0")
    ap.add_argument("--tune-placement", type=int, default=4,
                    help="coefficient stores the plan tries before the timed region (mj_plan_tune_placement; 1 = keep what the allocator gave)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks.  It has not imported torch and has not
        # touched the GPU; the ranks are CHILD processes (no exec of a process that initialised HIP), and rank 0's JSON line is
        # this process's stdout
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.share_gpu))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's rank count is the one that runs",
              file=sys.stderr)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:        # every rank writes its own synthetic files with an OpenMP team: an equal share of the host's cores each
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))      # (torch.distributed.run presets 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch                                # before libmijpeg: one HIP runtime per process
    import torch.distributed as dist
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # the Gloo transport announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): stdout carries ONE JSON line
        # and nothing else, so file descriptor 1 points at stderr while the process group comes up (and whenever it talks)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from pyjpegdecoder_amd.queue import DeviceImageQueue
    from pyjpegdecoder_amd.sharding import max_over_ranks, shard, sum_over_ranks
    from tools import synth

    layout = B.MJ_LAYOUT_XMAJOR if args.layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    queue_mode = args.total_images > 0
    if queue_mode:
        lo, hi = shard(args.total_images, rank, world)       # this rank's images of the job
        n_mine = hi - lo
    else:
        n_mine = args.batch

    # ---- synthetic inputs (host): the SURVEY §8d family, q85, 4:2:0, DRI = one MCU row --------------------
    distinct = max(1, min(args.distinct, n_mine))
    t0 = time.perf_counter()
    blob, offs = synth.synth_batch(distinct, 100000 * rank, W, H, 85, "420", args.restart_interval)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    gen_s = time.perf_counter() - t0
    files = [raws[i % distinct] for i in range(n_mine)]

    ctx = B.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    placement = None
    t0 = time.perf_counter()
    if queue_mode:
        # (across_passes: a pass — the rank's share of one job — hands over to the next without draining the GPU: the next
        # pass's plan is created while this pass's kernels run, as the next job's would be in a service; the fence drains)
        queue = DeviceImageQueue(ctx, files, args.queue_batch, layout, args.queue_depth, device=local_rank, across_passes=True,
                                 tune_placement=args.tune_placement, collect_first=args.queue_collect_first)
        torch.cuda.synchronize()
        host_prep_s, h2d_s = time.perf_counter() - t0, None

        def step():
            queue.run(wait=False)
        plan = None
    else:
        # host side of the path: header parse + restart segmentation (Python, not timed as "step")
        if args.segment == "gpu":       # headers only; parse each distinct file once, as a caller with real files would each file
            from pyjpegdecoder_amd import parse_jpeg
            prep = prepare_batch(files, layout, 0, [parse_jpeg(f, headers_only=True) for f in files])
        else:
            prep = prepare_batch(files, layout, 0)
        host_prep_s = time.perf_counter() - t0
        # device residency: torch owns the HBM buffers, libmijpeg gets raw pointers
        t0 = time.perf_counter()
        d_blob = torch.from_numpy(prep.blob).to(dev)
        torch.cuda.synchronize()
        h2d_s = time.perf_counter() - t0
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)

        def step():
            plan.execute(stream, d_rgb.data_ptr())
        # Placement (round 6, profiles/r06_placement.txt): where the plan's coefficient store lies relative to the output buffer puts
        # the fused launch into one of two classes 8-9 % apart, by the luck of two allocations.  A plan that is executed many times
        # into one buffer — this loop; a service's output slot — tries a few stores and keeps the fastest (mj_plan_tune_placement:
        # a few timed executes each, outside the timed region, after a second of warm-up so that no candidate is timed on a cold clock).
        if args.tune_placement > 1 and plan.stage1_form() & B.MJ_FORM_FUSED:
            from pyjpegdecoder_amd.placement import tuned_output
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 1.0:
                step()
                torch.cuda.synchronize()
            keep = [d_rgb]

            def alloc(nbytes):          # (the buffer that exists first; then others — all held until the choice is made, or torch hands the same block out again)
                t = keep.pop() if keep else torch.empty(nbytes, dtype=torch.uint8, device=dev)
                return t, t.data_ptr()
            d_rgb, _, placement = tuned_output(plan, stream, plan.info.rgb_bytes, alloc, out_candidates=3, store_candidates=args.tune_placement)
            placement["note"] = ("pyjpegdecoder_amd.placement.tuned_output before the warm-up: output buffers tried, for each the coefficient stores "
                                 "(and stage-0 stream buffers) the plan tried against it (mj_plan_tune_placement: ms per execute = stage 0 + fused "
                                 "launch, HIP events), and which pair stayed")

    if queue_mode and args.tune_placement > 1:      # (a second of passes first: no candidate is timed on a cold clock)
        qt = DeviceImageQueue(ctx, files[:min(len(files), args.queue_batch)], args.queue_batch, layout, 1, device=local_rank)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 1.0:
            qt.run()
        del qt

    def fence():
        if queue_mode:
            queue.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, dev if (world > 1 and args.backend == "nccl") else None)
    images_all_ranks = int(sum_over_ranks(float(n_mine), dev if (world > 1 and args.backend == "nccl") else None))

    # ---- per-launch times (HIP events on the launch stream) for the roofline objects, taken NOW: the parity check below leaves
    # the GPU idle for seconds, and launches timed in the first second after such a pause came out 2-4x too long (clocks)
    stage_times = None
    if rank == 0 and not queue_mode:
        plan.time_stages(2, d_rgb.data_ptr())           # (the stages' own kernels have not run yet when the step is one fused launch)
        launch_clock = None
        if plan.stage1_form() & B.MJ_FORM_FUSED:
            front_ms, main_ms = plan.time_execute(10, d_rgb.data_ptr())
            try:                                        # the shader clock the chip held during the last of those launches (power-limited)
                launch_clock = ctx.launch_clock()
            except Exception:
                pass
        else:
            front_ms = main_ms = None
        stage_times = (front_ms, main_ms) + tuple(plan.time_stages(10, d_rgb.data_ptr()))
        step()                                          # (the headline's own launch last: what the parity check looks at)
        torch.cuda.synchronize()

    # ---- parity of what was just timed, outside the timed region.  One plan: EVERY distinct image of the rank's batch against
    # the oracle (a pool of host threads), every replica against its first instance on the device, every image's status.
    # Queue mode: every distinct image of rank 0's first and last plan against the oracle, every plan's statuses. -------------
    parity = "unchecked"
    per = W * H * 3

    if queue_mode:
        status_bad = queue.bad
        if rank == 0:
            # rank 0's first and last plan once more, every distinct image of each against the oracle (the timed passes' statuses
            # are in status_bad); a plan's images are the share's files in order
            ok = status_bad == 0
            checked, bad_q = 0, []
            for kb in sorted({0, len(queue.batches) - 1}):
                queue.run(first=kb, count=1)
                out_t = queue.out_tensor(queue.slot_of(kb))
                base = kb * args.queue_batch
                idx = sorted({(base + i) % distinct: i for i in range(queue.batches[kb][3])}.values())     # one position per distinct file
                bad_k = images_not_matching_the_oracle([files[base + i] for i in idx],
                                                       lambda t, idx=idx, out_t=out_t: out_t[idx[t] * per:(idx[t] + 1) * per].cpu().numpy(), args.layout)
                checked += len(idx)
                bad_q += [base + idx[t] for t in bad_k]
            ok = ok and not bad_q and queue.bad == status_bad
            parity = (f"bit-exact vs oracle: every distinct image of rank 0's first and last plan ({checked} images), every plan's statuses ok"
                      if ok else f"MISMATCH (images {bad_q[:8]}, plans with a status not ok: {queue.bad})")
    else:
        out = plan.read(rgb=False)
        status_bad = int(np.count_nonzero(out["status"]))
        if rank == 0:
            n_chk = distinct if args.parity_images <= 0 else min(distinct, args.parity_images)
            ok = status_bad == 0
            imgs = d_rgb[:args.batch * per].view(args.batch, per)
            for k in range(1, (args.batch + distinct - 1) // distinct):          # replicas of the distinct files: on the device
                m = min(distinct, args.batch - k * distinct)
                ok = ok and bool(torch.equal(imgs[k * distinct:k * distinct + m], imgs[:m]))
            t0 = time.perf_counter()
            bad_imgs = images_not_matching_the_oracle(raws[:n_chk], lambda i: imgs[i].cpu().numpy(), args.layout)
            ok = ok and not bad_imgs
            parity = (f"bit-exact vs oracle: all {n_chk} distinct images of the batch ({time.perf_counter() - t0:.1f} s of oracle time on host threads), "
                      f"every replica identical to its first instance (device-side compare), every image's status ok") if ok else \
                     f"MISMATCH (images {sorted(bad_imgs)[:8]}, statuses not ok: {status_bad})"

    base_cfg = "BASELINE configs[3]" if queue_mode else "BASELINE configs[2]"
    std = args.restart_interval == 120 and (queue_mode or args.batch == 1024)
    if queue_mode:
        workload = (f"one job of {args.total_images} x 1920x1080 4:2:0 baseline JPEG, q85, DRI={args.restart_interval}, sharded over {world} GPU(s) "
                    f"({n_mine} images on rank 0), per-GPU image queue of {args.queue_batch}-image plans ({args.queue_depth} in flight) over HBM-resident files, "
                    f"on-GPU Huffman + dequant/IDCT/upsample/RGB" + (f" ({base_cfg})" if std and args.total_images == 10000 else " (NOT a BASELINE configuration)"))
    else:
        workload = (f"{args.batch} x 1920x1080 4:2:0 baseline JPEG per GPU, q85, " +
                    ("DRI=120 (one MCU row, 68 segments/image)" if args.restart_interval == 120 else f"restart interval {args.restart_interval}") +
                    ", on-GPU Huffman + dequant/IDCT/upsample/RGB" + (f" ({base_cfg})" if std else " (NOT the BASELINE configuration)"))

    line = None
    if rank == 0:
        mp_total = images_all_ranks * W * H / 1e6
        line = {
            "metric": "megapixels/sec decoded (1080p 4:2:0 baseline batch)",
            "value": round(mp_total * args.steps / dt, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if queue_mode else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "images_per_gpu": n_mine, "images_all_gpus": images_all_ranks,
                       "distinct_images_per_gpu": distinct, "layout": args.layout,
                       "restart_segmentation": "host" if queue_mode else args.segment,
                       "entropy_bytes_per_image": int(sum(map(len, raws)) // distinct),
                       "parallelism": f"image-sharded x{world}, no collective on the data path (gloo barrier + MAX of timings only)"},
            "timed_region_s": round(dt, 3),
            "parity": parity,
            "placement": placement if not queue_mode else {f"slot {k}": v for k, v in queue.placement.items()},
            "host": {"synth_encode_s": round(gen_s, 2), "parse_segment_assemble_s": round(host_prep_s, 2),
                     "h2d_blob_s": None if h2d_s is None else round(h2d_s, 3),
                     "note": "outside the timed region; inputs are HBM-resident when timing starts"},
        }

    # ---- roofline: HIP events on the launch stream, algorithmic bytes per launch ---------------------------
    # A plan whose execute is ONE fused launch (stage 1's lane walk and stage 2's strip worker side by side, fused.hip):
    # `roofline` is that launch — SURVEY 8d's end-to-end figure, E + 2 x 128 B/block + 3 B/pixel: the coefficients still go
    # through memory, written by the walk and read back by the same CU —, `roofline_other_stage` what runs in front of it
    # (stage 0), and `two_launches` the same plan's stages launched separately (mj_plan_execute_stage1 / _stage2), each
    # with its own roofline: stage 2 alone is the kernel BASELINE.json's 40 % target is about.
    if rank == 0 and not queue_mode:
        fused = bool(plan.stage1_form() & B.MJ_FORM_FUSED)
        front_ms, main_ms, s1_ms, s2_ms = stage_times
        copy_gbs = None
        try:            # second denominator SURVEY 8d asks for: what a plain device-to-device copy achieves here (16 B per lane, util_kernels.hip)
            copy_gbs = round(ctx.copy_rate_gbs(1 << 31, 5), 1)
        except Exception:
            pass
        ent_bytes = plan.info.entropy_bytes
        s1_bytes = ent_bytes + args.batch * BLOCKS_PER_IMAGE * 128
        s2_bytes = args.batch * STAGE2_BYTES_PER_IMAGE
        # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the bench): quoted
        # only for the workload that was profiled AND for the kernel sources the passes were taken from
        traffic, tnote = {}, None
        tfiles = sorted((ROOT / "profiles").glob("r*_hbm_traffic_batch1024.json"))
        tfile = tfiles[-1] if tfiles else None
        if args.batch == 1024 and args.layout == "xmajor" and args.restart_interval == 120 and tfile is not None:
            td = json.loads(tfile.read_text())
            if td.get("sources_sha16") == sources_sha16():
                for k, d in td["kernels"].items():
                    if "traffic_bytes_per_launch" in d:
                        if "fused" in k:
                            traffic["fused"] = int(d["traffic_bytes_per_launch"])
                        elif "destuff" in k or "scan_markers" in k:
                            traffic["stage0"] = traffic.get("stage0", 0) + int(d["traffic_bytes_per_launch"])
                            traffic["stage1"] = traffic.get("stage1", 0) + int(d["traffic_bytes_per_launch"])
                        elif "huffman" in k or "sync" in k or "vsegs" in k:
                            traffic["stage1"] = traffic.get("stage1", 0) + int(d["traffic_bytes_per_launch"])
                        elif "reconstruct" in k:
                            traffic["stage2"] = int(d["traffic_bytes_per_launch"])
            else:
                tnote = f"profiles/{tfile.name} was taken from other kernel sources ({td.get('sources_sha16')} != {sources_sha16()}): not quoted"

        def roof(name, nbytes, ms, note, tkey):
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic.get(tkey),
                    "traffic_source": (f"profiles/{tfile.name} (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes)" if tkey in traffic else tnote),
                    "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(ms, 4),
                    "device_copy_gbs": copy_gbs, "frac_of_device_copy": round(gbs / copy_gbs, 4) if copy_gbs else None,
                    "note": note}

        form = {B.MJ_FORM_WAVE: "k_huffman", B.MJ_FORM_LANES: "k_destuff + k_huffman_lanes13",
                B.MJ_FORM_SYNC: "k_destuff + k_sync_count + k_build_vsegs + k_huffman_lanes13"}.get(plan.stage1_form() & 15, "stage 1")
        r1 = roof(f"{form} (stage 0+1: byte-drop pass + Huffman decode; k_scan_markers too with --segment gpu)", s1_bytes, s1_ms,
                  "entropy bytes read + 128 B/block coefficients written; serial-decode (instruction issue) bound, quoted against HBM as SURVEY §8d asks",
                  "stage1")
        r2 = roof("k_reconstruct_fast (stage 2: dequant+IDCT+upsample+colour)", s2_bytes, s2_ms,
                  "128 B/block read + 3 B/pixel written = 12 487 680 B per 1080p image", "stage2")
        if fused:
            rf = roof("k_fused (ONE launch: stage 1's lane walk on 8 wavefronts per CU + stage 2's strip worker on the other 8, then on all 16; fused.hip)",
                      s1_bytes + s2_bytes, main_ms,
                      "SURVEY 8d's end-to-end figure: entropy bytes read + 128 B/block coefficients written and read back (a CU holds 272 "
                      "restart segments in lock-step: their MCUs do not fit its LDS, so the blocks go through the coefficient store and come "
                      "back through the same CU's L2 while the walk lasts) + 3 B/pixel written", "fused")
            r0 = roof("k_destuff (stage 0: the bit reader's byte rules" + (", behind k_scan_markers" if args.segment == "gpu" else "") + ")",
                      2 * ent_bytes, front_ms, "entropy-coded bytes read and the kept bytes written", "stage0")
            if launch_clock and launch_clock[0] > 0:
                rf["shader_clock_mhz"] = round(launch_clock[0], 0)
                rf["launch_ms_seen_by_workgroup_0"] = round(launch_clock[1], 4)
                rf["clock_note"] = ("shader-clock counter / 100 MHz counter between the start and the end of the launch's workgroup 0, in this "
                                    "process, for the last of the launches avg_launch_ms averages: the launch is power-limited, its time follows this clock")
            line["roofline"], line["roofline_other_stage"] = rf, r0
            # the same plan's stages as separate launches, timed the same way (HIP events, 10 launches each)
            line["two_launches"] = {"stage01_ms": round(s1_ms, 4), "stage2_ms": round(s2_ms, 4), "sum_ms": round(s1_ms + s2_ms, 4),
                                    "fused_ms": round(front_ms + main_ms, 4),
                                    "roofline_stage2": r2, "roofline_stage01": r1,
                                    "note": "mj_plan_execute_stage1 + _stage2 of the plan the headline ran as one launch; stage 2 alone is the "
                                            "batched dequant+IDCT kernel of BASELINE.json's >= 40 % of HBM target"}
        else:
            line["roofline"], line["roofline_other_stage"] = (r1, r2) if s1_ms >= s2_ms else (r2, r1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(raws[:64])
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not queue_mode and args.restart_interval == 120:
        if plan is not None:
            try:
                line["pipelined"] = pipelined_side(ctx, dev, torch, prep, d_blob, plan, d_rgb)
            except Exception as exc:
                line["pipelined"] = {"error": repr(exc)}
        if plan is not None and args.segment == "host":
            try:
                line["gpu_segmented"] = gpu_segmented_side(ctx, dev, torch, files, layout, d_rgb)
            except Exception as exc:
                line["gpu_segmented"] = {"error": repr(exc)}
            try:                                 # what the fp32 first level of stage 2 buys: the same batch through the exact-order kernel alone
                prep_x = prepare_batch(files, layout, B.MJ_FLAG_EXACT_ONLY)
                plan_x = B.Plan(ctx, prep_x.to_c(d_blob.data_ptr()), {"prep": prep_x, "n_images": args.batch})
                d_rgb_x = torch.empty(plan_x.info.rgb_bytes, dtype=torch.uint8, device=dev)
                plan_x.execute(stream, d_rgb_x.data_ptr())
                plan_x.sync()
                _, s2x = plan_x.time_stages(2, d_rgb_x.data_ptr())
                same = bool(torch.equal(d_rgb_x, d_rgb))
                plan_x.close()
                del d_rgb_x
                line["stage2_exact_only_ms"] = {"value": round(s2x, 3), "unit": "ms", "identical_output": same,
                                                "note": "k_reconstruct (the reference's float64 summation order for every block, MJ_FLAG_EXACT_ONLY) on the headline batch; "
                                                        "the headline's stage 2 is the fp32 first level with fp64 and exact-order levels behind it"}
            except Exception as exc:
                line["stage2_exact_only_ms"] = {"error": repr(exc)}
        if plan is not None:
            plan.close()                         # (its 6.4 GB coefficient store goes back to the context before the other batches come)
        try:
            line["idct_only"] = idct_only_side(ctx, dev, torch)
        except Exception as exc:
            line["idct_only"] = {"error": repr(exc)}
        try:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                line["single_file_latency"] = single_file_latency_side(raws[0], td)
        except Exception as exc:
            line["single_file_latency"] = {"error": repr(exc)}
        try:
            line["mixed_content"] = mixed_content_side(ctx, dev, torch, layout, line["ms_per_step"])
        except Exception as exc:
            line["mixed_content"] = {"error": repr(exc)}
        if not args.no_progressive:
            try:
                line["progressive"] = progressive_side(ctx, dev, torch)
            except Exception as exc:
                line["progressive"] = {"error": repr(exc)}
        # beside the headline: the same images written WITHOUT restart markers (one serial bitstream per image, the
        # usual case in the wild), decoded through the synchronisation passes; not BASELINE's configuration
        try:
            from oracle import oracle
            nb, nd = 256, 32
            blob2, offs2 = synth.synth_batch(nd, 700000, W, H, 85, "420", 0)
            raws2 = [blob2[int(offs2[i]):int(offs2[i + 1])].tobytes() for i in range(nd)]
            prep2 = prepare_batch([raws2[i % nd] for i in range(nb)], layout, 0)
            d_blob2 = torch.from_numpy(prep2.blob).to(dev)
            plan2 = B.Plan(ctx, prep2.to_c(d_blob2.data_ptr()), {"prep": prep2, "n_images": nb})
            d_rgb2 = d_rgb[:plan2.info.rgb_bytes]
            dt2 = timed_executes(torch, plan2, stream, d_rgb2.data_ptr(), 10, warm_s=0.5)
            st2 = plan2.read(rgb=False)["status"]
            bad2 = images_not_matching_the_oracle(raws2, lambda i: d_rgb2[i * W * H * 3:(i + 1) * W * H * 3].cpu().numpy(), args.layout)
            imgs2 = d_rgb2[:nb * W * H * 3].view(nb, W * H * 3)
            same2 = all(bool(torch.equal(imgs2[k * nd:(k + 1) * nd], imgs2[:nd])) for k in range(1, nb // nd))
            ok2 = not st2.any() and not bad2 and same2
            line["without_restart_markers"] = {"value": round(nb * W * H / 1e6 / dt2, 1), "unit": "MP/s", "ms_per_step": round(dt2 * 1e3, 3),
                                               "workload": f"{nb} x 1920x1080 4:2:0 baseline JPEG, q85, no DRI (one segment per image)",
                                               "images_not_ok": int((st2 != 0).sum()),
                                               "parity": (f"bit-exact vs oracle: all {nd} distinct files, every replica identical to its first instance, every image's status ok"
                                                          if ok2 else f"MISMATCH (files {bad2[:8]})")}
            plan2.close()
        except Exception as exc:
            line["without_restart_markers"] = {"error": repr(exc)}
        # and the PCIe-inclusive rate of the public API (never `value`): file bytes in host memory -> pixels in HBM,
        # batches back to back (BatchDecoder.decode_device_iter: the next batch's header parse + upload run under the
        # current batch's kernels), restart markers found on the GPU
        try:
            from pyjpegdecoder_amd import BatchDecoder
            nb = 512
            files3 = [raws[i % len(raws)] for i in range(nb)]
            bd = BatchDecoder(dev.index or 0, layout=args.layout, segment="gpu")
            for _ in bd.decode_device_iter(files3 for _ in range(3)):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_it = 16
            for out3 in bd.decode_device_iter(files3 for _ in range(n_it)):
                pass
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t0) / n_it
            del out3
            bd.close()
            line["host_bytes_to_device_pixels"] = {"value": round(nb * W * H / 1e6 / dt3, 1), "unit": "MP/s", "ms_per_batch": round(dt3 * 1e3, 3),
                                                   "workload": f"{n_it} batches of {nb} of the files above through BatchDecoder(segment='gpu').decode_device_iter",
                                                   "note": "includes header parse, batch assembly, H2D of the files and plan creation; PCIe-inclusive, not `value`"}
        except Exception as exc:
            line["host_bytes_to_device_pixels"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(line), flush=True)

    if plan is not None:
        plan.close()          # (idempotent)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
