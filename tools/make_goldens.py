#!/usr/bin/env python3
"""Generate tests/golden/ by RUNNING THE REFERENCE in this container.

The reference (/root/reference/jpeg_decoder.py, pure Python) is imported here — and only here — to
capture the seam arrays of SURVEY.md §8c.  Nothing of its source is copied: the outputs are data
(inputs + expected outputs).  The reference cannot travel to the GPU box, the vectors do.

Captured per JPEG fixture (module globals are resolved at call time, so wrapping them is enough):
  G1  coef   int16 [nblk,64]   zig-zag coefficients seen by undo_zigzag at :869
  G2  deq    int16 [nblk,8,8]  dequantised block passed to InverseDCT.__call__ (:872)
  G3  idct   int16 [nblk,8,8]  its return value
  G5  planes int16 (W,H,C)     cropped image_array before colour conversion (:1373)
  G6  rgb    uint8 (W,H,3)|(W,H)  final image_array
plus the attribute surface left on the object.

Also: the IDCT table, the upsample operators W (ResizeGrid via griddata, F5), and known-answer
vectors for InverseDCT (incl. the exact-tie blocks of F6), ResizeGrid and YCbCr_to_RGB.

Usage:  python tools/make_goldens.py [--big]      (--big adds the 1080p DRI fixture, ~1 min)
"""
import argparse
import contextlib
import hashlib
import io
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")

with contextlib.redirect_stdout(io.StringIO()):
    import jpeg_decoder as jd  # the reference

from tools import synth  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference")
FILES = GOLD / "files"

jd.JpegDecoder.show = lambda self: None  # never open a viewer


def run_reference(path: Path):
    """Decode `path` with the reference, capturing the seams."""
    cap = {"coef": [], "deq": [], "idct": [], "planes": None}
    state = {"in_dqt": False}
    o_undo, o_idct, o_dqt, o_eoi = jd.undo_zigzag, jd.InverseDCT.__call__, \
        jd.JpegDecoder.define_quantization_table, jd.JpegDecoder.end_of_image

    def w_undo(block):
        if not state["in_dqt"]:
            cap["coef"].append(np.array(block, dtype=np.int16))
        return o_undo(block)

    def w_idct(self, block):
        cap["deq"].append(np.array(block, dtype=np.int16))
        out = o_idct(self, block)
        cap["idct"].append(np.array(out, dtype=np.int16))
        return out

    def w_dqt(self, data):
        state["in_dqt"] = True
        try:
            return o_dqt(self, data)
        finally:
            state["in_dqt"] = False

    def w_eoi(self, data):
        cap["planes"] = np.array(self.image_array[0:self.image_width, 0:self.image_height, :], dtype=np.int16)
        return o_eoi(self, data)

    jd.undo_zigzag, jd.InverseDCT.__call__ = w_undo, w_idct
    jd.JpegDecoder.define_quantization_table, jd.JpegDecoder.end_of_image = w_dqt, w_eoi
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            dec = jd.JpegDecoder(path)
    finally:
        jd.undo_zigzag, jd.InverseDCT.__call__ = o_undo, o_idct
        jd.JpegDecoder.define_quantization_table, jd.JpegDecoder.end_of_image = o_dqt, o_eoi
    return dec, cap


def run_reference_progressive(path: Path):
    """Progressive file: snapshot the coefficient state (image_array) before every scan and before the final
    IDCT pass (:1308-1317), plus the usual planes / rgb."""
    cap = {"before_scan": [], "pre_idct": None, "planes": None}
    holder = {}
    o_scan, o_init, o_eoi = jd.JpegDecoder.progressive_dct_scan, jd.InverseDCT.__init__ if hasattr(jd.InverseDCT, "__init__") else None, jd.JpegDecoder.end_of_image

    def w_scan(self, *a, **k):
        holder["dec"] = self
        cap["before_scan"].append(np.array(self.image_array, dtype=np.int16))
        return o_scan(self, *a, **k)

    class SnapIDCT(jd.InverseDCT):
        def __init__(self):
            if "dec" in holder and cap["pre_idct"] is None:
                cap["pre_idct"] = np.array(holder["dec"].image_array, dtype=np.int16)

    def w_eoi(self, data):
        cap["planes"] = np.array(self.image_array[0:self.image_width, 0:self.image_height, :], dtype=np.int16)
        return o_eoi(self, data)

    o_cls = jd.InverseDCT
    jd.JpegDecoder.progressive_dct_scan, jd.InverseDCT, jd.JpegDecoder.end_of_image = w_scan, SnapIDCT, w_eoi
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            dec = jd.JpegDecoder(path)
    finally:
        jd.JpegDecoder.progressive_dct_scan, jd.InverseDCT, jd.JpegDecoder.end_of_image = o_scan, o_cls, o_eoi
    return dec, cap


def pixel_layout_to_blocks(arr: np.ndarray, dec) -> np.ndarray:
    """image_array holding coefficients at pixel coordinates (:1029, :1225) -> int16 [nblocks,64] zig-zag, blocks in
    the interleaved order (MCU raster, component, block_count) of the baseline seam."""
    comps = list(dec.color_components.values())
    nc = len(comps)
    hmax = max(c.horizontal_sampling for c in comps) if nc > 1 else 1
    vmax = max(c.vertical_sampling for c in comps) if nc > 1 else 1
    mw, mh = (8 * hmax, 8 * vmax) if nc > 1 else (8, 8)
    mx, my = -(-dec.image_width // mw), -(-dec.image_height // mh)
    zz = jd.zagzig
    blocks = []
    for m in range(mx * my):
        my_, mx_ = divmod(m, mx)
        for c in comps:
            h, v = (c.horizontal_sampling, c.vertical_sampling) if nc > 1 else (1, 1)
            for r in range(h * v):
                by, bx = divmod(r, h)
                X, Y = (mx_ * h + bx) * 8, (my_ * v + by) * 8
                blocks.append([arr[X + zz[k][0], Y + zz[k][1], c.order] for k in range(64)])
    return np.array(blocks, dtype=np.int16)


def attrs_of(dec) -> dict:
    return {
        "file_size": dec.file_size, "file_header": dec.file_header, "scan_finished": dec.scan_finished,
        "scan_mode": dec.scan_mode, "image_width": dec.image_width, "image_height": dec.image_height,
        "color_components": {str(k): list(v) for k, v in dec.color_components.items()},
        "sample_shape": list(dec.sample_shape),
        "huffman_tables": {str(k): {cw: int(val) for cw, val in t.items()} for k, t in dec.huffman_tables.items()},
        "quantization_tables": {str(k): np.asarray(v).tolist() for k, v in dec.quantization_tables.items()},
        "restart_interval": dec.restart_interval, "scan_count": dec.scan_count, "scan_amount": dec.scan_amount,
        "mcu_width": dec.mcu_width, "mcu_height": dec.mcu_height, "mcu_shape": list(dec.mcu_shape),
        "mcu_count_h": dec.mcu_count_h, "mcu_count_v": dec.mcu_count_v, "mcu_count": dec.mcu_count,
        "array_width": dec.array_width, "array_height": dec.array_height, "array_depth": dec.array_depth,
        "image_array_shape": list(dec.image_array.shape), "image_array_dtype": str(dec.image_array.dtype),
        "has_raw_file": hasattr(dec, "raw_file"),
    }


def pil_jpeg(rgb, **kw) -> bytes:
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


def file_fixtures(big: bool):
    """(name, bytes) of every JPEG fixture.  Content: SURVEY §8d family from tools/jpegenc.c."""
    S = synth.synth_rgb
    fx = []
    # config 1: 64x64 4:4:4 baseline (libjpeg/Pillow writer)
    fx.append(("c1_64x64_444_pil", pil_jpeg(S(1, 64, 64), quality=85, subsampling=0)))
    fx.append(("64x64_420_pil", pil_jpeg(S(2, 64, 64), quality=85, subsampling=2)))
    # non-multiple-of-MCU crop + optimised (non Annex-K) Huffman tables
    fx.append(("70x50_420_pil_opt", pil_jpeg(S(3, 70, 50), quality=75, subsampling=2, optimize=True)))
    fx.append(("64x48_422_pil", pil_jpeg(S(4, 64, 48), quality=85, subsampling=1)))
    # DRI: Pillow writer (restart every 2 MCU rows) and own writer (restart every 3 MCUs, partial last segment)
    fx.append(("128x64_420_pil_rst", pil_jpeg(S(5, 128, 64), quality=85, subsampling=2, restart_marker_rows=2)))
    fx.append(("128x64_420_dri3", synth.encode_rgb(S(6, 128, 64), 85, "420", 3)))
    fx.append(("100x36_420_dri7", synth.encode_rgb(S(7, 100, 36), 90, "420", 7)))
    fx.append(("64x64_grey_pil", pil_jpeg(S(8, 64, 64)[..., 1], quality=85)))
    fx.append(("50x70_grey_dri4", synth.encode_rgb(S(9, 50, 70), 80, "grey", 4)))
    fx.append(("48x80_440", synth.encode_rgb(S(10, 48, 80), 85, "440", 0)))
    fx.append(("72x40_422_dri2", synth.encode_rgb(S(11, 72, 40), 85, "422", 2)))
    fx.append(("40x40_444_dri5", synth.encode_rgb(S(12, 40, 40), 85, "444", 5)))
    # 4:1:1 (luma 4x1: MCU 32x8, chroma upsampled 8x8 -> 32x8): multiple of the MCU, and cropped with restart markers
    fx.append(("96x24_411", synth.encode_rgb(S(40, 96, 24), 85, "411", 0)))
    fx.append(("100x36_411_dri3", synth.encode_rgb(synth.synth_rgb(41, 100, 36, 25.0), 90, "411", 3)))
    # high quality + strong noise: long codes, many 0xFF stuffing bytes, large coefficient categories
    fx.append(("96x64_420_q100_noise", synth.encode_rgb(synth.synth_rgb(13, 96, 64, 60.0), 100, "420", 6)))
    fx.append(("64x64_444_q98_pil", pil_jpeg(synth.synth_rgb(14, 64, 64, 50.0), quality=98, subsampling=0)))
    # very smooth: mostly DC-only blocks (the exact-tie case F6 is frequent here)
    fx.append(("96x96_420_smooth", synth.encode_rgb(synth.synth_rgb(15, 96, 96, 0.0), 60, "420", 0)))
    flat = np.full((32, 48, 3), 77, dtype=np.uint8)
    flat[:, 24:] = (200, 30, 120)
    fx.append(("48x32_420_flat", synth.encode_rgb(flat, 50, "420", 0)))
    # non-interleaved baseline (one scan per component): plain, with DRI, non-multiple-of-8 size
    fx.append(("ni_40x24_444", synth.encode_rgb(S(30, 40, 24), 85, "444ni", 0)))
    fx.append(("ni_64x48_444_dri5", synth.encode_rgb(S(31, 64, 48), 90, "444ni", 5)))
    fx.append(("ni_37x29_444_dri4", synth.encode_rgb(synth.synth_rgb(32, 37, 29, 30.0), 95, "444ni", 4)))
    # config 2 shape (512x512 4:2:0, no DRI)
    fx.append(("c2_512x512_420", synth.encode_rgb(S(16, 512, 512), 85, "420", 0)))
    if big:
        fx.append(("c3_1920x1080_420_dri120", synth.encode_rgb(S(0, 1920, 1080), 85, "420", 120)))
    # progressive (SOF2): libjpeg's default 10-scan script with successive approximation, per-scan optimised tables
    fx.append(("prog_64x64_420_pil", pil_jpeg(S(20, 64, 64), quality=85, subsampling=2, progressive=True)))
    fx.append(("prog_64x64_444_pil", pil_jpeg(S(21, 64, 64), quality=90, subsampling=0, progressive=True)))
    fx.append(("prog_70x50_420_pil", pil_jpeg(S(22, 70, 50), quality=75, subsampling=2, progressive=True)))
    fx.append(("prog_64x48_422_pil", pil_jpeg(S(23, 64, 48), quality=85, subsampling=1, progressive=True)))
    fx.append(("prog_64x64_grey_pil", pil_jpeg(S(24, 64, 64)[..., 1], quality=85, progressive=True)))
    fx.append(("prog_96x80_420_noise_pil", pil_jpeg(synth.synth_rgb(25, 96, 80, 45.0), quality=95, subsampling=2, progressive=True)))
    fx.append(("prog_128x64_420_rst_pil", pil_jpeg(S(26, 128, 64), quality=85, subsampling=2, progressive=True, restart_marker_rows=1)))
    fx.append(("prog_48x32_420_flat_pil", pil_jpeg(flat, quality=50, subsampling=2, progressive=True)))
    return fx


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def capture_W(src, dst) -> np.ndarray:
    """Upsample operator of ResizeGrid for (src -> dst), as integer numerators over 15 (x2 along an axis: SURVEY F5) or
    over 31 (x4 along one axis, x1 along the other: the mesh points then lie on grid lines and two taps remain)."""
    den = 31 if max(dst[0] // src[0], dst[1] // src[1]) == 4 else 15
    r = jd.ResizeGrid()
    n = src[0] * src[1]
    r(np.zeros(src, dtype=np.int16), dst)           # fill the mesh caches exactly as the reference does
    new_xy, old_xy = r.mesh_cache[(src, dst)], r.indices_cache[src]
    W = np.zeros((dst[0] * dst[1], n))
    for k in range(n):
        b = np.zeros(n)
        b[k] = float(den) * 64
        W[:, k] = jd.griddata(old_xy, b, new_xy).ravel() / 64
    Wi = np.rint(W).astype(np.int8)
    assert np.abs(W - Wi).max() < 1e-9 and (Wi.sum(1) == den).all() and ((Wi != 0).sum(1) <= 3).all()
    return Wi


def example_known_answers():
    """The reference repository's own known answers: `progressive scan example/` holds a 4160x2340 4:2:0 progressive
    file whose restart interval is redefined between scans (4160 / 8320 MCUs, :474-503) and two PNGs of the padded
    image array as it stood after scan 1 and after scan 2.  The JPEG is copied as a data fixture; of each PNG the
    SHA-256 of the cropped image in the reference's [x, y, c] order and a strided sample are kept."""
    from PIL import Image
    src = REF / "progressive scan example"
    out = GOLD / "example"
    out.mkdir(parents=True, exist_ok=True)
    raw = (src / "base image.jpg").read_bytes()
    (out / "base_image.jpg").write_bytes(raw)
    sys.path.insert(0, str(ROOT))
    from pyjpegdecoder_amd import parse_jpeg
    p = parse_jpeg(raw)
    meta = {"source": "progressive scan example/base image.jpg", "width": p.image_width, "height": p.image_height,
            "scans": [{"components": s.component_ids, "ss": s.spectral_start, "se": s.spectral_end, "ah": s.bit_high,
                       "al": s.bit_low, "restart_interval": s.restart_interval, "entropy_end": int(s.entropy_end)} for s in p.scans],
            "after_scan": {}}
    samples = {}
    for k, name in ((1, "after scan 01.png"), (2, "after scan 02.png")):
        png = np.asarray(Image.open(src / name).convert("RGB"))             # (padded H, W, 3)
        xy = np.ascontiguousarray(np.swapaxes(png[:p.image_height, :p.image_width], 0, 1))
        meta["after_scan"][str(k)] = {"png": name, "png_shape": list(png.shape), "sha256_rgb_xmajor": sha(xy)}
        samples[f"after_scan_{k}"] = xy[::41, ::37]
    meta["sample_strides"] = [41, 37]
    np.savez_compressed(out / "samples.npz", **samples)
    (out / "known_answers.json").write_text(json.dumps(meta, indent=1))
    print("example known answers:", {k: v["sha256_rgb_xmajor"][:16] for k, v in meta["after_scan"].items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--example", action="store_true", help="only (re)generate tests/golden/example from the reference's example directory")
    args = ap.parse_args()
    if args.example:
        example_known_answers()
        return
    FILES.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(20261002)
    # upsample operators added after the first capture: made when missing, also with --only
    for src, dst in (((8, 8), (32, 8)),):
        f = GOLD / f"upsample_W_{src[0]}x{src[1]}_{dst[0]}x{dst[1]}.npy"
        if not f.exists():
            np.save(f, capture_W(src, dst))
            print("captured", f.name)

    if not args.only:
        # --- constant tables -------------------------------------------------------------------
        np.save(GOLD / "idct_table.npy", np.array(jd.InverseDCT.idct_table))
        shapes = [((8, 8), (16, 16)), ((8, 8), (16, 8)), ((8, 8), (8, 16))]
        for src, dst in shapes:
            np.save(GOLD / f"upsample_W_{src[0]}x{src[1]}_{dst[0]}x{dst[1]}.npy", capture_W(src, dst))

        # --- InverseDCT known answers ----------------------------------------------------------
        blocks = []
        for dcq in (4, -4, 12, -12, 20, 36, -20, 100, 1020, -1028, 8, 0, 2044, -2052):   # F6 ties (DC·q ≡ 4 mod 8)
            b = np.zeros((8, 8), dtype=np.int16); b[0, 0] = dcq; blocks.append(b)
        for pos in ((0, 4), (4, 0), (4, 4)):                         # rational table entries (±1/8)
            for val in (4, -4, 12, 36, -20, 8):
                b = np.zeros((8, 8), dtype=np.int16); b[pos] = val; blocks.append(b)
                b2 = b.copy(); b2[0, 0] = 4; blocks.append(b2)
                b3 = b.copy(); b3[0, 0] = -8; b3[4, 4] += 12; blocks.append(b3)
        for _ in range(400):                                         # only {0,4}x{0,4} entries: every sample is k/8
            b = np.zeros((8, 8), dtype=np.int16)
            for pos in ((0, 0), (0, 4), (4, 0), (4, 4)):
                if rng.random() < 0.7:
                    b[pos] = int(rng.integers(-60, 61)) * 4
            blocks.append(b)
        for _ in range(600):                                         # sparse, photo-like magnitudes
            b = np.zeros((8, 8), dtype=np.int16)
            nnz = int(rng.integers(1, 20))
            for _k in range(nnz):
                u, v = int(abs(rng.normal(0, 2.2))) % 8, int(abs(rng.normal(0, 2.2))) % 8
                b[u, v] = int(rng.integers(-40, 41)) * int(rng.integers(1, 30))
            blocks.append(b)
        for _ in range(200):                                         # dense
            blocks.append(rng.integers(-300, 301, (8, 8)).astype(np.int16))
        for _ in range(60):                                          # large magnitudes (still within int16 after IDCT)
            blocks.append(rng.integers(-2000, 2001, (8, 8)).astype(np.int16))
        blocks = np.stack(blocks)
        idct = jd.InverseDCT()
        outs = np.stack([idct(b) for b in blocks]).astype(np.int16)
        np.savez_compressed(GOLD / "idct_blocks.npz", blocks=blocks, out=outs)

        # --- ResizeGrid known answers -----------------------------------------------------------
        rz = jd.ResizeGrid()
        d = {}
        for src, dst in shapes:
            ins = rng.integers(-40, 300, (150,) + src).astype(np.int16)
            ins[:20] = rng.integers(-3000, 3000, (20,) + src)
            d[f"in_{dst[0]}x{dst[1]}"] = ins
            d[f"out_{dst[0]}x{dst[1]}"] = np.stack([rz(b, dst) for b in ins]).astype(np.int16)
        np.savez_compressed(GOLD / "resize_blocks.npz", **d)

        # --- YCbCr_to_RGB known answers (incl. the exact-tie lattices of DESIGN.md) --------------
        ycc = [rng.integers(-20, 280, (4000, 3)), rng.integers(-600, 900, (1500, 3))]
        t = []
        for Y in range(-5, 262):
            for cb in (3, 253, 128 + 375, 128 - 375):           # 1.772*(Cb-128) = ±221.5, ±664.5
                t.append((Y, cb, 128))
            for cr in (378, -122, 128 + 750):                    # 1.402*(Cr-128) = ±350.5, 1051.5
                t.append((Y - 350, 128, cr)); t.append((Y + 350, 128, cr))
        for cb in range(-130, 400):                              # G ties: 17207*cb' + 35707*cr' ≡ 25000 (mod 50000)
            cbp = cb - 128
            for crp in range(-300, 300):
                if (17207 * cbp + 35707 * crp) % 50000 == 25000:
                    for Y in (0, 1, 2, 100, 101, 200, 255, 300, -3):
                        t.append((Y, cb, crp + 128))
        ycc.append(np.array(t))
        ycc = np.concatenate(ycc).astype(np.int16)
        with contextlib.redirect_stdout(io.StringIO()):
            rgb = jd.YCbCr_to_RGB(ycc.reshape(-1, 1, 3)).reshape(-1, 3)
        np.savez_compressed(GOLD / "ycc_rgb.npz", ycc=ycc, rgb=rgb)

    # --- JPEG file fixtures ------------------------------------------------------------------------
    index = {}
    idx_path = GOLD / "files" / "index.json"
    if args.only and idx_path.exists():
        index = json.loads(idx_path.read_text())
    for name, data in file_fixtures(args.big):
        if args.only and args.only not in name:
            continue
        path = FILES / f"{name}.jpg"
        path.write_bytes(data)
        if name.startswith("prog_"):
            dec, cap = run_reference_progressive(path)
            rgb = np.asarray(dec.image_array)
            coef = pixel_layout_to_blocks(cap["pre_idct"], dec)
            per_scan = np.stack([pixel_layout_to_blocks(a, dec) for a in cap["before_scan"]])
            meta = attrs_of(dec)
            meta["sha256"] = {"coef": sha(coef), "planes": sha(cap["planes"]), "rgb": sha(rgb)}
            np.savez_compressed(FILES / f"{name}.npz", coef=coef, coef_before_scan=per_scan, planes=cap["planes"], rgb=rgb)
            index[name] = meta
            print(f"{name}: {len(data)} B, {coef.shape[0]} blocks, {per_scan.shape[0]} scans, rgb {rgb.shape}")
            continue
        dec, cap = run_reference(path)
        coef, deq, idct_o = np.stack(cap["coef"]), np.stack(cap["deq"]), np.stack(cap["idct"])
        if name.startswith("ni_"):
            # captured scan by scan (all Y blocks, all Cb, all Cr); the seam arrays are kept in the interleaved block
            # order of the coefficient store, like every other fixture: block of MCU m, component c at m * 3 + c
            nb = coef.shape[0] // 3
            order = np.arange(3 * nb).reshape(3, nb).T.reshape(-1)
            coef, deq, idct_o = coef[order], deq[order], idct_o[order]
        rgb = np.asarray(dec.image_array)
        meta = attrs_of(dec)
        meta["sha256"] = {"coef": sha(coef), "deq": sha(deq), "idct": sha(idct_o), "planes": sha(cap["planes"]), "rgb": sha(rgb)}
        if coef.shape[0] > 20000:      # big fixture: hashes + a strided sample only
            sel = np.arange(0, coef.shape[0], 97)
            np.savez_compressed(FILES / f"{name}.npz", block_index=sel, coef=coef[sel], deq=deq[sel], idct=idct_o[sel],
                                planes=cap["planes"][::23, ::19], rgb=rgb[::23, ::19])
            meta["sampled"] = {"block_stride": 97, "x_stride": 23, "y_stride": 19}
        else:
            np.savez_compressed(FILES / f"{name}.npz", coef=coef, deq=deq, idct=idct_o, planes=cap["planes"], rgb=rgb)
        index[name] = meta
        print(f"{name}: {len(data)} B, {coef.shape[0]} blocks, rgb {rgb.shape}")
    idx_path.write_text(json.dumps(index, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
