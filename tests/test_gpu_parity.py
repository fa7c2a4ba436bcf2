"""Parity of the HIP path (through the C ABI) with the reference's goldens and with the CPU oracle.
Needs a real MI355X: run with `-m gpu`."""
import hashlib
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_index, golden_names, load_golden, oracle_rgb_all

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from pyjpegdecoder_amd import BatchDecoder
    d = BatchDecoder(device=0, segment="host")          # (the host's restart segmentation: what most of these tests were written against;
    yield d                                              # `dec_gs` and the default-decoder tests cover the GPU's)
    d.close()


@pytest.fixture(scope="module")
def dec_exact():
    from pyjpegdecoder_amd import BatchDecoder
    d = BatchDecoder(device=0, exact_only=True, segment="host")
    yield d
    d.close()


@pytest.fixture(scope="module")
def dec_rm():
    from pyjpegdecoder_amd import BatchDecoder
    d = BatchDecoder(device=0, layout="rowmajor", segment="host")
    yield d
    d.close()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", golden_names())
def test_fixture_every_seam_bit_exact(dec, name):
    """G1 coefficients, G3 IDCT output, G5 planes, G6 RGB — all bit-exact against the reference."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], vec["coef"]), "G1 coefficients (bit-exact bar)"
    assert np.array_equal(seam["idct"], vec["idct"]), "G3 IDCT output"
    assert np.array_equal(seam["planes"], vec["planes"]), "G5 YCbCr planes"
    assert img.dtype == np.uint8 and img.shape == vec["rgb"].shape
    assert np.array_equal(img, vec["rgb"]), "G6 RGB (bar is +-1; we are exact)"


@pytest.mark.parametrize("name", golden_names())
def test_fixture_exact_order_kernel(dec_exact, name):
    """The exact-order stage-2 kernel (MJ_FLAG_EXACT_ONLY) alone — the fast kernel's referee."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec_exact.decode([raw], return_seams=True)
    assert np.array_equal(seam["idct"], vec["idct"]) and np.array_equal(seam["planes"], vec["planes"])
    assert np.array_equal(img, vec["rgb"])


@pytest.mark.parametrize("name", golden_names())
def test_fixture_production_launch(dec, name):
    """Same files through the launch that produces no seam outputs (fast pixel path)."""
    raw, vec = load_golden(name)
    assert np.array_equal(dec.decode([raw])[0], vec["rgb"])


def test_all_fixtures_in_one_mixed_batch(dec):
    names = golden_names()
    raws = [load_golden(n)[0] for n in names]
    imgs = dec.decode(raws)
    for n, img in zip(names, imgs):
        assert np.array_equal(img, load_golden(n)[1]["rgb"]), n


def test_full_size_1080p_dri_against_reference_hashes(dec):
    """BASELINE config 3 image: SHA-256 of every seam as the reference produced it."""
    name = "c3_1920x1080_420_dri120"
    raw, vec = load_golden(name)
    meta = golden_index()[name]
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert sha(seam["coef"]) == meta["sha256"]["coef"]
    assert sha(seam["idct"]) == meta["sha256"]["idct"]
    assert sha(seam["planes"]) == meta["sha256"]["planes"]
    assert sha(img) == meta["sha256"]["rgb"]
    st = meta["sampled"]
    assert np.array_equal(img[::st["x_stride"], ::st["y_stride"]], vec["rgb"])


@pytest.mark.parametrize("name", golden_names())
def test_rowmajor_fixture_every_seam(dec_rm, name):
    """MJ_LAYOUT_ROWMAJOR runs stage 2 on the transposed problem (blocks stored [u][v], transposed tables): the
    seams still come back in the reference's order and the image is the transpose of the reference's [x, y, c]."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec_rm.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], vec["coef"])
    assert np.array_equal(seam["idct"], vec["idct"])
    assert np.array_equal(seam["planes"], vec["planes"])
    assert np.array_equal(np.swapaxes(img, 0, 1), vec["rgb"])
    assert np.array_equal(np.swapaxes(dec_rm.decode([raw])[0], 0, 1), vec["rgb"]), "production launch"


def test_rowmajor_mixed_batch_and_1080p(dec_rm):
    names = golden_names()
    imgs = dec_rm.decode([load_golden(n)[0] for n in names])
    for n, img in zip(names, imgs):
        assert np.array_equal(np.swapaxes(img, 0, 1), load_golden(n)[1]["rgb"]), n
    name = "c3_1920x1080_420_dri120"
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    (img,), (seam,) = dec_rm.decode([raw], return_seams=True)
    assert img.shape == (1080, 1920, 3)
    assert sha(seam["coef"]) == meta["sha256"]["coef"] and sha(seam["idct"]) == meta["sha256"]["idct"]
    assert sha(seam["planes"]) == meta["sha256"]["planes"]
    assert sha(np.swapaxes(img, 0, 1)) == meta["sha256"]["rgb"]
    assert sha(np.swapaxes(dec_rm.decode([raw])[0], 0, 1)) == meta["sha256"]["rgb"]


RANDOM_CASES = [
    ("420", 333, 211, 5, 85), ("420", 16, 16, 0, 50), ("420", 17, 9, 1, 95), ("422", 250, 130, 9, 90),
    ("440", 130, 250, 4, 75), ("444", 99, 101, 13, 92), ("grey", 123, 77, 6, 88), ("420", 640, 480, 40, 85),
    ("420", 1, 1, 0, 85), ("444", 8, 8, 1, 30), ("420", 1024, 64, 64, 100),
    # 4:1:1 (32x8 MCUs, two-tap upsample over 31): strips of 2 MCUs x-major, of 8 (transposed 8x32 MCUs) row-major
    ("411", 333, 211, 5, 85), ("411", 250, 130, 0, 96), ("411", 31, 7, 1, 90), ("411", 640, 484, 20, 100), ("411", 33, 300, 3, 70),
]


@pytest.mark.parametrize("ss,w,h,ri,q", RANDOM_CASES)
def test_rowmajor_random_sizes_against_oracle(dec_rm, ss, w, h, ri, q):
    from oracle import oracle
    from tools import synth
    sigma = 40.0 if q >= 95 else 12.0
    raw = synth.encode_rgb(synth.synth_rgb(w * 1000 + h + 7, w, h, sigma), q, ss, ri)
    ref = oracle.decode(raw)
    (img,), (seam,) = dec_rm.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"]) and np.array_equal(seam["planes"], ref["planes"])
    assert np.array_equal(np.swapaxes(img, 0, 1), ref["rgb"])
    assert np.array_equal(np.swapaxes(dec_rm.decode([raw])[0], 0, 1), ref["rgb"])


@pytest.mark.parametrize("ss,w,h,ri,q", RANDOM_CASES)
def test_random_sizes_against_oracle(dec, ss, w, h, ri, q):
    """Seeded synthetic files at sizes the oracle finishes in well under a second; bit-exact on G1 and G6."""
    from oracle import oracle
    from tools import synth
    sigma = 40.0 if q >= 95 else 12.0
    raw = synth.encode_rgb(synth.synth_rgb(w * 1000 + h, w, h, sigma), q, ss, ri)
    ref = oracle.decode(raw)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"])
    assert np.array_equal(seam["planes"], ref["planes"])
    assert np.array_equal(img, ref["rgb"])
    # the production launch (no seam outputs) takes the fp32 upsample/colour fast path: same pixels
    assert np.array_equal(dec.decode([raw])[0], ref["rgb"])


def test_config2_idct_only_on_host_decoded_coefficients(dec):
    """BASELINE configs[1]: 512x512 4:2:0 files, entropy decode on the host (here: the oracle's), stage 2
    alone on the GPU through mj_idct_batch's plan path."""
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    raws = [synth.synth_jpeg(100 + i, 512, 512, 85, "420", 0) for i in range(8)]
    raws[0] = load_golden("c2_512x512_420")[0]
    prep = prepare_batch(raws)
    coefs, want = [], []
    for r in raws:
        o = oracle.decode(r)
        coefs.append(o["coef"]); want.append(o["rgb"])
    coef = np.concatenate(coefs)
    bc = prep.to_c()
    rgb = np.empty(sum(x.size for x in want), dtype=np.uint8)
    rc = dec.ctx.lib.mj_idct_batch(dec.ctx.handle, bc, coef.ctypes.data, rgb.ctypes.data)
    dec.ctx.check(rc)
    got = dec.split_outputs(prep, rgb)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    assert np.array_equal(got[0], load_golden("c2_512x512_420")[1]["rgb"])


@pytest.mark.parametrize("layout", ["xmajor", "rowmajor"])
def test_idct_tie_blocks_through_the_kernel(dec, layout):
    """The F6/F7 exact-tie blocks: feed the golden dequantised blocks as coefficients with an all-ones
    quantisation table through stage 2 and compare the IDCT seam."""
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    import ctypes
    g = np.load(GOLDEN / "idct_blocks.npz")
    blocks, want = g["blocks"], g["out"]          # [n,8,8] in [x,y] order
    n = blocks.shape[0]
    # zig-zag them: coef_zz[ZZ_GRID[y][x]] = block[x][y]
    from pyjpegdecoder_amd._parse import ZZ_GRID
    zz = np.zeros((n, 64), dtype=np.int16)
    for x in range(8):
        for y in range(8):
            zz[:, ZZ_GRID[y, x]] = blocks[:, x, y]
    # a greyscale "image" of n blocks: width 8*n, height 8
    d = (B.ImageDescC * 1)()
    d[0].width, d[0].height, d[0].ncomp = 8 * n, 8, 1
    d[0].hs[0] = d[0].vs[0] = 1
    d[0].mcu_count_h, d[0].mcu_count_v = n, 1
    d[0].n_segments = 1
    qt = np.ones((1, 64), dtype=np.uint16)
    bc = B.BatchC()
    bc.n_images = 1; bc.images = ctypes.cast(d, ctypes.POINTER(B.ImageDescC))
    bc.blob = None; bc.blob_mem = B.MJ_MEM_NONE
    bc.n_qt = 1; bc.qt = qt.ctypes.data
    bc.layout = B.MJ_LAYOUT_ROWMAJOR if layout == "rowmajor" else B.MJ_LAYOUT_XMAJOR
    bc.flags = B.MJ_FLAG_KEEP_IDCT | B.MJ_FLAG_KEEP_PLANES
    plan = B.Plan(dec.ctx, bc, {"n_images": 1, "keep": (d, qt)})
    try:
        plan.write_coef(zz)
        plan.execute_stage2()
        plan.sync()
        out = plan.read(rgb=True, idct=True)
    finally:
        plan.close()
    assert np.array_equal(out["idct"].reshape(n, 8, 8), want)


@pytest.mark.parametrize("layout", ["xmajor", "rowmajor"])
def test_colour_conversion_ties_through_the_kernel(dec, layout):
    """YCbCr_to_RGB golden triples (incl. exact .5 ties): build 4:4:4 DC-only blocks whose IDCT gives the
    wanted Y/Cb/Cr (DC*q = 8*(v-128) -> every sample = v), then compare the RGB."""
    from pyjpegdecoder_amd import _binding as B
    import ctypes
    g = np.load(GOLDEN / "ycc_rgb.npz")
    ycc, want = g["ycc"].astype(np.int32), g["rgb"]
    ok = (np.abs(ycc - 128) < 4000).all(axis=1)
    ycc, want = ycc[ok], want[ok]
    n = ycc.shape[0]
    cols = 1024
    rows = -(-n // cols)
    coef = np.zeros((rows * cols, 3, 64), dtype=np.int16)
    coef[:n, :, 0] = (ycc - 128) * 8
    d = (B.ImageDescC * 1)()
    d[0].width, d[0].height, d[0].ncomp = 8 * cols, 8 * rows, 3
    for c in range(3):
        d[0].hs[c] = d[0].vs[c] = 1
    d[0].mcu_count_h, d[0].mcu_count_v = cols, rows
    d[0].n_segments = 1
    qt = np.ones((1, 64), dtype=np.uint16)
    bc = B.BatchC()
    bc.n_images = 1; bc.images = ctypes.cast(d, ctypes.POINTER(B.ImageDescC))
    bc.blob = None; bc.blob_mem = B.MJ_MEM_NONE
    bc.n_qt = 1; bc.qt = qt.ctypes.data
    bc.layout = B.MJ_LAYOUT_ROWMAJOR if layout == "rowmajor" else B.MJ_LAYOUT_XMAJOR
    bc.flags = B.MJ_FLAG_KEEP_PLANES
    plan = B.Plan(dec.ctx, bc, {"n_images": 1, "keep": (d, qt)})
    try:
        plan.write_coef(coef.reshape(-1, 64))
        plan.execute_stage2()
        plan.sync()
        out = plan.read(rgb=True, planes=True)
    finally:
        plan.close()
    planes = out["planes"].reshape(8 * cols, 8 * rows, 3)          # seams keep the reference's [x, y, c] order
    as_xy = (lambda a: a.reshape(8 * rows, 8 * cols, 3).swapaxes(0, 1)) if layout == "rowmajor" else \
            (lambda a: a.reshape(8 * cols, 8 * rows, 3))
    rgb = as_xy(out["rgb"])
    # and once more through the production launch (fp32 fast colour path with its tie detection)
    bc.flags = 0
    plan = B.Plan(dec.ctx, bc, {"n_images": 1, "keep": (d, qt)})
    try:
        plan.write_coef(coef.reshape(-1, 64))
        plan.execute_stage2()
        plan.sync()
        rgb2 = as_xy(plan.read(rgb=True)["rgb"])
    finally:
        plan.close()
    assert np.array_equal(rgb2, rgb)
    # block i sits at MCU (i % cols, i // cols); sample two pixels of each block
    i = np.arange(n)
    bx, by = (i % cols) * 8, (i // cols) * 8
    assert np.array_equal(planes[bx, by, :], ycc.astype(np.int16)), "DC-only construction"
    assert np.array_equal(rgb[bx, by, :], want)
    assert np.array_equal(rgb[bx + 3, by + 5, :], want)


def test_jpegdecoder_class_surface(tmp_path):
    from pyjpegdecoder_amd import JpegDecoder
    for name in ("c1_64x64_444_pil", "70x50_420_pil_opt", "50x70_grey_dri4", "128x64_420_dri3"):
        raw, vec = load_golden(name)
        meta = golden_index()[name]
        f = tmp_path / f"{name}.jpg"
        f.write_bytes(raw)
        d = JpegDecoder(f)
        assert np.array_equal(d.image_array, vec["rgb"]) and d.image_array.dtype == np.uint8
        for k in ("file_size", "file_header", "scan_finished", "scan_mode", "image_width", "image_height",
                  "restart_interval", "scan_count", "scan_amount", "mcu_width", "mcu_height", "mcu_count_h",
                  "mcu_count_v", "mcu_count", "array_width", "array_height", "array_depth"):
            assert getattr(d, k) == meta[k], k
        assert list(d.sample_shape) == meta["sample_shape"] and list(d.mcu_shape) == meta["mcu_shape"]
        assert {str(k): v for k, v in d.huffman_tables.items()} == meta["huffman_tables"]
        assert not hasattr(d, "raw_file") and d.file_path == f
        assert set(d.handlers) == {b"\xFF\xC4", b"\xFF\xDB", b"\xFF\xDD", b"\xFF\xC0", b"\xFF\xC2", b"\xFF\xDA", b"\xFF\xD9"}
    with pytest.raises(AttributeError):
        JpegDecoder(str(f), verbose=True)     # the reference needs a Path too (`file.name`, :41)
    # save (:1490-1532): lossless round trip, never over an existing file, PNG when the suffix has no writer
    from PIL import Image
    first = d.save()
    assert first == f.with_suffix(".png") and first.exists()
    assert np.array_equal(np.swapaxes(np.asarray(Image.open(first)), 0, 1), d.image_array)
    second = d.save(first)
    assert second.name == f"{f.stem} (1).png" and np.array_equal(np.asarray(Image.open(second)), np.asarray(Image.open(first)))
    third = d.save(tmp_path / "picture.nosuchformat")
    assert third == tmp_path / "picture.png" and Image.open(third).format == "PNG"
    bmp = d.save(tmp_path / "picture.bmp")
    assert np.array_equal(np.swapaxes(np.asarray(Image.open(bmp)), 0, 1), d.image_array)


def test_corrupt_streams_raise_like_the_reference(dec):
    from pyjpegdecoder_amd import CorruptedJpeg, parse_jpeg
    raw, _ = load_golden("128x64_420_dri3")
    p = parse_jpeg(raw)
    s = p.scans[0]
    # (a) drop one restart marker -> host sees the wrong number of segments
    off = int(s.segment_offsets[2])
    bad = raw[:off - 2] + raw[off:]
    with pytest.raises(CorruptedJpeg):
        dec.decode([bad])
    # (b) move a restart marker by inserting two data bytes before it -> count-driven restart desyncs
    bad = raw[:off - 2] + b"\x12\x34" + raw[off - 2:]
    with pytest.raises(CorruptedJpeg):
        dec.decode([bad])
    # (c) truncate a segment hard -> it runs out of bits
    off1 = int(s.segment_offsets[1])
    bad = raw[:int(s.entropy_start) + 3] + raw[off1 - 2:]
    with pytest.raises(CorruptedJpeg):
        dec.decode([bad])
    # a good file still decodes afterwards on the same context
    assert np.array_equal(dec.decode([raw])[0], load_golden("128x64_420_dri3")[1]["rgb"])


def test_batch_of_1080p_roundtrip_properties(dec):
    """BASELINE-size batch properties that need no oracle run: determinism across launches and images,
    and equality with the same files decoded one by one."""
    from tools import synth
    blob, offs = synth.synth_batch(6, 7000, 1920, 1080, 85, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(6)]
    a = dec.decode(raws)
    b = dec.decode(raws[::-1])[::-1]
    for x, y in zip(a, b):
        assert x.shape == (1920, 1080, 3) and np.array_equal(x, y)
    single = dec.decode([raws[3]])[0]
    assert np.array_equal(single, a[3])
    from oracle import oracle
    assert np.array_equal(oracle.decode(raws[5])["rgb"], a[5])


@pytest.mark.parametrize("mode", ["wave", "lanes"])
def test_both_stage1_forms_every_fixture(dec, mode, monkeypatch, tune):
    """Stage 1 exists in two forms (one restart segment per wavefront / per lane); force each in turn and
    require the reference's coefficients and pixels from both."""
    tune("MJ_HUFFMAN", mode)
    names = golden_names()
    raws = [load_golden(n)[0] for n in names]
    imgs, seams = dec.decode(raws, return_seams=True)
    for n, img, seam in zip(names, imgs, seams):
        vec = load_golden(n)[1]
        assert np.array_equal(seam["coef"], vec["coef"]), (mode, n)
        assert np.array_equal(img, vec["rgb"]), (mode, n)


@pytest.mark.parametrize("mode", ["wave", "lanes"])
def test_both_stage1_forms_1080p_and_errors(dec, mode, monkeypatch, tune):
    from pyjpegdecoder_amd import CorruptedJpeg, parse_jpeg
    tune("MJ_HUFFMAN", mode)
    name = "c3_1920x1080_420_dri120"
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert sha(seam["coef"]) == meta["sha256"]["coef"] and sha(img) == meta["sha256"]["rgb"]
    # ragged batch: files whose segments have different MCU counts and lengths share wavefronts
    from tools import synth
    from oracle import oracle
    files = [synth.synth_jpeg(50 + i, 64 + 16 * i, 48 + 8 * (i % 5), 80 + i, "420", 1 + (i % 7), 30.0) for i in range(24)]
    for f, img in zip(files, dec.decode(files)):
        assert np.array_equal(img, oracle.decode(f)["rgb"])
    raw, _ = load_golden("128x64_420_dri3")
    s = parse_jpeg(raw).scans[0]
    off = int(s.segment_offsets[2])
    for bad in (raw[:off - 2] + b"\x12\x34" + raw[off - 2:],
                raw[:int(s.entropy_start) + 3] + raw[int(s.segment_offsets[1]) - 2:]):
        with pytest.raises(CorruptedJpeg):
            dec.decode([bad])


def prog_names():
    return sorted(n for n in golden_index() if n.startswith("prog_"))


@pytest.mark.parametrize("name", prog_names())
def test_progressive_fixture_bit_exact(dec, name):
    """BASELINE configs[4] path: progressive scans (DC/AC, first/refining, EOB runs, the reference's `|=`
    refinement semantics F8) on the GPU, then the ordinary stage 2 — coefficients, planes and RGB as the reference."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], vec["coef"]), "coefficient store before the final pass"
    assert np.array_equal(seam["planes"], vec["planes"])
    assert np.array_equal(img, vec["rgb"])
    assert np.array_equal(dec.decode([raw])[0], vec["rgb"])


@pytest.mark.parametrize("name", prog_names())
def test_progressive_rowmajor(dec_rm, name):
    """Progressive scans write the transposed ([u][v]) coefficient store of a row-major plan."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec_rm.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], vec["coef"]) and np.array_equal(seam["planes"], vec["planes"])
    assert np.array_equal(np.swapaxes(img, 0, 1), vec["rgb"])
    assert np.array_equal(np.swapaxes(dec_rm.decode([raw])[0], 0, 1), vec["rgb"])


@pytest.mark.parametrize("mode", ["wave", "lanes"])
def test_rowmajor_both_stage1_forms(dec_rm, mode, monkeypatch, tune):
    tune("MJ_HUFFMAN", mode)
    names = golden_names()
    imgs, seams = dec_rm.decode([load_golden(n)[0] for n in names], return_seams=True)
    for n, img, seam in zip(names, imgs, seams):
        vec = load_golden(n)[1]
        assert np.array_equal(seam["coef"], vec["coef"]), (mode, n)
        assert np.array_equal(np.swapaxes(img, 0, 1), vec["rgb"]), (mode, n)


def test_progressive_batch_and_class_surface(dec, tmp_path):
    from pyjpegdecoder_amd import JpegDecoder
    names = prog_names()
    raws = [load_golden(n)[0] for n in names]
    for n, img in zip(names, dec.decode(raws + [load_golden("64x64_420_pil")[0]])):     # mixed modes: grouped per plan
        assert np.array_equal(img, load_golden(n)[1]["rgb"]), n
    name = "prog_70x50_420_pil"
    f = tmp_path / "p.jpg"
    f.write_bytes(load_golden(name)[0])
    d = JpegDecoder(f)
    meta = golden_index()[name]
    assert np.array_equal(d.image_array, load_golden(name)[1]["rgb"])
    for k in ("scan_mode", "scan_count", "scan_amount", "image_width", "image_height", "file_header", "scan_finished",
              "mcu_count_h", "mcu_count_v", "mcu_count", "mcu_width", "mcu_height", "restart_interval"):
        assert getattr(d, k) == meta[k], k


def test_progressive_larger_random_against_oracle(dec):
    import io
    Image = pytest.importorskip("PIL.Image")      # Pillow writes the progressive test files; skip where it is absent
    from oracle import oracle
    from tools import synth
    for i, (w, h, ss, q) in enumerate([(333, 211, 2, 85), (256, 192, 0, 92), (200, 120, 1, 70)]):
        rgb = synth.synth_rgb(900 + i, w, h, 25.0)
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", quality=q, subsampling=ss, progressive=True)
        raw = b.getvalue()
        ref = oracle.decode(raw)
        (img,), (seam,) = dec.decode([raw], return_seams=True)
        assert np.array_equal(seam["coef"], ref["coef"])
        assert np.array_equal(img, ref["rgb"])


def test_progressive_spec_refinement_switch(dec):
    """the reference refines negative AC coefficients with `|=` on the two's complement, which is not what
    T.81 says.  Default = the reference's behaviour (pinned above); with spec_refine=True the progressive decode of a
    file must give exactly the coefficients of the same pixels coded baseline (libjpeg quantises both identically)."""
    import io
    Image = pytest.importorskip("PIL.Image")      # Pillow writes the progressive test files; skip where it is absent
    from pyjpegdecoder_amd import BatchDecoder
    from tools import synth
    rgb = synth.synth_rgb(77, 96, 80, 30.0)
    files = {}
    for prog in (False, True):
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", quality=88, subsampling=2, progressive=prog)
        files[prog] = b.getvalue()
    (_,), (base,) = dec.decode([files[False]], return_seams=True)
    (_,), (ref_like,) = dec.decode([files[True]], return_seams=True)
    d2 = BatchDecoder(device=0, spec_refine=True)
    try:
        (img_spec,), (spec,) = d2.decode([files[True]], return_seams=True)
        (img_base,) = d2.decode([files[False]])
    finally:
        d2.close()
    assert np.array_equal(spec["coef"], base["coef"]), "spec-correct refinement reproduces the baseline coefficients"
    assert np.array_equal(img_spec, img_base)
    diff = ref_like["coef"] != base["coef"]
    assert diff.any() and (base["coef"][diff] < 0).all(), "the reference's behaviour differs only on negative coefficients"


# ---- restart-marker scan on the GPU -------------------------------------------------------------
@pytest.fixture(scope="module")
def dec_gs():
    from pyjpegdecoder_amd import BatchDecoder
    d = BatchDecoder(device=0, segment="gpu", gpu_segment_min_files=1)     # single files too: these tests are about the GPU scan
    yield d
    d.close()


@pytest.mark.parametrize("name", golden_names())
def test_gpu_segmentation_fixture_every_seam(dec_gs, name):
    """Headers parsed on the host, restart markers and the end of the scan found by k_scan_markers: same seams."""
    raw, vec = load_golden(name)
    (img,), (seam,) = dec_gs.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], vec["coef"]) and np.array_equal(seam["planes"], vec["planes"])
    assert np.array_equal(img, vec["rgb"])


@pytest.mark.parametrize("mode", ["wave", "lanes"])
def test_gpu_segmentation_batches_and_1080p(dec_gs, mode, monkeypatch, tune):
    tune("MJ_HUFFMAN", mode)
    names = golden_names()
    for n, img in zip(names, dec_gs.decode([load_golden(n)[0] for n in names])):
        assert np.array_equal(img, load_golden(n)[1]["rgb"]), n
    name = "c3_1920x1080_420_dri120"
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    (img,), (seam,) = dec_gs.decode([raw], return_seams=True)
    assert sha(seam["coef"]) == meta["sha256"]["coef"] and sha(img) == meta["sha256"]["rgb"]
    from tools import synth
    from oracle import oracle
    files = [synth.synth_jpeg(90 + i, 40 + 24 * i, 200 - 8 * i, 70 + i, ("420", "444", "422", "440", "grey", "411")[i % 6], (i * 3) % 11, 25.0)
             for i in range(20)]
    for f, img in zip(files, dec_gs.decode(files)):
        assert np.array_equal(img, oracle.decode(f)["rgb"])


def test_gpu_segmentation_hands_back_what_it_cannot_segment(dec_gs):
    """A segment between the scan and EOI (here a COM) is MJ_ST_TAIL: the file is re-parsed on the host.  Progressive
    files never take the GPU scan.  A missing restart marker is a corrupt file either way."""
    from pyjpegdecoder_amd import CorruptedJpeg, parse_jpeg
    raw, vec = load_golden("128x64_420_dri3")
    assert raw[-2:] == b"\xff\xd9"
    with_com = raw[:-2] + b"\xff\xfe\x00\x06abcd" + b"\xff\xd9"
    plain = load_golden("64x64_420_pil")
    imgs = dec_gs.decode([with_com, plain[0], raw])
    assert np.array_equal(imgs[0], vec["rgb"]) and np.array_equal(imgs[2], vec["rgb"])
    assert np.array_equal(imgs[1], plain[1]["rgb"])
    name = prog_names()[0]
    praw, pvec = load_golden(name)
    assert np.array_equal(dec_gs.decode([praw])[0], pvec["rgb"])
    s = parse_jpeg(raw).scans[0]
    off = int(s.segment_offsets[2])
    no_marker = raw[:off - 2] + raw[off:]                        # one RSTn removed
    with pytest.raises(CorruptedJpeg):
        dec_gs.decode([no_marker])
    # more restart markers than the scan kernel's list holds (restart interval 1 on 4096 MCUs): handed back as well
    from tools import synth
    from oracle import oracle
    many = synth.synth_jpeg(77, 512, 512, 85, "444", 1, 10.0)
    assert np.array_equal(dec_gs.decode([many])[0], oracle.decode(many)["rgb"])
    mid = synth.synth_jpeg(78, 256, 128, 85, "420", 1, 10.0)     # 128 segments of one MCU: fits, scanned on the GPU
    assert np.array_equal(dec_gs.decode([mid])[0], oracle.decode(mid)["rgb"])
    trailing = raw + b"\x00" * 37                                # bytes after EOI are nobody's business
    assert np.array_equal(dec_gs.decode([trailing])[0], vec["rgb"])


def test_gpu_segmentation_device_blob_with_only_the_documented_slack(dec):
    """MJ_FLAG_GPU_SEGMENT asks a caller for 16 readable bytes behind blob_len, no more.  A plan of such a blob that takes the
    wave form of stage 1 (a generic sampling layout here; the kernel reads up to 508 bytes behind a segment's start) must
    work on its own padded copy: the caller's allocation ends 16 bytes behind a tiny last file, what follows it in memory
    is poison, and the decode has to come out right — twice, with the caller's bytes changed in between (the copy is made
    at every execute, on the execute's stream)."""
    import torch
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B, parse_jpeg
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import craft_jpeg
    files = [craft_jpeg.craft_baseline(40, 24, [(1, 1), (2, 1), (1, 2)], seed=5, restart_interval=2),
             craft_jpeg.craft_baseline(8, 8, [(1, 1), (2, 1), (1, 2)], seed=6, restart_interval=0)]        # one MCU: a few bytes of scan
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0, [parse_jpeg(f, headers_only=True) for f in files])
    assert prep.flags & B.MJ_FLAG_GPU_SEGMENT
    used = int(prep.file_offsets[-1])                       # the files, back to back (4-byte aligned each)
    prep.blob = prep.blob[:used]                            # blob_len = what the files take: the caller pads 16 bytes, not 1024
    dev = torch.device("cuda", 0)
    arena = torch.full((used + 16 + 4096,), 0xFF, dtype=torch.uint8, device=dev)      # poison behind the 16 bytes
    arena[:used] = torch.from_numpy(prep.blob).to(dev)
    arena[used:used + 16] = 0
    assert arena.data_ptr() % 16 == 0
    plan = B.Plan(dec.ctx, prep.to_c(arena.data_ptr()), {"prep": prep, "n_images": 2})
    try:
        assert plan.stage1_form() & 15 == B.MJ_FORM_WAVE
        for rep in range(2):
            plan.execute()
            plan.sync()
            out = plan.read(rgb=True)
            assert not out["status"].any()
            off = 0
            for f, (w, h, nc) in zip(files, prep.shapes):
                ref = oracle.decode(f)["rgb"]
                assert np.array_equal(out["rgb"][off:off + w * h * nc].reshape(ref.shape), ref), rep
                off += w * h * nc
            arena[:used] = torch.from_numpy(prep.blob).to(dev)          # (same bytes again: what matters is that execute re-reads them)
    finally:
        plan.close()


def test_two_contexts_on_one_device_from_two_threads():
    """mijpeg.h: one context per GPU per thread.  Two threads, a context each on device 0, decode different batches at the same
    time through every stage-1 form's launcher (whose launch-geometry caches are per device, not per process)."""
    import threading
    from oracle import oracle
    from pyjpegdecoder_amd import BatchDecoder
    from tools import synth
    jobs = [[synth.synth_jpeg(300 + 10 * t + i, 160 + 16 * i, 120 + 8 * t, 80, ("420", "444")[t], (0, 4)[i % 2], 14.0) for i in range(6)] for t in range(2)]
    outs, errs = [None, None], []

    def work(t):
        try:
            d = BatchDecoder(device=0)
            try:
                for _ in range(3):
                    outs[t] = d.decode(jobs[t])
            finally:
                d.close()
        except Exception as exc:      # noqa: BLE001 — reported below, in the main thread
            errs.append(exc)
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errs, errs
    for t in range(2):
        for f, img in zip(jobs[t], outs[t]):
            assert np.array_equal(img, oracle.decode(f)["rgb"])


def test_first_launches_of_every_form_from_two_threads_in_a_fresh_process(tmp_path):
    """The launchers raise a kernel's dynamic-LDS limit and size their persistent grids at a kernel's FIRST launch on a device —
    once per (kernel, device), under std::call_once / atomics since round 6 (round 5: plain static flags).  By the time a test of
    this suite runs, every kernel has been launched; so this one starts a FRESH process whose first act is two threads, a context
    each, creating and executing plans of different forms at the same moment (a barrier in front of every execute): the fused
    launch (x-major and row-major), the lane kernels of the two launches, the synchronisation form, the wave form — crossed over,
    so that both threads make the same kernel's first launch in the same round.  Every image against the oracle."""
    import subprocess
    import sys
    script = tmp_path / "two_threads.py"
    script.write_text(
        "import sys, threading\n"
        "import numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch\n"
        "from oracle import oracle\n"
        "from tools import synth\n"
        "from pyjpegdecoder_amd import _binding as B\n"
        "from pyjpegdecoder_amd.batch import prepare_batch\n"
        "def batch(kind, seed):\n"
        "    if kind == 'fused' or kind == 'fused_rm' or kind == 'two':\n"
        "        raws = [synth.synth_jpeg(seed + i, 160, 128, 85, '420', 10) for i in range(4)]; n = 400\n"
        "    elif kind == 'sync':\n"
        "        raws = [synth.synth_jpeg(seed + i, 640, 480, 85, '420', 0) for i in range(4)]; n = 24\n"
        "    else:\n"
        "        raws = [synth.synth_jpeg(seed + i, 96, 80, 85, '444', 3) for i in range(4)]; n = 6\n"
        "    files = [raws[i % 4] for i in range(n)]\n"
        "    lay = B.MJ_LAYOUT_ROWMAJOR if kind == 'fused_rm' else B.MJ_LAYOUT_XMAJOR\n"
        "    return raws, files, prepare_batch(files, lay, 0), lay\n"
        "orders = [['fused', 'sync', 'two', 'wave', 'fused_rm'], ['sync', 'fused', 'fused_rm', 'two', 'wave']]\n"
        "work = [[batch(k, 1000 * t + 100 * j) + (k,) for j, k in enumerate(orders[t])] for t in range(2)]\n"
        "gate = threading.Barrier(2)\n"
        "errs = []\n"
        "def run(t):\n"
        "    try:\n"
        "        ctx = B.Context(0)\n"
        "        for raws, files, prep, lay, kind in work[t]:\n"
        "            gate.wait(timeout=120)\n"
        "            plan = B.Plan(ctx, prep.to_c(), {'prep': prep, 'n_images': len(files)})\n"
        "            try:\n"
        "                if kind == 'two': plan.execute_stage1(); plan.execute_stage2()      # (the stages' own kernels, no process-wide switch)\n"
        "                else: plan.execute()\n"
        "                plan.sync()\n"
        "                out = plan.read(rgb=True)\n"
        "                form = plan.stage1_form()\n"
        "            finally:\n"
        "                plan.close()\n"
        "            assert not out['status'].any(), (t, kind)\n"
        "            if kind.startswith('fused'): assert form & B.MJ_FORM_FUSED, (kind, form)\n"
        "            if kind == 'sync': assert (form & 15) == B.MJ_FORM_SYNC, form\n"
        "            off = 0\n"
        "            for i, (f, (w, h, nc)) in enumerate(zip(files, prep.shapes)):\n"
        "                if i < 8:\n"
        "                    ref = oracle.decode(f)['rgb']\n"
        "                    got = out['rgb'][off:off + w * h * nc]\n"
        "                    got = got.reshape(ref.shape) if lay == B.MJ_LAYOUT_XMAJOR else np.swapaxes(got.reshape(h, w, nc), 0, 1)\n"
        "                    assert np.array_equal(got, ref), (t, kind, i)\n"
        "                off += w * h * nc\n"
        "        ctx.close()\n"
        "    except BaseException as exc:\n"
        "        errs.append(repr(exc)); gate.abort()\n"
        "ths = [threading.Thread(target=run, args=(t,)) for t in range(2)]\n"
        "[th.start() for th in ths]; [th.join(timeout=600) for th in ths]\n"
        "print('ERRORS', errs) if errs else print('OK two threads')\n"
        "sys.exit(1 if errs else 0)\n")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK two threads" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_the_environment_does_not_choose_the_stage1_form(dec, monkeypatch):
    """The library's switches are set through mj_set_option only: a stray MJ_HUFFMAN in the environment of a production
    process changes nothing (round 3's library read it with getenv at every plan creation)."""
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    raw, _ = load_golden("c3_1920x1080_420_dri120")
    prep = prepare_batch([raw] * 16, B.MJ_LAYOUT_XMAJOR, 0)
    forms = []
    for env in (None, "wave"):
        if env:
            monkeypatch.setenv("MJ_HUFFMAN", env)
        plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": 16})
        forms.append(plan.stage1_form() & 15)
        plan.close()
    assert forms[0] == forms[1] != B.MJ_FORM_WAVE
    B.set_option("MJ_HUFFMAN", "wave")
    try:
        plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": 16})
        assert plan.stage1_form() & 15 == B.MJ_FORM_WAVE
        plan.close()
    finally:
        B.set_option("MJ_HUFFMAN", None)
    with pytest.raises(ValueError):
        B.set_option("MJ_NO_SUCH_SWITCH", "1")


def test_device_resident_output_and_dlpack(dec, dec_rm):
    """§8 f-4: pixels stay in HBM as torch tensors (views of one packed buffer); any DLPack consumer can take them."""
    import torch
    names = ["70x50_420_pil_opt", "64x48_422_pil", "50x70_grey_dri4", "40x40_444_dri5", "128x64_420_dri3"]
    raws = [load_golden(n)[0] for n in names]
    outs = dec.decode_device(raws)
    for n, t in zip(names, outs):
        assert t.is_cuda and t.dtype == torch.uint8
        assert np.array_equal(t.cpu().numpy(), load_golden(n)[1]["rgb"]), n
        again = torch.from_dlpack(t)                       # zero-copy round trip
        assert again.data_ptr() == t.data_ptr()
    for n, t in zip(names, dec_rm.decode_device(raws)):
        want = load_golden(n)[1]["rgb"]
        assert np.array_equal(t.cpu().numpy(), np.swapaxes(want, 0, 1)), n


def test_native_host_front_end_through_decode_device(dec_gs, monkeypatch):
    """segment="gpu" + decode_device: a one-layout batch of baseline files is parsed and assembled by mj_host_assemble
    (never by _parse.py); mixed layouts, progressive files and files with a tail behind the scan fall back to the Python
    path.  Same pixels every way."""
    import torch
    from tools import synth
    from oracle import oracle
    import pyjpegdecoder_amd.batch as batch_mod
    files = [synth.synth_jpeg(300 + i, 96 + 16 * (i % 4), 80 + 8 * i, 60 + 3 * i, "420", (0, 6, 0, 3)[i % 4], 20.0) for i in range(12)]
    want = [oracle.decode(f)["rgb"] for f in files]
    calls = {"native": 0, "python": 0}
    real_native, real_parse = batch_mod.prepare_batch_native, batch_mod.parse_jpeg

    def spy_native(*a, **k):
        r = real_native(*a, **k)
        calls["native"] += isinstance(r, batch_mod.PreparedBatch)      # (a list = fine files, several plans: not an assembly yet)
        return r

    def spy_parse(*a, **k):
        calls["python"] += 1
        return real_parse(*a, **k)
    monkeypatch.setattr(batch_mod, "prepare_batch_native", spy_native)
    monkeypatch.setattr(batch_mod, "parse_jpeg", spy_parse)
    outs = dec_gs.decode_device(files)
    assert calls == {"native": 2, "python": 0}                 # files with and without restart markers: two plans
    for t, w in zip(outs, want):
        assert t.is_cuda and np.array_equal(t.cpu().numpy(), w)
    outs = dec_gs.decode_device(files)                         # the staging buffer is reused
    assert all(np.array_equal(t.cpu().numpy(), w) for t, w in zip(outs, want))
    # a tail behind one file's scan: that file alone is redone through the Python path
    with_com = files[3][:-2] + b"\xff\xfe\x00\x06abcd" + b"\xff\xd9"
    calls.update(native=0, python=0)
    outs = dec_gs.decode_device(files[:3] + [with_com] + files[4:])
    assert calls == {"native": 2, "python": 1}
    assert all(np.array_equal(t.cpu().numpy(), w) for t, w in zip(outs, want))
    # fine files of several kinds (another sampling layout; with and without restart markers): one native assembly per kind
    other = synth.synth_jpeg(400, 64, 64, 85, "444", 0, 20.0)
    calls.update(native=0, python=0)
    outs = dec_gs.decode_device(files + [other, other])
    assert calls["python"] == 0 and calls["native"] == 3          # 4:2:0 without markers, 4:2:0 with markers, 4:4:4
    assert all(np.array_equal(t.cpu().numpy(), w) for t, w in zip(outs, want))
    assert np.array_equal(outs[-1].cpu().numpy(), oracle.decode(other)["rgb"])
    # a progressive file among them: it alone takes the Python path, the rest stays with the front end
    praw, pvec = load_golden(prog_names()[0])
    calls.update(native=0, python=0)
    outs = dec_gs.decode_device(files[:2] + [other, praw])
    assert calls["native"] == 3 and calls["python"] == 1
    assert np.array_equal(outs[2].cpu().numpy(), oracle.decode(other)["rgb"]) and np.array_equal(outs[3].cpu().numpy(), pvec["rgb"])
    assert np.array_equal(outs[0].cpu().numpy(), want[0])
    # native_host=False: the Python path alone, same pixels
    from pyjpegdecoder_amd import BatchDecoder, NotJpeg
    d2 = BatchDecoder(0, segment="gpu", native_host=False)
    try:
        calls.update(native=0, python=0)
        outs = d2.decode_device(files)
        assert calls["native"] == 0 and calls["python"] == len(files)
        assert all(np.array_equal(t.cpu().numpy(), w) for t, w in zip(outs, want))
    finally:
        d2.close()
    with pytest.raises(NotJpeg):                               # declined natively, diagnosed by the Python loop
        dec_gs.decode_device(files[:2] + [b"not a jpeg at all"])


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipelined_stream_of_batches(dec_gs, depth):
    """decode_device_iter: the next batches are assembled and uploaded while the ones before are on the GPU (`depth` of them in
    flight — round 6: two by default, so that host threads, copy engine and GPU run side by side); results come in order and are
    the same pixels, also when a batch is declined by the front end, has a tail behind a scan, or is empty-handed."""
    from tools import synth
    from oracle import oracle
    from pyjpegdecoder_amd import CorruptedJpeg
    pool = [synth.synth_jpeg(500 + i, 120 + 8 * (i % 5), 64 + 4 * i, 70 + i, "420", (0, 4, 9)[i % 3], 18.0) for i in range(10)]
    want = [oracle.decode(f)["rgb"] for f in pool]
    praw, pvec = load_golden(prog_names()[0])
    with_com = pool[1][:-2] + b"\xff\xfe\x00\x06abcd" + b"\xff\xd9"
    batches = [[0, 1, 2, 3], [4, 5], [6, 7, 8, 9, 0, 1, 2], "prog", [3, 4], "tail", [9], [5, 6, 7, 8, 9, 0, 1, 2, 3, 4]]

    def gen():
        for b in batches:
            if b == "prog":
                yield [pool[0], praw]
            elif b == "tail":
                yield [pool[0], with_com, pool[2]]
            else:
                yield [pool[i] for i in b]
    outs = list(dec_gs.decode_device_iter(gen(), depth=depth))
    assert len(outs) == len(batches)
    for b, o in zip(batches, outs):
        if b == "prog":
            assert np.array_equal(o[0].cpu().numpy(), want[0]) and np.array_equal(o[1].cpu().numpy(), pvec["rgb"])
        elif b == "tail":
            assert [np.array_equal(t.cpu().numpy(), want[i]) for t, i in zip(o, (0, 1, 2))] == [True] * 3
        else:
            assert len(o) == len(b)
            for t, i in zip(o, b):
                assert t.is_cuda and np.array_equal(t.cpu().numpy(), want[i])
    # results of earlier batches stay valid while later ones are decoded (their buffers are theirs)
    assert np.array_equal(outs[0][0].cpu().numpy(), want[0])
    # a corrupt file surfaces as the reference's exception when its batch is handed out
    s0 = pool[1].find(b"\xff\xda")
    broken = pool[1][:s0 + 40] + bytes(200) + pool[1][s0 + 240:]
    with pytest.raises(CorruptedJpeg):
        list(dec_gs.decode_device_iter([[pool[0]], [broken, pool[2]], [pool[3]], [pool[5]], [pool[6]]], depth=depth))
    assert np.array_equal(dec_gs.decode_device([pool[4]])[0].cpu().numpy(), want[4])      # the decoder is still usable
    # a consumer that stops early leaves nothing open
    it = dec_gs.decode_device_iter(([pool[i]] for i in range(8)), depth=depth)
    first = next(it)
    assert np.array_equal(first[0].cpu().numpy(), want[0])
    it.close()
    assert np.array_equal(dec_gs.decode_device([pool[7]])[0].cpu().numpy(), want[7])


def test_one_call_in_overlapped_parts(dec_gs):
    """decode_device(files, parts=k): a large batch goes as k plans through the pipelined route (one part's upload under the next
    part's assembly and the kernels of the one before; by default from 512 files on).  Same tensors in the same order, also with
    a progressive file and a file with a tail behind its scan among them (those parts take the long way), and the reference's
    exception for a corrupt file."""
    from tools import synth
    from oracle import oracle
    from pyjpegdecoder_amd import CorruptedJpeg
    pool = [synth.synth_jpeg(700 + i, 96 + 8 * (i % 4), 64 + 8 * (i % 3), 75, "420", (0, 3, 6)[i % 3], 12.0) for i in range(12)]
    want = [oracle.decode(f)["rgb"] for f in pool]
    praw, pvec = load_golden(prog_names()[0])
    files = [pool[i % 12] for i in range(40)]
    for parts in (1, 2, 3, 4):
        got = dec_gs.decode_device(files, parts=parts)
        assert len(got) == 40
        for i, g in enumerate(got):
            assert g.is_cuda and np.array_equal(g.cpu().numpy(), want[i % 12]), (parts, i)
    mixed = list(files)
    mixed[7] = praw
    mixed[29] = pool[5][:-2] + b"\xff\xfe\x00\x06abcd" + b"\xff\xd9"
    got = dec_gs.decode_device(mixed, parts=4)
    for i, g in enumerate(got):
        assert np.array_equal(g.cpu().numpy(), pvec["rgb"] if i == 7 else want[i % 12]), i
    s0 = pool[1].find(b"\xff\xda")
    broken = list(files)
    broken[33] = pool[1][:s0 + 40] + bytes(200) + pool[1][s0 + 240:]
    with pytest.raises(CorruptedJpeg):
        dec_gs.decode_device(broken, parts=4)
    # the default: parts of 256 files or more
    many = [pool[i % 12] for i in range(520)]
    got = dec_gs.decode_device(many)
    assert len(got) == 520 and got[0].data_ptr() != got[519].data_ptr()
    for i in (0, 259, 260, 519):
        assert np.array_equal(got[i].cpu().numpy(), want[i % 12])
    assert sum(g.numel() for g in got[:260]) <= (got[259].data_ptr() - got[0].data_ptr()) + got[259].numel()      # one buffer per part


@pytest.mark.parametrize("cache_mb", ["0", "1", "64"])
def test_device_buffer_cache_limits(cache_mb, monkeypatch):
    """A context keeps the device buffers of destroyed plans for the next plan (bounded, MJ_CACHE_MB).  Whatever the bound
    — nothing kept, less than one plan's worth, plenty — plans of changing sizes decode the same pixels, and recycled
    buffers (which hold another plan's bytes, not zeros) change nothing."""
    from pyjpegdecoder_amd import BatchDecoder
    monkeypatch.setenv("MJ_CACHE_MB", cache_mb)
    d = BatchDecoder(0)
    try:
        names = golden_names()
        files = {n: load_golden(n) for n in names}
        rng = np.random.default_rng(int(cache_mb))
        for rep in range(12):
            pick = [names[i] for i in rng.integers(0, len(names), int(rng.integers(1, 9)))]
            for n, img in zip(pick, d.decode([files[n][0] for n in pick])):
                assert np.array_equal(img, files[n][1]["rgb"]), (rep, n)
        pn = prog_names()[0]
        assert np.array_equal(d.decode([load_golden(pn)[0]])[0], load_golden(pn)[1]["rgb"])
    finally:
        d.close()


def _pil_files(n, size, noise=6.0, **kw):
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    from tools import synth
    out = []
    for i in range(n):
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(700 + i, size[0], size[1], noise + 3 * (i % 5))).save(b, "JPEG", quality=70 + i % 20, **kw)
        out.append(b.getvalue())
    return out


@pytest.mark.parametrize("seg", ["host", "gpu"])
def test_files_with_their_own_huffman_tables_keep_the_fast_forms(seg):
    """Every file with optimised tables of its own: far more tables in the batch than LDS holds.  Each workgroup then loads
    the tables of the images its segments belong to (MJ_FORM_WG_TABLES) instead of the batch falling back to one
    wavefront per segment.  With restart markers (lane form) and without (synchronisation form)."""
    from oracle import oracle
    from pyjpegdecoder_amd import BatchDecoder, _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from pyjpegdecoder_amd._parse import parse_jpeg
    d = BatchDecoder(0, segment=seg)
    try:
        # (the synchronisation form groups 256 chunks of 2 KiB per workgroup: files of >= 512 KiB keep that to two images)
        for kw, base, n, size, noise in ((dict(optimize=True, subsampling=2, restart_marker_rows=1), B.MJ_FORM_LANES, 18, (1024, 1040), 6.0),
                                         (dict(optimize=True, subsampling=2), B.MJ_FORM_SYNC, 10, (1920, 1088), 30.0)):
            files = _pil_files(n, size, noise, **kw)
            assert base != B.MJ_FORM_SYNC or min(len(f) for f in files) > 600 * 1024
            assert len({bytes(parse_jpeg(f).scans[0].huffman[0x10].vals.tobytes()) + bytes(parse_jpeg(f).scans[0].huffman[0x10].bits.tobytes()) for f in files}) > 8
            prep, plan = d.plan(files)
            try:
                assert prep.n_huff > 16
                # (small batches of restart segments are cut into chunks as well: either fast form will do for those)
                want = {base | B.MJ_FORM_WG_TABLES, B.MJ_FORM_SYNC | B.MJ_FORM_WG_TABLES}
                assert plan.stage1_form() in want, plan.stage1_form()
            finally:
                plan.close()
            for f, img in zip(files, d.decode(files)):
                assert np.array_equal(img, oracle.decode(f)["rgb"])
        # medium files: three or four images per workgroup -> sixteen LUTs per workgroup instead of eight, still not the wave form
        for kw, base in ((dict(optimize=True, subsampling=2, restart_marker_rows=1), B.MJ_FORM_LANES), (dict(optimize=True, subsampling=2), B.MJ_FORM_SYNC)):
            files = _pil_files(24, (640, 480), 25.0, **kw)
            prep, plan = d.plan(files)
            try:
                form_medium = plan.stage1_form()
                # (whether a given size still fits is the planner's business; what is pinned is: never a wrong picture)
                assert form_medium in (base | B.MJ_FORM_WG_TABLES, B.MJ_FORM_SYNC | B.MJ_FORM_WG_TABLES, B.MJ_FORM_WAVE)
            finally:
                plan.close()
            for f, img in zip(files, d.decode(files)):
                assert np.array_equal(img, oracle.decode(f)["rgb"]), (base, form_medium)
        # small images: a workgroup's segments span more images than its table list holds -> the wave form, same pixels
        small = _pil_files(40, (96, 80), optimize=True, subsampling=2, restart_marker_rows=1)
        prep, plan = d.plan(small)
        try:
            assert plan.stage1_form() == B.MJ_FORM_WAVE
        finally:
            plan.close()
        for f, img in zip(small, d.decode(small)):
            assert np.array_equal(img, oracle.decode(f)["rgb"])
    finally:
        d.close()


def test_damaged_streams_never_hang_or_crash(dec, dec_gs, tune):
    """Robustness: random byte damage inside the entropy-coded data either decodes to some image or raises the
    reference's CorruptedJpeg — in both stage-1 forms and with either segmentation — and never takes the GPU down."""
    import os
    from pyjpegdecoder_amd import CorruptedJpeg, JpegError, parse_jpeg
    rng = np.random.default_rng(1234)
    for name in ("128x64_420_dri3", "64x64_420_pil", "40x40_444_dri5", "50x70_grey_dri4"):
        raw, vec = load_golden(name)
        sc = parse_jpeg(raw).scans[0]
        lo, hi = int(sc.entropy_start), int(sc.entropy_end)
        damaged = []
        for _ in range(24):
            b = bytearray(raw)
            for _ in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(lo, hi))
                b[pos] = int(rng.integers(0, 256))
            damaged.append(bytes(b))
        damaged.append(raw[:lo + (hi - lo) // 2] + raw[hi:])          # half the entropy data missing
        damaged.append(raw[:hi] + b"\xff\xd9")
        for mode in ("wave", "lanes"):
            tune("MJ_HUFFMAN", mode)
            try:
                for d in (dec, dec_gs):
                    for f in damaged:
                        try:
                            img = d.decode([f])[0]
                            assert img.shape == vec["rgb"].shape
                        except JpegError:
                            pass
                    assert np.array_equal(d.decode([raw])[0], vec["rgb"])       # the decoder is still healthy
            finally:
                tune("MJ_HUFFMAN", None)


def test_damaged_progressive_streams_both_walks_agree(dec, monkeypatch, tune):
    """Random byte damage inside the scans of progressive files: the stream walks (progressive_fast.hip) and the general
    walk (progressive.hip) implement the same reader — zeros behind a segment's end, the same overrun / desync / bad-code
    rules — so they must fail on the same files and, where the damage still decodes, leave the same pixels."""
    from pyjpegdecoder_amd import JpegError, parse_jpeg
    rng = np.random.default_rng(4321)
    names = prog_names()
    for name in names[:6]:
        raw, vec = load_golden(name)
        scans = parse_jpeg(raw).scans
        damaged = []
        for _ in range(20):
            b = bytearray(raw)
            for _ in range(int(rng.integers(1, 4))):
                sc = scans[int(rng.integers(0, len(scans)))]
                lo, hi = int(sc.entropy_start), int(sc.entropy_end)
                if hi > lo:
                    b[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
            damaged.append(bytes(b))
        outcomes = {}
        for walk in ("stream", "general"):
            if walk == "general":
                tune("MJ_PROG_FAST", "0")
            else:
                tune("MJ_PROG_FAST", None)
            res = []
            for f in damaged:
                try:
                    res.append(dec.decode([f])[0])
                except JpegError:
                    res.append(None)
            outcomes[walk] = res
            assert np.array_equal(dec.decode([raw])[0], vec["rgb"]), (name, walk)       # the decoder is still healthy
        for i, (a, b) in enumerate(zip(outcomes["stream"], outcomes["general"])):
            assert (a is None) == (b is None), (name, i)
            if a is not None:
                assert np.array_equal(a, b), (name, i)


def test_reference_repository_example_progressive_file(dec):
    """4160x2340 4:2:0 progressive with DRI redefined between scans: the two truncations whose
    decoded images the reference repository ships as PNGs, and the whole file against the oracle."""
    from test_oracle_golden import _example, example_cut
    from oracle import oracle
    raw, meta, samples = _example()
    cuts = [example_cut(raw, meta, 1), example_cut(raw, meta, 2)]
    imgs = dec.decode(cuts + [raw])
    for k in (1, 2):
        assert sha(imgs[k - 1]) == meta["after_scan"][str(k)]["sha256_rgb_xmajor"], f"after scan {k}"
    assert np.array_equal(imgs[2], oracle.decode(raw)["rgb"])


def test_caller_stream_and_device_blob(dec):
    """The C ABI takes a caller's HIP stream and device pointers (blob in, pixels out): work queued on a non-default
    torch stream is ordered with that stream's other work and nothing else is synchronised."""
    import torch
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    names = ["128x64_420_dri3", "100x36_420_dri7", "96x64_420_q100_noise"]
    raws = [load_golden(n)[0] for n in names]
    prep = prepare_batch(raws)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        d_blob = torch.from_numpy(prep.blob).to(dev, non_blocking=True)
        plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(raws)})
        try:
            d_rgb = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
            for _ in range(3):                                   # re-execution of one plan is idempotent
                plan.execute(st.cuda_stream, d_rgb.data_ptr())
            flipped = 255 - d_rgb                                # consumer on the same stream, no explicit sync in between
            st.synchronize()
            out = dec.split_outputs(prep, d_rgb.cpu().numpy())
            for n, img in zip(names, out):
                assert np.array_equal(img, load_golden(n)[1]["rgb"]), n
            assert torch.equal(flipped, 255 - d_rgb)
            assert not plan.read(rgb=False)["status"].any()
            # mj_plan_sync waits for the plan's own latest execute on whichever stream it went: no stream synchronisation by
            # the caller, and the pixels are complete
            d_rgb2 = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
            plan.execute(st.cuda_stream, d_rgb2.data_ptr())
            plan.sync()
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                assert torch.equal(d_rgb2.to("cpu", non_blocking=False), d_rgb.cpu())
        finally:
            plan.close()


def test_one_large_image_and_ragged_batch(dec, dec_rm, tune):
    """A 24-megapixel 4:2:0 image (6016x4000, no DRI: one 4 MB segment through the wave form; with DRI through the
    lane form) and a batch whose images differ in size (per-image tile prefix, binary search in stage 2)."""
    from tools import synth
    from oracle import oracle
    big = synth.synth_jpeg(5, 6016, 4000, 85, "420", 0, 12.0)
    ref = oracle.decode(big)["rgb"]
    assert np.array_equal(dec.decode([big])[0], ref)
    assert np.array_equal(np.swapaxes(dec_rm.decode([big])[0], 0, 1), ref)
    big_dri = synth.synth_jpeg(5, 6016, 4000, 85, "420", 47, 12.0)          # 47 MCUs per segment: 2000 segments
    ragged = [big_dri] + [synth.synth_jpeg(100 + i, 333 + 160 * i, 1200 - 97 * i, 85, "420", 11, 12.0) for i in range(9)]
    import os
    tune("MJ_HUFFMAN", "lanes")
    try:
        outs = dec.decode(ragged)
    finally:
        tune("MJ_HUFFMAN", None)
    assert np.array_equal(outs[0], ref)
    for f, img in zip(ragged[1:], outs[1:]):
        assert np.array_equal(img, oracle.decode(f)["rgb"])


def test_full_config3_batch_properties(dec):
    """BASELINE configs[2] at full size: 1024 x 1920x1080 4:2:0 with DRI = one MCU row, 64 distinct images tiled.
    Size-independent properties: every copy of an image decodes to the same bytes wherever it sits in the batch
    (checksums of all 1024 outputs), no image reports a status, and every one of the distinct images equals the oracle."""
    from tools import synth
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, n, distinct = 1920, 1080, 1024, 64
    blob, offs = synth.synth_batch(distinct, 4242, W, H, 85, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    files = [raws[(7 * i + i // distinct) % distinct] for i in range(n)]          # copies land in different waves / lanes
    prep = prepare_batch(files)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": n})
    try:
        plan.execute()
        plan.sync()
        out = plan.read(rgb=True)
    finally:
        plan.close()
    assert not out["status"].any()
    per = W * H * 3
    rgb = out["rgb"].reshape(n, per)
    sums = rgb.view(np.uint64).sum(axis=1, dtype=np.uint64) ^ rgb.view(np.uint64)[:, ::977].sum(axis=1, dtype=np.uint64) * np.uint64(2654435761)
    by_image = {}
    for i in range(n):
        by_image.setdefault((7 * i + i // distinct) % distinct, set()).add(int(sums[i]))
    assert all(len(v) == 1 for v in by_image.values()), "copies of one image differ"
    assert len({next(iter(v)) for v in by_image.values()}) == distinct
    for d, want in enumerate(oracle_rgb_all(raws)):           # every distinct image against the oracle, its copies against each other (above)
        i = next(k for k in range(n) if (7 * k + k // distinct) % distinct == d)
        assert np.array_equal(rgb[i].reshape(W, H, 3), want), d


def test_randomised_sweep_against_oracle(dec, dec_rm, dec_gs, tune):
    """80 seeded files over size x sampling x quality x restart interval x noise, decoded in mixed batches by the
    x-major, the row-major and the GPU-segmenting decoder; every pixel against the oracle."""
    from tools import synth
    from oracle import oracle
    rng = np.random.default_rng(20261003)
    files = []
    for i in range(80):
        ss = ("420", "444", "422", "440", "grey", "444ni", "411")[int(rng.integers(0, 7))]
        w, h = int(rng.integers(1, 260)), int(rng.integers(1, 200))
        q = int(rng.choice([20, 50, 75, 85, 92, 98, 100]))
        ri = int(rng.choice([0, 0, 1, 2, 3, 7, 16, 50]))
        sigma = float(rng.choice([0.0, 5.0, 12.0, 40.0]))
        files.append(synth.synth_jpeg(1000 + i, w, h, q, ss, ri, sigma))
    refs = [oracle.decode(f)["rgb"] for f in files]
    for d, fix in ((dec, lambda a: a), (dec_rm, lambda a: np.swapaxes(a, 0, 1)), (dec_gs, lambda a: a)):
        for i, (img, ref) in enumerate(zip(d.decode(files), refs)):
            assert np.array_equal(fix(img), ref), i
    # and once more with every stage-1 form forced (the sync form then cuts even these small segments where it can)
    import os
    for mode in ("wave", "lanes", "sync"):
        tune("MJ_HUFFMAN", mode)
        try:
            for i, (img, ref) in enumerate(zip(dec.decode(files), refs)):
                assert np.array_equal(img, ref), (mode, i)
        finally:
            tune("MJ_HUFFMAN", None)


# ---- long segments: synchronisation passes + virtual segments (huffman_sync.hip) ---------------------------------------
def test_sync_form_every_fixture(dec, monkeypatch, tune):
    """MJ_HUFFMAN=sync forces the third stage-1 form on everything: segments shorter than a chunk are one virtual
    segment each, longer ones are cut up; coefficients and pixels as the reference's."""
    tune("MJ_HUFFMAN", "sync")
    names = golden_names()
    imgs, seams = dec.decode([load_golden(n)[0] for n in names], return_seams=True)
    for n, img, seam in zip(names, imgs, seams):
        vec = load_golden(n)[1]
        assert np.array_equal(seam["coef"], vec["coef"]), n
        assert np.array_equal(img, vec["rgb"]), n


@pytest.mark.parametrize("ss,w,h,ri,q,sigma", [
    ("420", 640, 480, 0, 85, 12.0), ("444", 500, 375, 0, 92, 25.0), ("422", 777, 333, 0, 75, 12.0), ("grey", 900, 700, 0, 85, 20.0),
    ("420", 1920, 1080, 0, 85, 12.0), ("420", 1024, 768, 1500, 90, 30.0), ("440", 333, 999, 0, 85, 5.0), ("420", 800, 600, 0, 100, 60.0),
    ("420", 1200, 900, 0, 40, 0.0),
])
def test_sync_form_files_without_restart_markers(dec, dec_rm, ss, w, h, ri, q, sigma, monkeypatch, tune):
    from oracle import oracle
    from tools import synth
    tune("MJ_HUFFMAN", "sync")
    raw = synth.synth_jpeg(w * 7 + h, w, h, q, ss, ri, sigma)
    ref = oracle.decode(raw)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"])
    assert np.array_equal(img, ref["rgb"])
    assert np.array_equal(np.swapaxes(dec_rm.decode([raw])[0], 0, 1), ref["rgb"])


def test_sync_form_batch_and_damage(dec, monkeypatch, tune):
    """A batch of DRI-less files picks the form by itself (no env); damaged streams still end in CorruptedJpeg or an image."""
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import JpegError, parse_jpeg
    files = [synth.synth_jpeg(300 + i, 640 + 64 * (i % 5), 512 - 32 * (i % 3), 85, ("420", "444", "422")[i % 3], 0, 15.0) for i in range(48)]
    for f, img in zip(files, dec.decode(files)):
        assert np.array_equal(img, oracle.decode(f)["rgb"])
    # the same files with headers-only parsing: the GPU finds the end of each scan, then the chunks
    from pyjpegdecoder_amd import BatchDecoder
    d2 = BatchDecoder(device=0, segment="gpu", layout="rowmajor", gpu_segment_min_files=1)
    try:
        for f, img in zip(files[:12], d2.decode(files[:12])):
            assert np.array_equal(np.swapaxes(img, 0, 1), oracle.decode(f)["rgb"])
    finally:
        d2.close()
    tune("MJ_HUFFMAN", "sync")
    rng = np.random.default_rng(99)
    raw = files[0]
    sc = parse_jpeg(raw).scans[0]
    for _ in range(20):
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(sc.entropy_start, sc.entropy_end))] = int(rng.integers(0, 256))
        try:
            assert dec.decode([bytes(b)])[0].shape == oracle.decode(raw)["rgb"].shape
        except JpegError:
            pass
    assert np.array_equal(dec.decode([raw])[0], oracle.decode(raw)["rgb"])


def test_sync_form_piece_boundaries_inside_stuffing(dec, monkeypatch, tune):
    """Stage 0 of long segments works in 16 KiB pieces; a piece whose first byte is the 00 of an FF 00 pair must drop it
    on the word of the piece before.  A big, noisy, quality-100 file has hundreds of pieces and an 0xFF every ~100
    bytes: make sure some boundaries do split a pair, then require the reference's coefficients."""
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import parse_jpeg
    tune("MJ_HUFFMAN", "sync")
    hits = 0
    for seed in (1, 2, 3):
        raw = synth.synth_jpeg(900 + seed, 1600, 1200, 100, "420", 0, 70.0)
        sc = parse_jpeg(raw).scans[0]
        hits += sum(1 for pos in range(sc.entropy_start + 16384, sc.entropy_end, 16384) if raw[pos - 1] == 0xFF)
        ref = oracle.decode(raw)
        (img,), (seam,) = dec.decode([raw], return_seams=True)
        assert np.array_equal(seam["coef"], ref["coef"]), seed
        assert np.array_equal(img, ref["rgb"]), seed
    assert hits >= 1, "no piece boundary fell between an 0xFF and its stuffed zero: enlarge the test"


def test_progressive_randomised_sweep(dec, dec_rm):
    """30 Pillow-written progressive files over size x sampling x quality x restart rows (libjpeg's scan script with
    successive approximation): batches of mixed geometry, every pixel against the oracle, both layouts."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from oracle import oracle
    from tools import synth
    rng = np.random.default_rng(77)
    files = []
    for i in range(30):
        w, h = int(rng.integers(8, 300)), int(rng.integers(8, 220))
        rgb = synth.synth_rgb(2000 + i, w, h, float(rng.choice([0.0, 8.0, 30.0])))
        kw = dict(quality=int(rng.choice([30, 60, 85, 95])), progressive=True)
        grey = i % 7 == 3
        if not grey:
            kw["subsampling"] = int(rng.integers(0, 3))
        if i % 3 == 0:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        b = io.BytesIO()
        Image.fromarray(rgb[..., 1] if grey else rgb).save(b, "JPEG", **kw)
        files.append(b.getvalue())
    refs = [oracle.decode(f)["rgb"] for f in files]
    for i, (img, ref) in enumerate(zip(dec.decode(files), refs)):
        assert np.array_equal(img, ref), i
    for i, (img, ref) in enumerate(zip(dec_rm.decode(files), refs)):
        assert np.array_equal(np.swapaxes(img, 0, 1), ref), i


@pytest.mark.parametrize("form", ["levels", "two_row_bands", "general_walk", "general_walk_levels", "split_all", "split_all_three_rows", "split_none",
                                  "split_largest", "chunks", "chunks_small", "chunks_split_all"])
def test_progressive_launch_forms(dec, dec_rm, form, tune):
    """The progressive stage 1 has two walks (the stream walks of progressive_fast.hip; progressive.hip's general one) and two
    launch schedules (band pipeline; one launch per dependency level).  The default — stream walks, one MCU row per
    band — is what every other progressive test runs; here the other combinations decode every progressive fixture in one
    mixed batch, coefficient store included, plus larger random files with long EOB runs and restart intervals."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from oracle import oracle
    from tools import synth
    env = {"levels": {"MJ_PROG_BANDS": "0"}, "two_row_bands": {"MJ_PROG_ROWS": "2"}, "general_walk": {"MJ_PROG_FAST": "0"},
           "general_walk_levels": {"MJ_PROG_FAST": "0", "MJ_PROG_BANDS": "0"},
           # (round 4) every refining AC scan walked as scout + parts (by default only those with 1 KiB or more per band), or none
           "split_all": {"MJ_PROG_SPLIT": "2"}, "split_all_three_rows": {"MJ_PROG_SPLIT": "2", "MJ_PROG_ROWS": "3"},
           "split_none": {"MJ_PROG_SPLIT": "0"},
           # (round 5) only each image's largest refining scan split, two parts per band (what 700-1 100 files take)
           "split_largest": {"MJ_PROG_SPLIT": "3"},
           # (round 5) the first AC scans cut into self-synchronising chunks, one per lane (progressive_chunks.hip: what batches of
           # 512 files and more take), with the default 512-byte chunks, with tiny ones (many wrong guesses, many repairs), and
           # beside split refining scans
           "chunks": {"MJ_PROG_CHUNKS": "2"}, "chunks_small": {"MJ_PROG_CHUNKS": "2", "MJ_PROG_CHUNK": "128"},
           "chunks_split_all": {"MJ_PROG_CHUNKS": "2", "MJ_PROG_SPLIT": "2", "MJ_PROG_CHUNK": "256"}}[form]
    for k, v in env.items():
        tune(k, v)
    names = prog_names()
    imgs, seams = dec.decode([load_golden(n)[0] for n in names], return_seams=True)
    for n, img, seam in zip(names, imgs, seams):
        vec = load_golden(n)[1]
        assert np.array_equal(seam["coef"], vec["coef"]), (form, n)
        assert np.array_equal(img, vec["rgb"]), (form, n)
    files = []
    for i, (w, h, ss, q, sigma, rows) in enumerate([(640, 360, 2, 85, 25.0, 0), (333, 211, 1, 95, 40.0, 2), (257, 129, 0, 40, 0.0, 0),
                                                    (200, 300, 2, 70, 8.0, 1)]):
        kw = dict(quality=q, subsampling=ss, progressive=True)
        if rows:
            kw["restart_marker_rows"] = rows
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(4100 + i, w, h, sigma)).save(b, "JPEG", **kw)
        files.append(b.getvalue())
    refs = [oracle.decode(f)["rgb"] for f in files]
    for i, (img, ref) in enumerate(zip(dec.decode(files), refs)):
        assert np.array_equal(img, ref), (form, i)
    for i, (img, ref) in enumerate(zip(dec_rm.decode(files), refs)):
        assert np.array_equal(np.swapaxes(img, 0, 1), ref), (form, i)
    if form.startswith("chunks"):          # ... and on the device itself: no image may have needed the host layer's second decode
        from pyjpegdecoder_amd import _binding as B
        from pyjpegdecoder_amd.batch import prepare_batch
        for i, f in enumerate(files + [load_golden(n)[0] for n in names]):
            prep = prepare_batch([f], B.MJ_LAYOUT_XMAJOR, 0)
            plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": 1})
            try:
                assert plan.stage1_form() & B.MJ_FORM_COUNT_RESOLVED, (form, i)
                plan.execute()
                plan.sync()
                assert plan.read(rgb=False)["status"][0] == 0, (form, i)
            finally:
                plan.close()


def test_progressive_chunks_damaged_streams_end_like_the_wavefront_walks(dec, tune):
    """A damaged first AC scan in the chunked form: whatever the counting walks make of it — records that do not fit together
    (MJ_ST_UNCONVERGED: the host layer decodes the image again with the wavefront walks), a code that does not exist, a run past
    the block — the caller sees what it sees without chunks: the same exception class, or the same pixels."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from tools import synth
    from pyjpegdecoder_amd import JpegError, parse_jpeg
    b = io.BytesIO()
    Image.fromarray(synth.synth_rgb(7711, 320, 240, 20.0)).save(b, "JPEG", quality=85, subsampling=2, progressive=True)
    raw = b.getvalue()
    scans = [sc for sc in parse_jpeg(raw).scans if sc.spectral_start > 0 and sc.bit_high == 0]
    assert len(scans) >= 3
    rng = np.random.default_rng(31)

    def outcome():
        try:
            return ("image", dec.decode([bytes(buf)])[0])
        except JpegError as exc:
            return ("error", type(exc).__name__)
    for trial in range(16):
        buf = bytearray(raw)
        sc = scans[trial % len(scans)]
        for _ in range(int(rng.integers(1, 3))):
            at = int(rng.integers(sc.entropy_start + 4, sc.entropy_end - 4))
            v = int(rng.integers(0, 255))
            buf[at] = v if v != 0xFF and buf[at - 1] != 0xFF else buf[at]      # (no new markers: the segmentation stays the file's)
        tune("MJ_PROG_CHUNKS", "0")
        want = outcome()
        tune("MJ_PROG_CHUNKS", "2")
        tune("MJ_PROG_CHUNK", ("128", "512")[trial % 2])
        got = outcome()
        assert got[0] == want[0], (trial, got[0], want)
        if got[0] == "image":
            assert np.array_equal(got[1], want[1]), trial
        else:
            assert got[1] == want[1], (trial, got, want)


def test_progressive_batches_of_2048_files_take_the_chunked_first_scans(dec):
    """From 2 048 images on a progressive plan walks its first AC scans in chunks, one per lane, in front of the band pipeline
    (progressive_chunks.hip; below that the wavefront walks are faster: profiles/r05_progressive_chunks.txt).  2 048 small files
    (16 distinct ones, optimised tables each, one with restart intervals): the plan says so, every status is ok, every distinct
    image is the oracle's and every copy its first instance's; 2 047 files keep the wavefront walks."""
    Image = pytest.importorskip("PIL.Image")
    torch = pytest.importorskip("torch")
    import io
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, nd = 96, 64, 16
    raws = []
    for i in range(nd):
        kw = dict(quality=(60, 85, 95)[i % 3], subsampling=2, progressive=True)
        if i == 5:
            kw["restart_marker_rows"] = 1
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(91000 + i, W, H, 5.0 + 3 * (i % 5))).save(b, "JPEG", **kw)
        raws.append(b.getvalue())
    dev = torch.device("cuda", 0)
    for n, want_chunks in ((2048, True), (2047, False)):
        prep = prepare_batch([raws[i % nd] for i in range(n)], B.MJ_LAYOUT_XMAJOR, 0)
        d_blob = torch.from_numpy(prep.blob).to(dev)
        plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
        try:
            form = plan.stage1_form()
            assert form & 15 == B.MJ_FORM_SCANS and bool(form & B.MJ_FORM_COUNT_RESOLVED) == want_chunks, (n, form)
            d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
            plan.execute(0, d_rgb.data_ptr())
            plan.sync()
            assert not plan.read(rgb=False)["status"].any()
            imgs = d_rgb[:n // nd * nd * W * H * 3].view(n // nd, nd, W * H * 3)
            for i in range(nd):
                assert np.array_equal(imgs[0, i].cpu().numpy().reshape(W, H, 3), oracle.decode(raws[i])["rgb"]), (n, i)
            assert bool((imgs == imgs[0]).all()), n
        finally:
            plan.close()


# ---- round 2: boundary and fallbacks ---------------------------------------------------------------------------------
def test_one_shot_entry_point_through_ctypes(dec):
    """`mj_decode_baseline_batch` is what INTEGRATION.md's reference-side stub binds: call it directly (no Plan wrapper)
    on two fixtures of different layouts and hold pixels and coefficients to the reference's."""
    import ctypes
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    lib = B.load_library()
    for names in (["c1_64x64_444_pil"], ["128x64_420_dri3", "64x64_420_pil"]):
        raws = [load_golden(n)[0] for n in names]
        prep = prepare_batch(raws, B.MJ_LAYOUT_XMAJOR, 0)
        batch = prep.to_c()
        n_px = sum(w * h * nc for (w, h, nc) in prep.shapes)
        blocks = sum(int(d.mcu_count_h) * int(d.mcu_count_v) * (1 if d.ncomp == 1 else sum(d.hs[c] * d.vs[c] for c in range(3))) for d in prep.descs)
        rgb = np.zeros(n_px, dtype=np.uint8)
        coef = np.zeros((blocks, 64), dtype=np.int16)
        status = np.full(len(raws), -1, dtype=np.int32)
        rc = lib.mj_decode_baseline_batch(dec.ctx.handle, ctypes.byref(batch), rgb.ctypes.data, coef.ctypes.data, status.ctypes.data)
        assert rc == B.MJ_OK, lib.mj_last_error(dec.ctx.handle)
        assert not status.any()
        off = boff = 0
        for n, (w, h, nc) in zip(names, prep.shapes):
            vec = load_golden(n)[1]
            assert np.array_equal(rgb[off:off + w * h * nc].reshape(vec["rgb"].shape), vec["rgb"]), n
            nb = vec["coef"].shape[0]
            assert np.array_equal(coef[boff:boff + nb], vec["coef"]), n
            off += w * h * nc
            boff += nb


def test_files_without_restart_markers_at_size_every_image_settles(dec):
    """256 x 1080p written without restart markers (bench.py's `without_restart_markers` family; 16 distinct files): the plan takes
    the synchronisation form, EVERY image's status is ok — round 4 found a quarter of such a batch MJ_ST_UNCONVERGED after the four
    repair rounds queued until then (flat image regions re-synchronise badly: chains of five and six wrongly guessed entry states) —
    and every distinct image equals the oracle's, every replica its first instance."""
    torch = pytest.importorskip("torch")
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, nd, n = 1920, 1080, 16, 256
    blob, offs = synth.synth_batch(nd, 0, W, H, 85, "420", 0)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(nd)]
    dev = torch.device("cuda", 0)
    prep = prepare_batch([raws[i % nd] for i in range(n)], B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
    try:
        assert plan.stage1_form() & 15 == B.MJ_FORM_SYNC
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        plan.execute(0, d_rgb.data_ptr())
        plan.sync()
        status = plan.read(rgb=False)["status"]
        assert not status.any(), np.unique(status, return_counts=True)
        imgs = d_rgb.view(n, W * H * 3)
        for i, want in enumerate(oracle_rgb_all(raws)):
            assert np.array_equal(imgs[i].cpu().numpy().reshape(W, H, 3), want), i
        for k in range(1, n // nd):
            assert torch.equal(imgs[k * nd:(k + 1) * nd], imgs[:nd]), k
    finally:
        plan.close()


def test_sync_rounds_that_do_not_settle_fall_back_to_the_serial_walk(dec, monkeypatch, tune):
    """The synchronisation rounds are a fixed number, queued without a host round trip.  Make them fail on purpose — no
    run-up in front of the chunks, tiny chunks: nearly every guessed entry state is wrong and a chain of wrong guesses
    only gets one link shorter per round — and require (a) the device to notice (MJ_ST_UNCONVERGED, not a wrong image,
    not CorruptedJpeg), (b) the host layer to decode the file again without synchronisation, bit-exact."""
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    tune("MJ_HUFFMAN", "sync")
    tune("MJ_SYNC_WARM", "0")
    tune("MJ_SYNC_CHUNK", "256")
    tune("MJ_SYNC_ROUNDS", "0")          # (32 are queued by default; see test_files_without_restart_markers_at_size_every_image_settles)
    raw = synth.synth_jpeg(4242, 640, 480, 85, "420", 0, 12.0)
    prep = prepare_batch([raw], B.MJ_LAYOUT_XMAJOR, 0)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": 1})
    try:
        assert plan.stage1_form() & 15 == B.MJ_FORM_SYNC
        plan.execute()
        plan.sync()
        assert plan.read(rgb=False)["status"][0] == B.MJ_ST_UNCONVERGED
    finally:
        plan.close()
    ref = oracle.decode(raw)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"]) and np.array_equal(img, ref["rgb"])
    # a plan created with MJ_FLAG_NO_SYNC never takes the form, whatever the environment says
    prep = prepare_batch([raw], B.MJ_LAYOUT_XMAJOR, B.MJ_FLAG_NO_SYNC)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": 1})
    try:
        assert plan.stage1_form() & 15 != B.MJ_FORM_SYNC
    finally:
        plan.close()


def _sync_plan_statuses(dec, raws):
    """One plan over `raws`, one execute, no fallback layer: (stage-1 form, per-image statuses)."""
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    prep = prepare_batch(raws, B.MJ_LAYOUT_XMAJOR, 0)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": len(raws)})
    try:
        form = plan.stage1_form()
        plan.execute()
        plan.sync()
        return form, plan.read(rgb=False)["status"].copy()
    finally:
        plan.close()


@pytest.mark.parametrize("count,bits", [(None, None), ("resolved", "10"), ("resolved", "13"), ("classic", None)])
def test_sync_counting_walks_classic_and_resolved(dec, count, bits, tune):
    """The counting walks of the synchronisation form exist twice: the round-3 kernel with its repair rounds (MJ_SYNC_COUNT=classic:
    what batches with more than eight tables, or a table in both roles, still take) and the walk on resolved tables with a repair
    work list (round 5, the default).  The same mixed batch of files without restart markers — three sampling layouts in one
    plan each — through both, with small chunks (many chunks per
    file, many wrong guesses) and every index width: every status ok without the fallback layer, every image the oracle's."""
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    tune("MJ_HUFFMAN", "sync")
    tune("MJ_SYNC_CHUNK", "256")
    tune("MJ_SYNC_COUNT", count)
    tune("MJ_SYNC_BITS", bits)
    for k, ss in enumerate(("420", "444", "grey", "422")):                 # (a plan holds one sampling layout)
        raws = [synth.synth_jpeg(900 + 10 * k + i, 320 + 48 * i, 240 + 16 * (i % 3), (85, 60, 95)[i % 3], ss, 0, 10.0 + 5 * (i % 3)) for i in range(4)]
        form, status = _sync_plan_statuses(dec, raws)
        assert form & 15 == B.MJ_FORM_SYNC
        assert bool(form & B.MJ_FORM_COUNT_RESOLVED) == (count != "classic")
        assert not status.any(), (ss, status)
        for i, (raw, img) in enumerate(zip(raws, dec.decode(raws))):
            assert np.array_equal(img, oracle.decode(raw)["rgb"]), (ss, i)


def test_sync_repair_lanes_walk_on_where_an_exit_state_changes(dec, tune):
    """No run-up at all and tiny chunks: nearly every chunk's entry state is guessed wrong, and a quarter of the old walks had not
    found their way by the end of their chunk either — the repair launch's lanes then start from a state that is itself wrong, and
    the lane in front, whose chunk now leaves in another state, has to walk on into theirs (huffman_sync.hip: k_count<true>).
    Every image still settles inside one execute (statuses ok without the fallback layer) and equals the oracle's; with the walk-on
    switched off (MJ_SYNC_ROUNDS=1: one chunk per lane) the device reports what did not settle instead of a wrong image."""
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    tune("MJ_HUFFMAN", "sync")
    tune("MJ_SYNC_CHUNK", "256")
    tune("MJ_SYNC_WARM", "0")
    raws = [synth.synth_jpeg(7000 + i, 640, 480, 85, "420", 0, 12.0) for i in range(6)]
    form, status = _sync_plan_statuses(dec, raws)
    assert form & B.MJ_FORM_COUNT_RESOLVED
    assert not status.any(), status
    for i, (raw, img) in enumerate(zip(raws, dec.decode(raws))):
        assert np.array_equal(img, oracle.decode(raw)["rgb"]), i
    tune("MJ_SYNC_ROUNDS", "1")
    _, status = _sync_plan_statuses(dec, raws)
    assert set(np.unique(status)) <= {0, B.MJ_ST_UNCONVERGED} and (status == B.MJ_ST_UNCONVERGED).any(), status
    for i, (raw, img) in enumerate(zip(raws, dec.decode(raws))):          # ... and the host layer decodes those again, exactly
        assert np.array_equal(img, oracle.decode(raw)["rgb"]), i


def test_sync_form_plans_on_concurrent_streams_always_settle(dec):
    """Several small plans of files without restart markers, each on a stream of its own, executed side by side again and again
    (a sharded queue's ranks sharing a GPU do exactly this).  Small batches mean 256-byte chunks with a 128-byte run-up: a
    third of the guesses are wrong and in every image a few old walks had not found their way by the end of their chunk, so the
    repair launch's lanes walk on into chunks that other lanes — started from a stale state — are writing too.  Which write
    stands must not depend on when the workgroups of the launch get to run (huffman_sync.hip: the walk from furthest left takes
    the chunk): every execute of every plan has to settle on the device — no MJ_ST_UNCONVERGED, no fallback — and give the
    oracle's pixels."""
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    dev = torch.device("cuda", 0)
    plans = []
    try:
        for k in range(4):
            raws = [synth.synth_jpeg(8800 + 10 * k + i, 480 + 32 * k, 320 + 16 * i, 85, "420", 0, 14.0) for i in range(6)]
            prep = prepare_batch(raws, B.MJ_LAYOUT_XMAJOR, 0)
            d_blob = torch.from_numpy(prep.blob).to(dev)
            plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(raws), "blob": d_blob})
            assert plan.stage1_form() & 15 == B.MJ_FORM_SYNC and plan.stage1_form() & B.MJ_FORM_COUNT_RESOLVED
            plans.append((plan, raws, torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev), torch.cuda.Stream(device=dev)))
        for it in range(60):
            for plan, _, rgb, st in plans:
                plan.execute(st.cuda_stream, rgb.data_ptr())
            for plan, _, _, _ in plans:
                plan.sync()
            for k, (plan, _, _, _) in enumerate(plans):
                status = plan.read(rgb=False)["status"]
                assert not status.any(), (it, k, status)
        for plan, raws, rgb, _ in plans:
            flat, off = rgb.cpu().numpy(), 0
            for raw in raws:
                want = oracle.decode(raw)["rgb"]
                assert np.array_equal(flat[off:off + want.size].reshape(want.shape), want)
                off += want.size
    finally:
        for plan, _, _, _ in plans:
            plan.close()


@pytest.mark.parametrize("layout", ["planar", "planar_rowmajor"])
def test_planar_layouts(layout):
    """MJ_LAYOUT_PLANAR_*: the components of the reference's image_array (:1373-1386) as three planes per image — every
    fixture (all sampling layouts, greyscale, odd sizes), one mixed call, host arrays and device tensors alike."""
    from pyjpegdecoder_amd import BatchDecoder
    names = golden_names() + prog_names()[:2]
    raws = [load_golden(n)[0] for n in names]
    d = BatchDecoder(device=0, layout=layout)
    try:
        for n, img in zip(names, d.decode(raws)):
            ref = load_golden(n)[1]["rgb"]                      # (W, H, 3) or (W, H)
            if layout == "planar_rowmajor":
                ref = np.swapaxes(ref, 0, 1)
            want = np.moveaxis(ref, 2, 0) if ref.ndim == 3 else ref
            assert img.shape == want.shape and np.array_equal(img, want), n
        torch = pytest.importorskip("torch")
        outs = d.decode_device(raws[:6])
        for n, t in zip(names[:6], outs):
            ref = load_golden(n)[1]["rgb"]
            if layout == "planar_rowmajor":
                ref = np.swapaxes(ref, 0, 1)
            want = np.moveaxis(ref, 2, 0) if ref.ndim == 3 else ref
            assert np.array_equal(t.cpu().numpy(), want), n
    finally:
        d.close()


@pytest.mark.parametrize("n", [1024, 256])
def test_config5_progressive_1080p_batch_at_size(dec, n):
    """BASELINE configs[4] at its stated size: 1024 x 1080p 4:2:0 progressive (libjpeg's default 10-scan script as Pillow
    writes it; 8 distinct files tiled).  Every distinct image against the oracle, every replica against its first instance.
    256 files: the size at which the plan walks the luma refinements as scout + parts (round 4; at 1024 the chip has no wave
    slots to spare for that and every scan is walked once)."""
    import io
    torch = pytest.importorskip("torch")
    Image = pytest.importorskip("PIL.Image")
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, nd = 1920, 1080, 8
    raws = []
    for i in range(nd):
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(7700 + i, W, H)).save(b, "JPEG", quality=85, subsampling=2, progressive=True)
        raws.append(b.getvalue())
    files = [raws[i % nd] for i in range(n)]
    dev = torch.device("cuda", 0)
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
    try:
        assert plan.stage1_form() & 15 == B.MJ_FORM_SCANS
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        plan.execute(0, d_rgb.data_ptr())
        plan.sync()
        assert not plan.read(rgb=False)["status"].any()
        per = W * H * 3
        imgs = d_rgb.view(n, per)
        for i, want in enumerate(oracle_rgb_all(raws)):
            assert np.array_equal(imgs[i].cpu().numpy().reshape(W, H, 3), want), i
        first = imgs[:nd]
        for k in range(1, n // nd):
            assert torch.equal(imgs[k * nd:(k + 1) * nd], first), k
    finally:
        plan.close()


def test_config4_sharded_queue_two_ranks_on_one_gpu():
    """BASELINE configs[3]'s driver — `bench.py --total-images N`: one job sharded over the ranks, a per-GPU image queue on
    each, timings aggregated over gloo — run here as two processes sharing cuda:0 on a small job."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--share-gpu", "--total-images", "50", "--queue-batch", "8",
           "--distinct", "6", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["images_all_gpus"] == 50 and line["config"]["images_per_gpu"] == 25
    assert line["parity"].startswith("bit-exact")
    assert abs(line["value"] - 50 * 1920 * 1080 / 1e6 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]      # whole job / time per pass


def test_bench_gpus_flag_starts_the_ranks_itself():
    """Plain `python bench.py --gpus 2` (no launcher around it, the way the driver calls N = 1): the flag alone must produce two
    ranks — both on cuda:0 here (--share-gpu) — and a line that says n_gpus = 2."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "64",
           "--distinct", "8", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                          # rank 0's line and nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["images_per_gpu"] == 64 and line["parity"].startswith("bit-exact")


def _idct_blocks_through_stage2(dec, blocks_xy, layout, cols=2048):
    """Dequantised int16 blocks [n,8,8] ([x,y]) as the coefficients of a greyscale image with an all-ones quantisation table
    through Plan.execute_stage2; returns the IDCT seam [n,8,8] and the level counts."""
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd._parse import ZZ_GRID
    import ctypes
    n = blocks_xy.shape[0]
    rows = -(-n // cols)
    zz = np.zeros((rows * cols, 64), dtype=np.int16)
    for x in range(8):
        for y in range(8):
            zz[:n, ZZ_GRID[y, x]] = blocks_xy[:, x, y]
    d = (B.ImageDescC * 1)()
    d[0].width, d[0].height, d[0].ncomp = 8 * cols, 8 * rows, 1
    d[0].hs[0] = d[0].vs[0] = 1
    d[0].mcu_count_h, d[0].mcu_count_v = cols, rows
    d[0].n_segments = 1
    qt = np.ones((1, 64), dtype=np.uint16)
    bc = B.BatchC()
    bc.n_images = 1; bc.images = ctypes.cast(d, ctypes.POINTER(B.ImageDescC))
    bc.blob = None; bc.blob_mem = B.MJ_MEM_NONE
    bc.n_qt = 1; bc.qt = qt.ctypes.data
    bc.layout = B.MJ_LAYOUT_ROWMAJOR if layout == "rowmajor" else B.MJ_LAYOUT_XMAJOR
    bc.flags = B.MJ_FLAG_KEEP_IDCT
    plan = B.Plan(dec.ctx, bc, {"n_images": 1, "keep": (d, qt)})
    try:
        plan.write_coef(zz)
        plan.execute_stage2()
        plan.sync()
        out = plan.read(rgb=False, idct=True)
        levels = plan.idct_levels()
    finally:
        plan.close()
    return out["idct"].reshape(rows * cols, 8, 8)[:n], levels


@pytest.mark.parametrize("layout", ["xmajor", "rowmajor"])
def test_idct_random_blocks_attack_the_fp32_bound(dec, layout):
    """The fast stage 2 accepts an fp32 IDCT sample when it is at least 2.0e-7 * A + 1e-6 away from a half-integer (A = sum
    of |dequantised coefficient|; DESIGN.md section 3 argues the bound).  This drives more than a million arbitrary int16
    blocks through Plan.execute_stage2 against oracle.idct_xy (jpeg_decoder.py:1561-1573 restated): dense uniform, heavy
    tailed, one and two coefficients up to the int16 limit, and the golden exact-tie blocks perturbed and scaled — every
    sample must be the reference's, and the number of blocks each level passed on is printed."""
    from oracle import oracle
    rng = np.random.default_rng(20261003 + (layout == "rowmajor"))
    n = 131072
    fams = {}
    fams["dense uniform +-1000"] = rng.integers(-1000, 1001, size=(n, 8, 8))
    lap = rng.laplace(0.0, 40.0, size=(n, 8, 8))
    lap[rng.random((n, 8, 8)) < 0.01] *= 40                                   # a few very large ones
    fams["heavy tailed"] = np.clip(np.rint(lap), -12000, 12000)
    one = np.zeros((n, 8, 8))
    idx = rng.integers(0, 64, size=n)
    one.reshape(n, 64)[np.arange(n), idx] = rng.integers(-32767, 32768, size=n)
    two = one.copy() // 2
    idx2 = rng.integers(0, 64, size=n)
    two.reshape(n, 64)[np.arange(n), idx2] += rng.integers(-16000, 16001, size=n)
    fams["one coefficient up to the int16 limit"] = one
    fams["two coefficients"] = two
    g = np.load(GOLDEN / "idct_blocks.npz")["blocks"].astype(np.int64)          # exact ties of the reference (F6/F7)
    tie = g[rng.integers(0, g.shape[0], size=n)].copy()
    tie[:, 0, 0] += 8 * rng.integers(-100, 101, size=n)                        # ~ integer shift of every sample: still next to the tie
    pert = rng.random(n) < 0.5
    k = rng.integers(1, 64, size=n)
    tie.reshape(n, 64)[np.arange(n)[pert], k[pert]] += rng.integers(-2, 3, size=int(pert.sum()))
    fams["golden tie blocks shifted and perturbed"] = tie
    scaled = g[rng.integers(0, g.shape[0], size=n)] * rng.choice([3, 5, 7, 9, 11, 25, 101], size=n)[:, None, None]
    fams["golden tie blocks scaled by odd factors"] = np.clip(scaled, -32767, 32767)
    mixed = np.clip(np.rint(rng.laplace(0.0, 12.0, size=(n, 8, 8))), -2000, 2000)
    mixed[:, 4:, :] = 0; mixed[:, :, 4:] = 0                                   # low-frequency blocks: where real files live
    mixed += tie * (rng.random(n) < 0.3)[:, None, None]
    fams["low-frequency noise, a third of them on top of a tie block"] = mixed
    total = 0
    for name, blocks in fams.items():
        blocks = blocks.astype(np.int16)
        got, (nb, l2, l3) = _idct_blocks_through_stage2(dec, blocks, layout)
        want = oracle.idct_xy(blocks)
        bad = np.flatnonzero((got != want).any(axis=(1, 2)))
        print(f"[{layout}] {name}: {blocks.shape[0]} blocks, {l2} to the fp64 level ({100.0 * l2 / max(nb, 1):.2f} %), {l3} to the exact-order routine")
        assert bad.size == 0, f"{name}: {bad.size} blocks differ, first {bad[0]}: {blocks[bad[0]].tolist()}"
        total += blocks.shape[0]
    assert total >= 900000


def test_colour_lattice_around_the_green_patch_threshold(dec):
    """YCbCr -> RGB (jpeg_decoder.py:1689-1700) on a lattice chosen against the fast path's decisions: every (Cb, Cr) in
    [-260, 260]^2 whose green numerator 17207 cb + 35707 cr leaves a remainder within 2 of +-25000 (the patch threshold of
    reconstruct_fast.hip), plus |c| in {124..126, 249..251} against everything, each with several Y.  4:4:4 DC-only blocks
    (every sample of a block = the wanted value); compared with the oracle's float64 expression."""
    from pyjpegdecoder_amd import _binding as B
    from oracle import oracle
    import ctypes
    cb, cr = np.meshgrid(np.arange(-260, 261), np.arange(-260, 261), indexing="ij")
    nnum = 17207 * cb + 35707 * cr
    rem = np.abs(((nnum + 25000) % 50000) - 25000)                            # distance of the remainder from 0 ... 25000
    near = rem >= 24998
    edge = np.isin(np.abs(cb), [124, 125, 126, 249, 250, 251]) | np.isin(np.abs(cr), [124, 125, 126, 249, 250, 251])
    sel = near | edge
    cbs, crs = cb[sel], cr[sel]
    ys = np.array([0, 1, 17, 128, 200, 254, 255, 300, -20])
    ycc = np.stack([np.repeat(ys, cbs.size), np.tile(cbs + 128, ys.size), np.tile(crs + 128, ys.size)], axis=1).astype(np.int32)
    n = ycc.shape[0]
    cols = 1024
    rows = -(-n // cols)
    coef = np.zeros((rows * cols, 3, 64), dtype=np.int16)
    coef[:n, :, 0] = (ycc - 128) * 8
    d = (B.ImageDescC * 1)()
    d[0].width, d[0].height, d[0].ncomp = 8 * cols, 8 * rows, 3
    for c in range(3):
        d[0].hs[c] = d[0].vs[c] = 1
    d[0].mcu_count_h, d[0].mcu_count_v = cols, rows
    d[0].n_segments = 1
    qt = np.ones((1, 64), dtype=np.uint16)
    bc = B.BatchC()
    bc.n_images = 1; bc.images = ctypes.cast(d, ctypes.POINTER(B.ImageDescC))
    bc.blob = None; bc.blob_mem = B.MJ_MEM_NONE
    bc.n_qt = 1; bc.qt = qt.ctypes.data
    bc.layout = B.MJ_LAYOUT_XMAJOR
    plan = B.Plan(dec.ctx, bc, {"n_images": 1, "keep": (d, qt)})
    try:
        plan.write_coef(coef.reshape(-1, 64))
        plan.execute_stage2()
        plan.sync()
        rgb = plan.read(rgb=True)["rgb"]
    finally:
        plan.close()
    img = np.asarray(rgb, dtype=np.uint8).reshape(8 * cols, 8 * rows, 3)          # x-major
    got = img[::8, ::8].transpose(1, 0, 2).reshape(rows * cols, 3)[:n]            # block (row r, col c) = index r * cols + c
    want = oracle.ycbcr_to_rgb(ycc.reshape(-1, 1, 3).astype(np.int16)).reshape(-1, 3)
    bad = np.flatnonzero((got != want).any(axis=1))
    print(f"colour lattice: {n} (Y, Cb, Cr) triples, {int(near.sum())} chroma pairs at the green threshold, {int(edge.sum())} at the B/R edges")
    assert bad.size == 0, f"{bad.size} differ, first: ycc {ycc[bad[0]].tolist()} got {got[bad[0]].tolist()} want {want[bad[0]].tolist()}"


@pytest.mark.parametrize("order", ["striped", "binned", "blob"])
def test_mixed_content_batch_at_size(dec, order, monkeypatch, tune):
    """1024 x 1080p DRI files of MIXED content (bench.py's `mixed_content`: quality 50..95, noise 0..80 above / below a random
    split row, so restart segments differ several-fold in bits), through the lane form with its segments dealt out by length
    in each of the three orders (MJ_SEG_ORDER): all 64 distinct files against the oracle, every replica against its first
    instance."""
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, nd, n = 1920, 1080, 64, 1024
    blob, offs = synth.synth_mixed_batch(nd, 424200, W, H, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(nd)]
    files = [raws[i % nd] for i in range(n)]
    tune("MJ_SEG_ORDER", order)
    dev = torch.device("cuda", 0)
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
    try:
        assert plan.stage1_form() & 15 == B.MJ_FORM_LANES
        d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        plan.execute(0, d_rgb.data_ptr())
        plan.sync()
        assert not plan.read(rgb=False)["status"].any()
        per = W * H * 3
        imgs = d_rgb.view(n, per)
        for i, want in enumerate(oracle_rgb_all(raws)):
            assert np.array_equal(imgs[i].cpu().numpy().reshape(W, H, 3), want), i
        first = imgs[:nd]
        for k in range(1, n // nd):
            assert torch.equal(imgs[k * nd:(k + 1) * nd], first), k
    finally:
        plan.close()


def test_config4_one_ranks_share_at_size():
    """BASELINE configs[3], one GPU's part at its stated size: 1 250 of the job's 10 000 1080p files through `bench.py
    --total-images 1250` (the per-GPU image queue with its default plan size), first and last image against the oracle."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, str(ROOT / "bench.py"), "--total-images", "1250", "--distinct", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["images_per_gpu"] == 1250
    assert line["parity"].startswith("bit-exact"), line["parity"]
    assert line["value"] > 0


def _with_ac_table(base, bits, vals):
    """`base` (a greyscale baseline file) with its AC table 0 replaced by (bits, vals) — DHT payloads may hold several tables."""
    out, pos = bytearray(base[:2]), 2
    while True:
        assert base[pos] == 0xFF
        marker, ln = base[pos + 1], int.from_bytes(base[pos + 2:pos + 4], "big")
        body = base[pos + 4:pos + 2 + ln]
        if marker == 0xC4:
            tabs, q = [], 0
            while q < len(body):
                n = sum(body[q + 1:q + 17])
                tabs.append(body[q:q + 17 + n])
                q += 17 + n
            body = b"".join(bytes([0x10]) + bytes(bits) + bytes(vals) if t[0] == 0x10 else t for t in tabs)
        out += bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + body
        pos += 2 + ln
        if marker == 0xDA:
            return bytes(out) + base[pos:]


def _craft_grey_stream(width, blocks, ac_table=None, pad=0):
    """A greyscale baseline file of `width` x 8 whose entropy-coded data is written here symbol by symbol: `blocks` = per block
    the list of AC (run, size, value-bits or None) symbols after a zero DC difference; value-bits None = the symbol's value
    bits are NOT written (what follows a run past index 63, jpeg_decoder.py:855-856).  `ac_table` = (bits, vals) replaces the
    file's AC table (symbols the Annex-K table does not have: sizes above 10)."""
    from pyjpegdecoder_amd import parse_jpeg
    from tools import synth
    base = synth.synth_jpeg(3, width, 8, 85, "grey", 0)
    if ac_table is not None:
        base = _with_ac_table(base, *ac_table)
    p = parse_jpeg(base)
    sc = p.scans[0]

    def codes(t):
        out, code, k = {}, 0, 0
        for l in range(1, 17):
            code <<= 1
            for _ in range(int(t.bits[l - 1])):
                out[int(t.vals[k])] = (code, l)
                k += 1
                code += 1
        return out
    dc, ac = codes(sc.huffman[0x00]), codes(sc.huffman[0x10])
    bits = []

    def put(value, n):
        bits.extend((value >> (n - 1 - i)) & 1 for i in range(n))
    for syms in blocks:
        put(*dc[0])
        for run, size, vbits in syms:
            put(*ac[(run << 4) | size])
            if vbits is not None and size:
                put(vbits, size)
    while len(bits) % 8:
        bits.append(1)
    data = bytearray()
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        data.append(b)
        if b == 0xFF:
            data.append(0x00)
    head = base[:sc.entropy_start]
    if pad:                                        # a comment segment of `pad` bytes behind SOI: moves the entropy-coded data's alignment
        head = head[:2] + b"\xFF\xFE" + (2 + pad).to_bytes(2, "big") + b"x" * pad + head[2:]
    return head + bytes(data) + b"\xFF\xD9"


@pytest.mark.parametrize("form", ["lanes", "lanes11", "wave", "sync"])
def test_runs_past_the_block_and_coefficient_63(dec, form, monkeypatch, tune):
    """The reference ends a block when a run carries the index to 64 or beyond and leaves that symbol's value bits unread
    (jpeg_decoder.py:849, :855-856); a symbol that lands exactly on index 63 ends it too, after its value.  Hand-written
    streams drive both through every stage-1 form — for the resolved-table lane form that is its after-the-loop store of
    coefficient 63 and its parity-of-the-position correction, through a resolved entry, an entry whose value bits are taken
    arithmetically and a code from a second-level table — against the oracle."""
    from oracle import oracle
    one = (0, 1, 1)                                   # run 0, size 1, value +1
    blocks = [
        [one] * 62 + [(1, 1, None)],                  # 62 coefficients, then run 1: index 64 -> over, value bit unread (resolved entry)
        [one] * 62 + [(1, 10, None)],                 # ... through a 16-bit code (second-level table) whose 10 value bits stay unread
        [one] * 60 + [(3, 4, None)],                  # index 61 + 3 = 64
        [one] * 60 + [(2, 9, 0x1FF)],                 # index 61 + 2 = 63: the last coefficient, value bits read (arithmetic entry)
        [one] * 63,                                   # coefficient 63 straight from a resolved entry
        [(0, 2, 3), (0, 0, None)],                    # an ordinary block behind them: value 3, end of block
        [(15, 0, None)] * 3 + [(14, 1, 0)],           # three ZRLs, then run 14: index 1 + 48 + 14 = 63, value -1
        [(15, 0, None)] * 3 + [(15, 1, None)],        # ... run 15: index 64, over
        [(0, 3, 5), (0, 0, None)],
    ]
    raw = _craft_grey_stream(8 * len(blocks), blocks)
    ref = oracle.decode(raw)
    assert ref["coef"][4, 63] == 1 and ref["coef"][3, 63] == 511 and ref["coef"][6, 63] == -1 and ref["coef"][8, 1] == 5
    tune("MJ_HUFFMAN", form)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"])
    assert np.array_equal(img, ref["rgb"])


@pytest.mark.parametrize("form", ["lanes", "lanes11", "wave", "sync"])
def test_value_sizes_up_to_15_and_bursts_of_unresolved_entries(dec, form, tune):
    """The reference reads whatever size a symbol names (jpeg_decoder.py:862: sizes 11..15 are outside T.81's baseline tables but
    inside its code).  A table with every (run, size) up to size 15 and code lengths 2..16, and hand-written blocks whose
    symbols are ALL of the kind the resolved-table lane form cannot finish in its table — code + value bits beyond 13, codes
    of 14..16 bits — eight to sixty-two in a row, up to 31 bits each: the per-lane stream window has to keep up with a lane
    that takes a dword per symbol (huffman_lanes13.hip: the top-up margin), the arithmetic step has to EXTEND 15-bit values,
    and the forms have to agree with the oracle on the coefficients.  (Pixels are compared where the samples fit int16:
    jpeg_decoder.py:1573's cast is undefined beyond.)"""
    from oracle import oracle
    from tools import craft_jpeg
    bits, vals = craft_jpeg.wide_ac_table(5)
    length = {}
    k = 0
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            length[vals[k]] = l
            k += 1
    rng = np.random.default_rng(11)

    def burst(n, sizes, run=0):
        out = []
        for _ in range(n):
            s = int(rng.choice(sizes))
            mag = int(rng.integers(1 << (s - 1), 1 << s))
            v = mag if rng.random() < 0.5 else -mag
            out.append((run, s, v if v >= 0 else v + (1 << s) - 1))
        return out
    big = [s for s in range(11, 16)]
    # sizes whose code + value bits exceed 13 in this table (entries that are not resolved), by size
    open_sizes = [s for s in range(7, 16) if length[s] + s > 13]
    assert set(big) <= set(open_sizes)
    blocks = [
        burst(8, big) + [(0, 0, None)],                       # eight symbols of 11..15 value bits in a row, then end of block
        burst(62, big) + [(0, 1, 1)],                         # a whole block of them: 62 x up to 31 bits, coefficient 63 from a resolved entry
        burst(63, open_sizes),                                # ... ending on coefficient 63 with an unresolved entry
        [(0, 2, 3), (0, 0, None)],                            # an ordinary block behind them
        burst(20, [15]) + [(0, 0, None)],                     # twenty 15-bit values (EXTEND at its widest)
        burst(12, big, run=1) + [(0, 0, None)],               # with runs in between
        [(0, 1, 1)] * 5 + burst(9, big) + [(0, 1, 0)] * 5 + burst(9, big) + [(0, 0, None)],   # resolved and unresolved stretches alternate
        [(0, 3, 5), (0, 0, None)],
    ]
    raw = _craft_grey_stream(8 * len(blocks), blocks, ac_table=(bits, vals))
    ref = oracle.decode(raw)
    assert int(np.abs(ref["coef"].astype(np.int64)).max()) >= 16384 and ref["coef"][7, 1] == 5 and ref["coef"][3, 1] == 3
    tune("MJ_HUFFMAN", form)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"])
    small = [3, 7]                                            # blocks whose samples fit int16
    for b in small:
        assert np.array_equal(img[8 * b:8 * b + 8], ref["rgb"][8 * b:8 * b + 8]), b


@pytest.mark.parametrize("form,pad", [("lanes", 0), ("lanes", 1), ("lanes", 2), ("lanes", 3), ("wave", 0), ("sync", 0), ("sync", 1)])
def test_streams_dense_with_0xff_bytes(dec, form, pad, tune):
    """Stage 0 (csrc/destuff.hip) takes sixteen source bytes per lane and turn and assumes what entropy-coded data guarantees — an
    0xFF is followed by its stuffed 0x00 — with the general step behind it for turns that break the assumption or hold more than
    three dropped bytes in one lane's sixteen.  A hand-written stream whose value bits are runs of ones, in four stretches of
    different density: lanes with one, two and three stuffed bytes to take out in registers, turns that go the slow way, 0xFF as
    the last byte of a turn (the carry into the next) — of a fast turn and of a SLOW one, at every alignment of the data (`pad`:
    a slow turn that starts off a dword boundary ends in a step of one dword, and the state behind its last byte has to come
    through 63 empty lanes: bench.py's 256-image parity check found that one, round 5) —, through the restart-segment kernel
    (lanes, wave) and the piece-wise one (sync) — coefficients and pixels against the oracle."""
    from oracle import oracle
    from pyjpegdecoder_amd import parse_jpeg
    rng = np.random.default_rng(5)

    def block(density):
        out, k = [], 1
        while k < 60:
            if rng.random() < density:
                size = int(rng.integers(6, 11))
                out.append((0, size, (1 << size) - 1))
                k += 1
            else:
                run, size = int(rng.integers(0, 3)), int(rng.integers(1, 4))
                out.append((run, size, int(rng.integers(0, 1 << size))))
                k += run + 1
        return out + [(0, 0, None)]
    blocks = [block(d) for d in [0.02] * 250 + [0.15] * 250 + [0.9] * 100 + [0.05] * 200 + [0.4] * 400]
    raw = _craft_grey_stream(8 * len(blocks), blocks, pad=pad)
    sc = parse_jpeg(raw).scans[0]
    ff = np.frombuffer(raw[sc.entropy_start:sc.entropy_end], dtype=np.uint8) == 0xFF
    n = ff.size // 1024 * 1024
    per_lane = ff[:n].reshape(-1, 64, 16).sum(axis=2)                     # 0xFF bytes per lane and turn
    worst = per_lane.max(axis=1)
    last_ff = ff[1023:n:1024]                                             # an 0xFF as a turn's last byte
    assert (worst <= 3).sum() >= 8 and (worst > 3).sum() >= 8             # turns of both kinds
    assert all((per_lane[worst <= 3] == k).sum() >= 10 for k in (1, 2, 3))
    assert (last_ff & (worst <= 3)).sum() >= 1 and (last_ff & (worst > 4)).sum() >= 2, (last_ff & (worst <= 3)).sum()
    ref = oracle.decode(raw)
    tune("MJ_HUFFMAN", form)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], ref["coef"])
    assert np.array_equal(img, ref["rgb"])


def test_lane_form_waves_whose_lanes_use_different_ac_tables(dec, tune):
    """A batch of >= 1024 restart segments from files that assign the batch's three AC tables to their components in four
    different ways (the Annex-K pair as it is, swapped, and a third table on luma or on Cb): consecutive segments — the lanes of
    one wavefront of the resolved-table lane form — then read different tables for the same block of the MCU, so the table base
    must be a per-lane operand of the symbol loop (huffman_lanes13.hip: `lutb`), as must the canonical search's table.  Every
    file against the oracle, coefficients and pixels; value sizes up to 15 on the third table."""
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from tools import craft_jpeg
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    wide = craft_jpeg.wide_ac_table(9)
    maps = [((0, 0), (1, 1), (1, 1)), ((0, 1), (1, 0), (1, 0)), ((0, 2), (1, 0), (1, 1)), ((1, 1), (0, 2), (0, 0))]
    W, H = 256, 128                                           # 16 x 8 MCUs, restart interval 8: 16 segments per file
    raws = [craft_jpeg.craft_baseline(W, H, ((2, 2), (1, 1), (1, 1)), seed=100 + i, restart_interval=8, tables=maps[i % 4],
                                      ac_tables=[wide], max_size=6 if i % 8 < 4 else 12, density=0.3)
            for i in range(24)]
    files = [raws[(i * 7) % 24] for i in range(72)]           # 1152 segments; neighbours in the batch have different maps
    tune("MJ_HUFFMAN", "lanes")
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    dev = torch.device("cuda", 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(files)})
    try:
        form = plan.stage1_form()
        assert form & 15 == B.MJ_FORM_LANES and form & B.MJ_FORM_RESOLVED, form
        plan.execute(0, 0)
        plan.sync()
        out = plan.read(rgb=True, coef=True)
        assert not out["status"].any(), out["status"]
        refs = [oracle.decode(r) for r in raws]
        per_rgb, per_blk = W * H * 3, (W // 16) * (H // 16) * 6
        for i in range(len(files)):
            ref = refs[(i * 7) % 24]
            assert np.array_equal(out["coef"][i * per_blk:(i + 1) * per_blk], ref["coef"]), i
            if (i * 7) % 24 % 8 < 4:                          # (modest coefficients: samples fit int16, pixels are defined)
                assert np.array_equal(out["rgb"][i * per_rgb:(i + 1) * per_rgb].reshape(W, H, 3), ref["rgb"]), i
    finally:
        plan.close()


def test_default_decoder_takes_the_fast_host_route(monkeypatch):
    """BatchDecoder() as a user gets it: restart markers found on the GPU and — in decode_device — headers read and the batch
    assembled by the native front end (mj_host_assemble) for calls of 8 files or more, the Python marker loop for fewer and
    for what the front end declines (a progressive file among them); every image against the oracle either way."""
    torch = pytest.importorskip("torch")
    import io
    from PIL import Image
    from oracle import oracle
    from tools import synth
    from pyjpegdecoder_amd import BatchDecoder, batch as batch_mod
    calls = {"native": 0, "python": 0}
    real_native, real_parse = batch_mod.prepare_batch_native, batch_mod.parse_jpeg

    def native(*a, **kw):
        calls["native"] += 1
        return real_native(*a, **kw)

    def parse(*a, **kw):
        calls["python"] += 1
        return real_parse(*a, **kw)
    monkeypatch.setattr(batch_mod, "prepare_batch_native", native)
    monkeypatch.setattr(batch_mod, "parse_jpeg", parse)
    d = BatchDecoder(device=0)
    try:
        assert d.gpu_segment and d.native_host
        files = [synth.synth_jpeg(900 + i, 320, 240, 85, "420", 20 if i % 2 else 0) for i in range(12)]
        out = d.decode_device(files)
        assert calls["native"] >= 1 and calls["python"] == 0
        for f, t in zip(files, out):
            assert np.array_equal(t.cpu().numpy(), oracle.decode(f)["rgb"])
        host = d.decode(files)                                  # numpy out: headers in Python, markers on the GPU
        for f, a in zip(files, host):
            assert np.array_equal(a, oracle.decode(f)["rgb"])
        calls["native"] = calls["python"] = 0
        few = d.decode_device(files[:3])                        # a handful of files: segmented on the host
        assert calls["native"] == 0 and calls["python"] == 3
        for f, t in zip(files[:3], few):
            assert np.array_equal(t.cpu().numpy(), oracle.decode(f)["rgb"])
        b = io.BytesIO()
        Image.fromarray(synth.synth_rgb(77, 200, 120)).save(b, "JPEG", quality=80, progressive=True)
        mixed = files[:9] + [b.getvalue()]
        for f, t in zip(mixed, d.decode_device(mixed)):
            assert np.array_equal(t.cpu().numpy(), oracle.decode(f)["rgb"])
    finally:
        d.close()


# ---- stages 1 and 2 in one launch (fused.hip) ------------------------------------------------------------------------------
def _fused_batch(ss, W, H, n, distinct, seed, ri=None, gpu_segment=False, layout=None, flags=0):
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd import parse_jpeg
    from pyjpegdecoder_amd.batch import prepare_batch
    mw = 32 if ss == "411" else (16 if ss in ("420", "422") else 8)
    ri = (W + mw - 1) // mw if ri is None else ri
    blob, offs = synth.synth_batch(distinct, seed, W, H, 85, ss, ri)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    files = [raws[(5 * i + i // distinct) % distinct] for i in range(n)]
    layout = B.MJ_LAYOUT_XMAJOR if layout is None else layout
    parsed = [parse_jpeg(f, headers_only=True) for f in files] if gpu_segment else None
    return raws, files, prepare_batch(files, layout, flags, parsed)


def _decode_plan(ctx, prep, n, torch, opts=()):
    """(rgb on the device, statuses, stage1_form) of a plan's executes under library options: three of them — the second is
    captured into a graph, the third replays it — each into its own zeroed buffer and each with the coefficient store
    POISONED first (a fused launch's reconstruction wavefronts read what its decoder wavefronts have just written: a block
    read too early must not find the previous execute's copy of itself); the three outputs must be one."""
    from pyjpegdecoder_amd import _binding as B
    for k, v in opts:
        B.set_option(k, v)
    try:
        dev = torch.device("cuda", 0)
        d_blob = torch.from_numpy(prep.blob).to(dev)
        torch.cuda.synchronize()
        plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
        try:
            out = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
            keep = None
            for k, poison in enumerate((0x5A, 0xC3, 0x7E)):
                out.zero_()
                torch.cuda.synchronize()                # (the fill runs on torch's stream, the plan on the context's)
                plan.fill_coef(poison)
                plan.execute(0, out.data_ptr())
                plan.sync()
                if keep is None:
                    keep = out.clone()
                else:
                    assert torch.equal(out, keep), f"execute {k} of the plan differs from its first"
            return keep, plan.read(rgb=False)["status"], plan.stage1_form()
        finally:
            plan.close()
    finally:
        for k, _ in opts:
            B.set_option(k, None)


def test_fused_launch_config3_at_size_every_distinct_image(dec):
    """BASELINE configs[2] through the ONE launch mj_plan_execute makes of it (fused.hip: lane-walk wavefronts and
    reconstruction wavefronts in one workgroup per CU): 1024 x 1080p 4:2:0, one restart interval per MCU row.  Every distinct
    image against the oracle, every copy against its first instance, and the whole output against the two launches'."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    W, H, n, distinct = 1920, 1080, 1024, 256        # (256 distinct files, like bench.py: round 5's stage-0 slip showed in ONE of bench.py's 256 and in none of 64)
    raws, files, prep = _fused_batch("420", W, H, n, distinct, 515100)
    fused, st, form = _decode_plan(dec.ctx, prep, n, torch)
    assert form & B.MJ_FORM_FUSED and form & 15 == B.MJ_FORM_LANES, form
    assert not st.any()
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0")])
    assert not form2 & B.MJ_FORM_FUSED and not st2.any()
    per = W * H * 3
    differ = (fused.view(n, per) != two.view(n, per)).any(dim=1).nonzero().flatten().tolist()
    assert not differ, f"{len(differ)} images differ between the fused launch and the two launches, first {differ[:12]}"
    imgs = fused.view(n, per)
    first = {}
    for i in range(n):
        first.setdefault((5 * i + i // distinct) % distinct, i)
    for i in range(n):
        d = (5 * i + i // distinct) % distinct
        if first[d] != i:
            assert torch.equal(imgs[i], imgs[first[d]]), i
    for d, want in enumerate(oracle_rgb_all(raws)):
        assert np.array_equal(imgs[first[d]].cpu().numpy().reshape(W, H, 3), want), d


@pytest.mark.parametrize("ss,W,H,n,distinct", [("444", 640, 480, 1021, 12), ("422", 800, 608, 1300, 10), ("440", 512, 512, 700, 9), ("420", 1000, 700, 521, 7),
                                                ("420", 1920, 1080, 1250, 5), ("444", 264, 200, 1500, 6)])
def test_fused_launch_other_layouts_and_ragged_workgroups(dec, ss, W, H, n, distinct, tune):
    """The fused launch on 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 batches whose image count is no multiple of the images per workgroup
    (the last workgroup is short, some have one image), image sizes that are no multiple of the MCU or of a strip, with every
    number of consumer wavefronts the option allows: each against the two launches byte for byte, the distinct files against
    the oracle, with the restart markers found on the host and on the GPU."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    raws, files, prep = _fused_batch(ss, W, H, n, distinct, 77000 + W)
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0"), ("MJ_HUFFMAN", "lanes")])
    assert not st2.any() and not form2 & B.MJ_FORM_FUSED
    per = W * H * 3
    imgs = two.view(n, per)
    for d, want in enumerate(oracle_rgb_all(raws)):
        i = next(k for k in range(n) if (5 * k + k // distinct) % distinct == d)
        assert np.array_equal(imgs[i].cpu().numpy().reshape(W, H, 3), want), d
    for cons in (None, "1", "3", "8"):
        opts = [("MJ_HUFFMAN", "lanes")] + ([("MJ_FUSED_CONSUMERS", cons)] if cons else [])
        fused, st, form = _decode_plan(dec.ctx, prep, n, torch, opts)
        assert form & B.MJ_FORM_FUSED, (form, cons)
        assert not st.any() and torch.equal(fused, two), cons
    _, _, prep_g = _fused_batch(ss, W, H, n, distinct, 77000 + W, gpu_segment=True)
    fused, st, form = _decode_plan(dec.ctx, prep_g, n, torch, [("MJ_HUFFMAN", "lanes")])
    assert form & B.MJ_FORM_FUSED and not st.any() and torch.equal(fused, two)


@pytest.mark.parametrize("ss,W,H,n,distinct", [("420", 1920, 1080, 1024, 16), ("420", 1000, 700, 521, 7), ("444", 640, 480, 1021, 12), ("422", 800, 608, 1300, 10),
                                                ("440", 512, 512, 700, 9), ("420", 1920, 1080, 1250, 5)])
def test_fused_launch_row_major(dec, ss, W, H, n, distinct, tune):
    """Row-major plans (the strip worker runs on the transposed image): a fused launch's consumers take PIECES of MCU rows behind
    the one producer wave that holds the row.  Against the two launches byte for byte (poisoned coefficient store, three
    executes), the distinct files against the oracle, markers found on the host and on the GPU, 1..8 consumers."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    raws, files, prep = _fused_batch(ss, W, H, n, distinct, 88000 + W, layout=B.MJ_LAYOUT_ROWMAJOR)
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0"), ("MJ_HUFFMAN", "lanes")])
    assert not st2.any() and not form2 & B.MJ_FORM_FUSED
    per = W * H * 3
    imgs = two.view(n, per)
    for d, want in enumerate(oracle_rgb_all(raws)):
        i = next(k for k in range(n) if (5 * k + k // distinct) % distinct == d)
        assert np.array_equal(np.swapaxes(imgs[i].cpu().numpy().reshape(H, W, 3), 0, 1), want), d
    for cons in (None, "1", "5"):
        opts = [("MJ_HUFFMAN", "lanes")] + ([("MJ_FUSED_CONSUMERS", cons)] if cons else [])
        fused, st, form = _decode_plan(dec.ctx, prep, n, torch, opts)
        assert form & B.MJ_FORM_FUSED, (form, cons)
        assert not st.any() and torch.equal(fused, two), cons
    _, _, prep_g = _fused_batch(ss, W, H, n, distinct, 88000 + W, gpu_segment=True, layout=B.MJ_LAYOUT_ROWMAJOR)
    fused, st, form = _decode_plan(dec.ctx, prep_g, n, torch, [("MJ_HUFFMAN", "lanes")])
    assert form & B.MJ_FORM_FUSED and not st.any() and torch.equal(fused, two)


@pytest.mark.parametrize("layout", ["xmajor", "rowmajor"])
@pytest.mark.parametrize("W,H,n,distinct", [(1920, 1080, 1024, 32), (640, 480, 1021, 16)])
def test_fused_launch_segments_dealt_out_by_length(dec, W, H, n, distinct, layout, tune):
    """Files of MIXED content (restart segments that differ several-fold in length: the lane launch deals them out by length)
    through the fused launch's second form: one pool of jobs for the whole launch, the hand-off across workgroups and XCDs
    (write-through coefficient stores, progress words in global memory, an agent-scope acquire on the reading side).  Against
    the two launches byte for byte — coefficient store poisoned before each of three executes —, every distinct file against
    the oracle, 1..8 consumers."""
    torch = pytest.importorskip("torch")
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    blob, offs = synth.synth_mixed_batch(distinct, 313100 + W, W, H, "420", (W + 15) // 16)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    files = [raws[(5 * i + i // distinct) % distinct] for i in range(n)]
    lay = B.MJ_LAYOUT_XMAJOR if layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    prep = prepare_batch(files, lay, 0)
    base = [("MJ_HUFFMAN", "lanes"), ("MJ_SEG_ORDER", "striped")]
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, base + [("MJ_FUSED", "0")])
    assert not st2.any() and not form2 & B.MJ_FORM_FUSED
    per = W * H * 3
    imgs = two.view(n, per)
    for d, want in enumerate(oracle_rgb_all(raws)):
        i = next(k for k in range(n) if (5 * k + k // distinct) % distinct == d)
        got = imgs[i].cpu().numpy()
        got = got.reshape(W, H, 3) if layout == "xmajor" else np.swapaxes(got.reshape(H, W, 3), 0, 1)
        assert np.array_equal(got, want), d
    # (MJ_FUSED_PATIENCE: polls before a consumer gives a job up to the clean-up launch and ends — what keeps the launch from
    # waiting for a workgroup that another plan's kernel keeps off the chip.  0 = every job that is not ready at once, and every
    # ticket its wave would have drawn after it: nearly the whole batch goes through the clean-up launch; 3 = some of it)
    for extra in ([], [("MJ_FUSED_CONSUMERS", "1")], [("MJ_FUSED_CONSUMERS", "8")], [("MJ_FUSED_PATIENCE", "0")], [("MJ_FUSED_PATIENCE", "3")]):
        fused, st, form = _decode_plan(dec.ctx, prep, n, torch, base + extra)
        assert form & B.MJ_FORM_FUSED, (form, extra)
        assert not st.any(), np.unique(st)
        differ = (fused.view(n, per) != imgs).any(dim=1).nonzero().flatten().tolist()
        assert not differ, (extra, len(differ), differ[:10])


@pytest.mark.parametrize("mixed", [False, True])
def test_fused_launches_of_several_plans_at_once(dec, mixed, tune):
    """Three plans' fused launches in flight together, each on a stream of its own: a fused launch wants a CU's whole LDS, so
    the launches' workgroups find their CUs at different times.  Whole images per workgroup: no workgroup waits for another.
    Segments dealt out by length (mixed content): a consumer may wait for a workgroup that has not started — its patience is
    bounded, what it gives up goes through the clean-up launch, and every launch ends with the right pixels."""
    torch = pytest.importorskip("torch")
    from tools import synth
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, n, distinct = 1920, 1080, 1024, 12
    if mixed:
        blob, offs = synth.synth_mixed_batch(distinct, 424242, W, H, "420", 120)
    else:
        blob, offs = synth.synth_batch(distinct, 424242, W, H, 85, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(distinct)]
    files = [raws[(5 * i + i // distinct) % distinct] for i in range(n)]
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    ref, st, form = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0")])
    assert not st.any() and not form & B.MJ_FORM_FUSED
    dev = torch.device("cuda", 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    torch.cuda.synchronize()
    plans = [B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n}) for _ in range(3)]
    try:
        assert all(p.stage1_form() & B.MJ_FORM_FUSED for p in plans)
        outs = [torch.zeros(plans[0].info.rgb_bytes, dtype=torch.uint8, device=dev) for _ in plans]
        streams = [torch.cuda.Stream(device=dev) for _ in plans]
        torch.cuda.synchronize()
        for rnd in range(4):
            for p in plans:
                p.fill_coef(0x33 + rnd)
            for p, o, s_ in zip(plans, outs, streams):
                p.execute(s_.cuda_stream, o.data_ptr())
            for p in plans:
                p.sync()
            for k, (p, o) in enumerate(zip(plans, outs)):
                assert not p.read(rgb=False)["status"].any(), (rnd, k)
                assert torch.equal(o, ref), (rnd, k)
                o.zero_()
            torch.cuda.synchronize()
    finally:
        for p in plans:
            p.close()


def test_fused_launch_only_where_it_applies(dec, dec_rm, tune):
    """What a fused launch cannot take keeps the two launches (mj_plan_stage1_form says which): in x-major output a restart
    interval of three MCU rows (a column is only complete when every segment is in its last row — the last third of the walks:
    measured slower than the two launches, form_select.h) or unrelated to the row (25 MCUs on a row of 40), planar pixels, seam
    outputs, the exact-order stage 2, MJ_FUSED=0 — and whatever it is, the pixels are the oracle's.  (Round 5 also kept out half
    rows, two rows and every interval in row-major output: those fuse now, test_fused_launch_any_restart_interval.)"""
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from pyjpegdecoder_amd import _binding as B
    W, H, n, distinct = 640, 480, 800, 5
    for kind in ("three_rows", "odd_interval", "planar", "seams", "exact", "off"):
        ri = {"three_rows": 120, "odd_interval": 25}.get(kind)
        layout = B.MJ_LAYOUT_PLANAR_XMAJOR if kind == "planar" else None
        flags = {"seams": B.MJ_FLAG_KEEP_IDCT, "exact": B.MJ_FLAG_EXACT_ONLY}.get(kind, 0)
        raws, files, prep = _fused_batch("420", W, H, n, distinct, 31000, ri=ri, layout=layout, flags=flags)
        out, st, form = _decode_plan(dec.ctx, prep, n, torch, [("MJ_HUFFMAN", "lanes")] + ([("MJ_FUSED", "0")] if kind == "off" else []))
        assert not form & B.MJ_FORM_FUSED, (kind, form)
        assert not st.any()
        got = out[:W * H * 3].cpu().numpy()
        want = oracle.decode(files[0])["rgb"]
        got = np.moveaxis(got.reshape(3, W, H), 0, 2) if kind == "planar" else got.reshape(W, H, 3)
        assert np.array_equal(got, want), kind
    raws, files, prep = _fused_batch("420", W, H, n, distinct, 31000)
    _, _, form = _decode_plan(dec.ctx, prep, n, torch, [("MJ_HUFFMAN", "lanes")])
    assert form & B.MJ_FORM_FUSED


@pytest.mark.parametrize("layout,ss,W,H,n,distinct,ri", [
    ("xmajor", "420", 640, 480, 800, 5, 20),          # half an MCU row per segment (round 5: two launches)
    ("xmajor", "444", 640, 480, 700, 6, 20),          # a quarter of a row; 240 segments per image: two passes on 256 CUs
    ("xmajor", "422", 800, 608, 900, 7, 10),          # a fifth of a row: 380 segments per image, one image per pass, four passes
    ("xmajor", "420", 640, 480, 800, 5, 80),          # TWO rows per segment: a column is complete in the second half of the walks
    ("xmajor", "444", 328, 200, 900, 6, 82),          # ... 41 MCUs per row, 25 rows: the last segment holds one row
    ("xmajor", "440", 640, 480, 800, 5, 240),         # ... three rows: the two launches
    ("rowmajor", "420", 640, 480, 800, 5, 20),        # row-major: pieces of rows behind half-row segments
    ("rowmajor", "420", 640, 480, 800, 5, 80),        # ... two rows per segment
    ("rowmajor", "444", 640, 480, 600, 6, 25),        # ... an interval with no relation to the row (80 MCUs): pieces straddle segments
    ("rowmajor", "440", 512, 512, 700, 9, 100),       # ... longer than a row and not a multiple of it; the last segment short
    ("rowmajor", "422", 800, 608, 650, 4, 8),         # ... shorter than a piece (16 MCUs): several segments per piece; 475 per image, three passes
    ("rowmajor", "422", 800, 608, 650, 4, 7),         # ... 543 segments per image: more than a workgroup's producers have lanes -> the two launches
    ("xmajor", "411", 640, 480, 600, 5, 20),          # 4:1:1 (32-pixel MCUs: 20 per row), one row per segment — round 6: fused as well
    ("rowmajor", "411", 640, 480, 600, 5, 40),        # ... row-major 4:1:1 keeps the two launches (form_select.h)
])
def test_fused_launch_any_restart_interval(dec, layout, ss, W, H, n, distinct, ri, tune):
    """Round 6: the fused launch no longer needs one restart interval per MCU row.  A consumer's job is ready when the producer
    waves that hold its MCUs are past them, worked out per MCU (fused.hip: JobGeo): x-major plans take every interval that
    divides the row and the interval of two rows, row-major plans any.  Against the two launches byte for byte (coefficient store poisoned, three executes),
    every distinct file against the oracle, 1 / 8 consumers, markers found on the host and on the GPU."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    lay = B.MJ_LAYOUT_XMAJOR if layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    raws, files, prep = _fused_batch(ss, W, H, n, distinct, 99000 + W + ri, ri=ri, layout=lay)
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0"), ("MJ_HUFFMAN", "lanes")])
    assert not st2.any() and not form2 & B.MJ_FORM_FUSED
    per = W * H * 3
    imgs = two.view(n, per)
    for d, want in enumerate(oracle_rgb_all(raws)):
        i = next(k for k in range(n) if (5 * k + k // distinct) % distinct == d)
        got = imgs[i].cpu().numpy()
        got = got.reshape(W, H, 3) if layout == "xmajor" else np.swapaxes(got.reshape(H, W, 3), 0, 1)
        assert np.array_equal(got, want), d
    mw, mh = (32 if ss == "411" else (16 if ss in ("420", "422") else 8)), (16 if ss in ("420", "440") else 8)
    spi = -(-(-(-W // mw) * -(-H // mh)) // ri)
    mpr = -(-W // mw)
    want_fused = spi <= 512 and not (ss == "411" and layout == "rowmajor")     # (4:1:1 transposed: 13 KB strips, no room beside the walk)
    want_fused = want_fused and (layout == "rowmajor" or mpr % ri == 0 or ri == 2 * mpr)
    for cons in (None, "1", "8"):
        opts = [("MJ_HUFFMAN", "lanes")] + ([("MJ_FUSED_CONSUMERS", cons)] if cons else [])
        fused, st, form = _decode_plan(dec.ctx, prep, n, torch, opts)
        assert bool(form & B.MJ_FORM_FUSED) == want_fused, (form, cons, spi)
        assert not st.any() and torch.equal(fused, two), cons
    _, _, prep_g = _fused_batch(ss, W, H, n, distinct, 99000 + W + ri, ri=ri, gpu_segment=True, layout=lay)
    fused, st, form = _decode_plan(dec.ctx, prep_g, n, torch, [("MJ_HUFFMAN", "lanes")])
    assert bool(form & B.MJ_FORM_FUSED) == want_fused and not st.any() and torch.equal(fused, two)


@pytest.mark.parametrize("layout", ["xmajor", "rowmajor"])
@pytest.mark.parametrize("ss,W,H,n,distinct", [("422", 1920, 1080, 1024, 8), ("420", 640, 480, 4400, 6), ("444", 96, 1096, 1601, 5)])
def test_fused_launch_in_passes(dec, layout, ss, W, H, n, distinct, tune):
    """Round 6: a workgroup whose images hold more restart segments than its producers have lanes (8 x 64) takes them in PASSES —
    1024 x 1080p 4:2:2 (135 MCU rows: four images = 540 segments; round 5: two launches, 9.1 ms against 6.3 for 4:2:0), 4400 small
    files (18 per workgroup), tall narrow files (137 rows, ragged: the last workgroup has fewer passes).  The consumers go from one
    pass's jobs to the next's; a producer's progress word keeps rising (pass << 20 | MCUs).  Against the two launches byte for
    byte, every distinct file against the oracle."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    lay = B.MJ_LAYOUT_XMAJOR if layout == "xmajor" else B.MJ_LAYOUT_ROWMAJOR
    raws, files, prep = _fused_batch(ss, W, H, n, distinct, 55000 + W, layout=lay)
    two, st2, form2 = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0"), ("MJ_HUFFMAN", "lanes")])
    assert not st2.any() and not form2 & B.MJ_FORM_FUSED
    per = W * H * 3
    imgs = two.view(n, per)
    for d, want in enumerate(oracle_rgb_all(raws)):
        i = next(k for k in range(n) if (5 * k + k // distinct) % distinct == d)
        got = imgs[i].cpu().numpy()
        got = got.reshape(W, H, 3) if layout == "xmajor" else np.swapaxes(got.reshape(H, W, 3), 0, 1)
        assert np.array_equal(got, want), d
    mh = 16 if ss in ("420", "440") else 8
    shape = B.fused_shape_rule(n, -(-H // mh), hmax=2 if ss in ("420", "422") else 1, vmax=2 if ss in ("420", "440") else 1, transposed=layout == "rowmajor")
    assert shape["ok"] and shape["passes"] >= 2, shape
    for cons in (None, "2"):
        opts = [("MJ_HUFFMAN", "lanes")] + ([("MJ_FUSED_CONSUMERS", cons)] if cons else [])
        fused, st, form = _decode_plan(dec.ctx, prep, n, torch, opts)
        assert form & B.MJ_FORM_FUSED, (form, cons)
        assert not st.any() and torch.equal(fused, two), cons


def test_placement_tuning_keeps_the_pixels_and_one_store(dec):
    """mj_plan_tune_placement: a fused plan tries a few coefficient stores (fresh allocations) and keeps the fastest — the output
    is what it was, a captured graph does not survive with the old store's address in it, the stores that lost are gone (the
    device's free memory is not down by more than one store), and a plan that is not fused is left alone."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    W, H, n, distinct = 640, 480, 800, 5
    raws, files, prep = _fused_batch("420", W, H, n, distinct, 41000)
    dev = torch.device("cuda", 0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    B.set_option("MJ_HUFFMAN", "lanes")
    try:
        plan = B.Plan(dec.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": n})
    finally:
        B.set_option("MJ_HUFFMAN", None)
    try:
        assert plan.stage1_form() & B.MJ_FORM_FUSED
        out = torch.zeros(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):                                   # (the third execute replays a captured graph)
            plan.execute(stream, out.data_ptr())
        plan.sync()
        want = out.clone()
        store0 = plan.device_buffers()["coef"]
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        ms, kept = plan.tune_placement(stream, out.data_ptr(), 4)
        assert len(ms) == 4 and all(m > 0 for m in ms) and 0 <= kept < 4
        assert (plan.device_buffers()["coef"] == store0) == (kept == 0)
        assert plan.best_ms > 0 and plan.best_ms <= min(ms) * 1.05
        # (the candidates that lost went back to the device — which hands freed memory on lazily: what is pinned is that the call
        # does not keep them, i.e. a second call does not take free memory further down)
        free1 = torch.cuda.mem_get_info()[0]
        plan.tune_placement(stream, out.data_ptr(), 4)
        torch.cuda.synchronize()
        assert free1 - torch.cuda.mem_get_info()[0] <= plan.info.total_blocks * 128 + (64 << 20), (free0, free1, torch.cuda.mem_get_info()[0])
        for poison in (0x3C, 0xA5, 0x69):                    # plain, captured, replayed — each on a poisoned store
            out.zero_()
            torch.cuda.synchronize()
            plan.fill_coef(poison)
            plan.execute(stream, out.data_ptr())
            plan.sync()
            assert torch.equal(out, want) and not plan.read(rgb=False)["status"].any()
    finally:
        plan.close()
    raw, _ = load_golden("64x64_420_pil")
    prep1 = __import__("pyjpegdecoder_amd.batch", fromlist=["prepare_batch"]).prepare_batch([raw] * 4, B.MJ_LAYOUT_XMAJOR, 0)
    small = B.Plan(dec.ctx, prep1.to_c(), {"prep": prep1, "n_images": 4})
    try:
        ms, kept = small.tune_placement(0, 0, 3)
        assert ms == [0.0, 0.0, 0.0] and kept == 0
    finally:
        small.close()


def test_fused_launch_damaged_files_do_not_stall_their_workgroup(dec, tune):
    """Files of a fused batch with entropy-coded bytes overwritten (markers left where they are): the statuses are those of the
    two launches, damaged files that still decode give the same pixels, every other image of the batch — the damaged files'
    workgroup neighbours included — is untouched, and the launch returns (what the consumer wavefronts wait for is the
    producers' progress, which does not depend on what the bytes say)."""
    torch = pytest.importorskip("torch")
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    W, H, n, distinct = 640, 480, 800, 6
    raws, files, _ = _fused_batch("420", W, H, n, distinct, 61000)
    rng = np.random.default_rng(3)
    files = list(files)
    hurt = [3, 4, 257, 511, 799]
    for i in hurt:
        b = bytearray(files[i])
        lo, done = len(b) // 3, 0
        while done < 6:
            pos = int(rng.integers(lo, len(b) - 4))
            v = int(rng.integers(0, 255))
            if 0xFF in (b[pos - 1], b[pos], b[pos + 1]) or v == b[pos]:
                continue                                           # (no marker made, none unmade: the host's segmentation stands)
            b[pos] = v
            done += 1
        files[i] = bytes(b)
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    two, st2, _ = _decode_plan(dec.ctx, prep, n, torch, [("MJ_FUSED", "0"), ("MJ_HUFFMAN", "lanes")])
    fused, st, form = _decode_plan(dec.ctx, prep, n, torch, [("MJ_HUFFMAN", "lanes")])
    assert form & B.MJ_FORM_FUSED
    assert np.array_equal(st != 0, st2 != 0), (np.flatnonzero(st), np.flatnonzero(st2))
    assert set(np.flatnonzero(st)) <= set(hurt) and B.MJ_ST_INTERNAL not in st
    per = W * H * 3
    a, b2 = fused.view(n, per), two.view(n, per)
    good = torch.tensor([i for i in range(n) if st[i] == 0], device=a.device)
    assert torch.equal(a[good], b2[good])


# ---- sampling layouts outside the common ones (any factors 1..4 per component) ----------------------------------------
def _odd_layout_names():
    g = np.load(GOLDEN / "odd_layouts.npz")
    return sorted({k.rsplit(".", 1)[0] for k in g.files})


@pytest.mark.parametrize("name", _odd_layout_names())
def test_unusual_sampling_layouts_against_the_reference(dec, dec_rm, name):
    """4:1:0, 1x4, factors of 3, chroma factors above one, luma below the chroma resolution, fourteen blocks per MCU — files the
    reference decoded (tools/make_layout_goldens.py): coefficients, planes and image bit for bit, both orientations."""
    g = np.load(GOLDEN / "odd_layouts.npz")
    raw = g[name + ".jpg"].tobytes()
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    assert np.array_equal(seam["coef"], g[name + ".coef"])
    assert np.array_equal(seam["planes"], g[name + ".planes"])
    assert np.array_equal(img, g[name + ".rgb"])
    (img_rm,) = dec_rm.decode([raw])
    assert np.array_equal(img_rm, g[name + ".rgb"].transpose(1, 0, 2))
    from oracle import oracle
    assert np.array_equal(seam["idct"], oracle.decode(raw, want_idct=True)["idct"])     # the :872 seam, against the (pinned) oracle


def test_unusual_layouts_through_the_class_surface_and_planar_output(tmp_path):
    """The drop-in class on a 4:1:0 baseline file and on a crafted progressive one (subset DC scans); planar output of the generic
    stage 2."""
    from pyjpegdecoder_amd import BatchDecoder, JpegDecoder
    g = np.load(GOLDEN / "odd_layouts.npz")
    f = tmp_path / "y4x2.jpg"
    f.write_bytes(g["y4x2.jpg"].tobytes())
    d = JpegDecoder(f)
    assert np.array_equal(d.image_array, g["y4x2.rgb"]) and d.scan_mode == "baseline_dct"
    assert tuple(d.sample_shape) == (32, 16) and (d.mcu_width, d.mcu_height) == (32, 16)
    gp = np.load(GOLDEN / "crafted_progressive.npz")
    f2 = tmp_path / "subset.jpg"
    f2.write_bytes(gp["dc_y_cr_refined.jpg"].tobytes())
    d2 = JpegDecoder(f2)
    assert np.array_equal(d2.image_array, gp["dc_y_cr_refined.rgb"]) and d2.scan_mode == "progressive_dct" and d2.scan_amount == 10
    dp = BatchDecoder(device=0, layout="planar_rowmajor")
    try:
        (pl,) = dp.decode([g["y2x2_c1x2_2x1.jpg"].tobytes()])
    finally:
        dp.close()
    assert np.array_equal(pl, g["y2x2_c1x2_2x1.rgb"].transpose(2, 1, 0))


def test_unusual_sampling_layouts_random_files_against_the_oracle(dec):
    """Every combination class again on random crafted files (tools/craft_jpeg.py), batched by layout, with and without
    restart markers and with the GPU marker scan, against the oracle (itself pinned on the twelve reference-decoded files)."""
    from oracle import oracle
    from pyjpegdecoder_amd import BatchDecoder
    from tools.craft_jpeg import random_baseline
    rng = np.random.default_rng(7)
    files = [random_baseline(rng, 1000 + i) for i in range(40)]
    want = [oracle.decode(f)["rgb"] for f in files]
    got = dec.decode(files)
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), i
    d2 = BatchDecoder(device=0, segment="gpu")
    try:
        got2 = d2.decode(files)
    finally:
        d2.close()
    for i, (a, b) in enumerate(zip(got2, want)):
        assert np.array_equal(a, b), i


# ---- progressive files written scan by scan: scripts and layouts no encoder at hand produces ------------------------------
def _crafted_progressive_names():
    g = np.load(GOLDEN / "crafted_progressive.npz")
    return sorted({k.rsplit(".", 1)[0] for k in g.files})


@pytest.mark.parametrize("fast", ["1", "0"])
@pytest.mark.parametrize("name", _crafted_progressive_names())
def test_crafted_progressive_scripts_against_the_reference(dec, name, fast, monkeypatch, tune):
    """Interleaved DC scans over a subset of the components (also refined through the same subsets), several bands and
    refinement levels, end-of-band runs over hundreds of blocks, restart intervals, 4:1:0 / 2x4 / 3x1 luma and luma below the
    chroma resolution — decoded by the reference (tools/make_layout_goldens.py).  Both families of walks (MJ_PROG_FAST)."""
    g = np.load(GOLDEN / "crafted_progressive.npz")
    tune("MJ_PROG_FAST", fast)
    (img,), (seam,) = dec.decode([g[name + ".jpg"].tobytes()], return_seams=True)
    assert np.array_equal(seam["coef"], g[name + ".coef"])
    assert np.array_equal(seam["planes"], g[name + ".planes"])
    assert np.array_equal(img, g[name + ".rgb"])


def test_crafted_progressive_random_scripts_against_the_oracle(dec):
    """Random scan scripts (bands cut at random, DC components grouped at random, successive approximation two or three levels
    deep, with and without restart intervals), random layouts the reference can finish (every component 1x1 or at the full
    resolution), batched: against the oracle (tools/craft_jpeg.random_progressive; the reference agrees with the oracle on
    every one of these files — checked where the reference lives, tools/crosscheck_reference.py --crafted)."""
    from oracle import oracle
    from tools.craft_jpeg import random_progressive
    rng = np.random.default_rng(11)
    files = [random_progressive(rng, 3000 + i) for i in range(36)]
    want = [oracle.decode(f) for f in files]
    got, seams = dec.decode(files, return_seams=True)
    for i in range(len(files)):
        assert np.array_equal(seams[i]["coef"], want[i]["coef"]), i
        assert np.array_equal(got[i], want[i]["rgb"]), i


def test_library_first_then_torch_in_a_fresh_process():
    """PyTorch's ROCm wheels bundle a HIP runtime under the soname of /opt/rocm's; whichever loads first serves the process, and
    torch finds "No HIP GPUs" behind /opt/rocm's.  The binding therefore imports torch before it loads libmijpeg.so — a program
    that creates a decoder first and touches torch.cuda afterwards must work (a fresh process: this one has torch loaded long since)."""
    import subprocess
    import sys
    code = ("from pyjpegdecoder_amd import BatchDecoder\n"
            "d = BatchDecoder(0)\n"
            "import torch\n"
            "t = torch.arange(4, device='cuda') + 1\n"
            "assert t.sum().item() == 10\n"
            "d.close()\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
