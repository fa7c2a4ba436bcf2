"""Pins the CPU oracle (oracle/jpeg_oracle.c) against vectors captured from the reference itself
(tools/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from conftest import GOLDEN, golden_index, golden_names, load_golden
from oracle import oracle


def test_idct_table_matches_reference_table():
    ref = np.load(GOLDEN / "idct_table.npy")
    assert ref.shape == (8, 8, 8, 8) and ref.dtype == np.float64
    # bit-for-bit, not allclose: rounding at exact ties depends on the last ulp
    assert np.array_equal(oracle.idct_table().view(np.uint64), ref.view(np.uint64))


def test_idct_known_answers_including_ties():
    g = np.load(GOLDEN / "idct_blocks.npz")
    out = oracle.idct_xy(g["blocks"])
    assert np.array_equal(out, g["out"])
    # F6: DC*q = 4 rounds away from zero (0.5 -> 1), -4 -> -1
    assert g["out"][0, 0, 0] == 129 and g["out"][1, 0, 0] == 127


@pytest.mark.parametrize("dst", [(16, 16), (16, 8), (8, 16)])
def test_upsample_known_answers(dst):
    g = np.load(GOLDEN / "resize_blocks.npz")
    ins, outs = g[f"in_{dst[0]}x{dst[1]}"], g[f"out_{dst[0]}x{dst[1]}"]
    for i in range(ins.shape[0]):
        assert np.array_equal(oracle.upsample(ins[i], dst), outs[i])


def test_upsample_operator_shape():
    for dst in [(16, 16), (16, 8), (8, 16)]:
        W = oracle.load_W((8, 8), dst)
        assert W.shape == (dst[0] * dst[1], 64)
        assert (W.sum(1) == 15).all() and ((W != 0).sum(1) <= 3).all() and W.min() >= 0


def test_operator_from_diagonals_equals_the_matrices_captured_in_round_1():
    """Every other pair of shapes gets its operator from the captured cell diagonals (tools/make_layout_goldens.py checks
    the formula against griddata for all 84 pairs of factors 1..4 when it writes them); here: the formula reproduces the four
    dense matrices captured earlier, up to the common denominator."""
    diag = np.load(GOLDEN / "upsample_diagonals.npz")
    assert sorted(diag.files) == sorted(f"{8 * h}x{8 * v}" for h in range(1, 5) for v in range(1, 5))
    for dst in [(16, 16), (16, 8), (8, 16), (32, 8)]:
        W1 = np.load(GOLDEN / f"upsample_W_8x8_{dst[0]}x{dst[1]}.npy").astype(np.int64)
        Wf = oracle.operator_from_diagonals((8, 8), dst, diag["8x8"]).astype(np.int64)
        assert np.array_equal(Wf * W1.sum(1)[:, None], W1 * Wf.sum(1)[:, None])
    W = oracle.load_W((16, 8), (32, 24))          # a pair only the formula serves: numerators over 31 * 23
    assert W.shape == (32 * 24, 128) and (W.sum(1) == 31 * 23).all() and ((W != 0).sum(1) <= 3).all() and W.min() >= 0


def odd_layout_names():
    g = np.load(GOLDEN / "odd_layouts.npz")
    return sorted({k.rsplit(".", 1)[0] for k in g.files})


@pytest.mark.parametrize("name", odd_layout_names())
def test_unusual_sampling_layouts_every_seam(name):
    """Files with sampling factors outside the common layouts (4:1:0, 1x4, factors of 3, chroma factors above one, luma below
    the chroma resolution, fourteen blocks per MCU), decoded by the reference (tools/make_layout_goldens.py): coefficients,
    the planes before the colour conversion and the image."""
    g = np.load(GOLDEN / "odd_layouts.npz")
    o = oracle.decode(g[name + ".jpg"].tobytes())
    assert np.array_equal(o["coef"], g[name + ".coef"])
    assert np.array_equal(o["planes"], g[name + ".planes"])
    assert np.array_equal(o["rgb"], g[name + ".rgb"])


def crafted_progressive_names():
    g = np.load(GOLDEN / "crafted_progressive.npz")
    return sorted({k.rsplit(".", 1)[0] for k in g.files})


@pytest.mark.parametrize("name", crafted_progressive_names())
def test_crafted_progressive_scripts_every_seam(name):
    """Progressive files written scan by scan (tools/craft_jpeg.py) with scripts no encoder at hand produces — interleaved DC
    scans over a subset of the components, several bands and refinement levels, end-of-band runs over hundreds of blocks, 4:1:0
    and other unusual layouts — as the reference decoded them (tools/make_layout_goldens.py): coefficient store after the last
    scan, planes, image."""
    g = np.load(GOLDEN / "crafted_progressive.npz")
    o = oracle.decode(g[name + ".jpg"].tobytes())
    assert np.array_equal(o["coef"], g[name + ".coef"])
    assert np.array_equal(o["planes"], g[name + ".planes"])
    assert np.array_equal(o["rgb"], g[name + ".rgb"])


def test_ycbcr_to_rgb_known_answers_including_ties():
    g = np.load(GOLDEN / "ycc_rgb.npz")
    assert np.array_equal(oracle.ycbcr_to_rgb(g["ycc"]), g["rgb"])


@pytest.mark.parametrize("name", golden_names())
def test_file_fixture_every_seam(name):
    raw, vec = load_golden(name)
    meta = golden_index()[name]
    out = oracle.decode(raw, want_idct=True)
    parsed = out["parsed"]
    assert np.array_equal(out["coef"], vec["coef"]), "G1 zig-zag coefficients"
    # G2: dequantised blocks
    deq = []
    bpm_comp = []
    scan = parsed.scans[-1]                      # non-interleaved files (ni_*): one scan per component
    comps = [parsed.color_components[c] for c in (scan.component_ids if len(parsed.scans) == 1 else parsed.color_components)]
    for c in comps:
        bpm_comp += [c.quantization_table_id] * (c.repeat if len(comps) > 1 else 1)
    for i in range(out["coef"].shape[0]):
        q = parsed.quantization_tables[bpm_comp[i % len(bpm_comp)]]
        d, _ = oracle.dequant_idct(out["coef"][i], q)
        deq.append(d[0])
    assert np.array_equal(np.stack(deq), vec["deq"]), "G2 dequantised blocks"
    assert np.array_equal(out["idct"], vec["idct"]), "G3 IDCT output"
    assert np.array_equal(out["planes"], vec["planes"]), "G5 YCbCr planes"
    assert np.array_equal(out["rgb"], vec["rgb"]), "G6 image_array"
    assert out["rgb"].dtype == np.uint8 and list(out["rgb"].shape) == meta["image_array_shape"]
    # file_header after the scan: the reference ends on EOI handling (+2 marker, +2 bogus length read)
    assert out["end_pos"] == scan.entropy_end


def prog_names():
    return [n for n in golden_index() if n.startswith("prog_")]


@pytest.mark.parametrize("name", prog_names())
def test_progressive_fixture_scan_by_scan(name):
    """progressive_dct_scan (:908-1304) restated: the coefficient store after every scan, then the final pass."""
    from pyjpegdecoder_amd import parse_jpeg
    raw, vec = load_golden(name)
    parsed = parse_jpeg(raw)
    assert parsed.scan_mode == "progressive_dct" and len(parsed.scans) == vec["coef_before_scan"].shape[0]
    for k in range(1, len(parsed.scans)):
        coef, st, _ = oracle.progressive_entropy_decode(parsed, upto=k)
        assert st == 0
        assert np.array_equal(coef, vec["coef_before_scan"][k]), f"state before scan {k + 1}"
    out = oracle.decode(raw)
    assert np.array_equal(out["coef"], vec["coef"]), "coefficients before the IDCT pass (incl. F8 refinement semantics)"
    assert np.array_equal(out["planes"], vec["planes"])
    assert np.array_equal(out["rgb"], vec["rgb"])


def _example():
    import json
    d = GOLDEN / "example"
    meta = json.loads((d / "known_answers.json").read_text())
    raw = (d / "base_image.jpg").read_bytes()
    return raw, meta, np.load(d / "samples.npz")


def example_cut(raw, meta, k):
    """The file as it stood after scan k: everything up to that scan's terminating marker, then EOI."""
    return raw[:meta["scans"][k - 1]["entropy_end"]] + b"\xff\xd9"


@pytest.mark.parametrize("k", [1, 2])
def test_reference_repository_known_answers_after_scan(k):
    """The reference repository's own example (progressive, 4160x2340, restart interval redefined between scans):
    its PNGs of the image after scan 1 and after scan 2 are known answers nobody here generated; the oracle's decode
    of the file truncated after that scan must reproduce them exactly."""
    import hashlib
    from oracle import oracle
    raw, meta, samples = _example()
    out = oracle.decode(example_cut(raw, meta, k))["rgb"]
    assert out.shape == (meta["width"], meta["height"], 3)
    sx, sy = meta["sample_strides"]
    assert np.array_equal(out[::sx, ::sy], samples[f"after_scan_{k}"])
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == meta["after_scan"][str(k)]["sha256_rgb_xmajor"]
