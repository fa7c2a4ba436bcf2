"""Host logic (container parser, scan segmentation) against the attribute surface captured from the
reference.  CPU only."""
import numpy as np
import pytest

from conftest import golden_index, golden_names, load_golden
from pyjpegdecoder_amd import _parse
from pyjpegdecoder_amd.errors import CorruptedJpeg, NotJpeg, UnsupportedJpeg


@pytest.mark.parametrize("name", golden_names(small_only=False))
def test_attribute_surface_matches_reference(name):
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    p = _parse.parse_jpeg(raw)
    scan = p.scans[0]
    assert p.scan_mode == meta["scan_mode"]
    assert (p.image_width, p.image_height) == (meta["image_width"], meta["image_height"])
    assert {str(k): list(v) for k, v in p.color_components.items()} == \
        {k: [tuple(x) if isinstance(x, list) else x for x in v] for k, v in meta["color_components"].items()} or \
        {str(k): [list(x) if isinstance(x, tuple) else x for x in v] for k, v in p.color_components.items()} == meta["color_components"]
    assert list(p.sample_shape) == meta["sample_shape"]
    assert {str(k): dict(v.tree) for k, v in p.huffman.items()} == meta["huffman_tables"]
    assert {str(k): v.tolist() for k, v in p.quantization_tables.items()} == meta["quantization_tables"]
    assert p.restart_interval == meta["restart_interval"]
    assert p.scan_amount == meta["scan_amount"]
    assert (scan.mcu_width, scan.mcu_height) == (meta["mcu_width"], meta["mcu_height"])
    assert (scan.mcu_count_h, scan.mcu_count_v, scan.mcu_count) == (meta["mcu_count_h"], meta["mcu_count_v"], meta["mcu_count"])
    assert (p.array_width, p.array_height, p.array_depth) == (meta["array_width"], meta["array_height"], meta["array_depth"])
    assert p.file_header == meta["file_header"]
    assert p.reached_eoi == meta["scan_finished"]


@pytest.mark.parametrize("name", golden_names(small_only=False))
def test_restart_segmentation(name):
    raw, _ = load_golden(name)
    p = _parse.parse_jpeg(raw)
    scan = p.scans[0]
    so = scan.segment_offsets
    assert so[0] == scan.entropy_start and so[-1] == scan.entropy_end
    if scan.restart_interval:
        assert so.size - 1 == -(-scan.mcu_count // scan.restart_interval)
        for k, off in enumerate(so[1:-1]):
            assert raw[off - 2] == 0xFF and raw[off - 1] == 0xD0 + (k % 8)
    # what ends a scan: EOI after the last one, the next SOS in non-interleaved files
    assert raw[scan.entropy_end:scan.entropy_end + 2] == (b"\xFF\xD9" if len(p.scans) == 1 else b"\xFF\xDA")
    assert raw[p.scans[-1].entropy_end:p.scans[-1].entropy_end + 2] == b"\xFF\xD9"


def test_not_jpeg():
    with pytest.raises(NotJpeg):
        _parse.parse_jpeg(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)


def test_unsupported_and_corrupt_headers():
    raw, _ = load_golden("c1_64x64_444_pil")
    i = raw.index(b"\xFF\xC0")
    bad = bytearray(raw); bad[i + 4] = 12                      # precision 12
    with pytest.raises(UnsupportedJpeg):
        _parse.parse_jpeg(bytes(bad))
    bad = bytearray(raw); bad[i + 9] = 4                       # 4 components
    with pytest.raises(UnsupportedJpeg):
        _parse.parse_jpeg(bytes(bad))
    bad = bytearray(raw); bad[i + 7] = 0; bad[i + 8] = 0       # width 0
    with pytest.raises(CorruptedJpeg):
        _parse.parse_jpeg(bytes(bad))


def test_undo_zigzag_is_the_reference_layout():
    zz = np.arange(64)
    xy = _parse.undo_zigzag(zz)
    # [x, y]: x = horizontal; zig-zag 1 is the first horizontal frequency, 2 the first vertical one
    assert xy[1, 0] == 1 and xy[0, 1] == 2 and xy[7, 7] == 63 and xy[2, 0] == 5


@pytest.mark.parametrize("name", sorted(n for n in golden_index() if n.startswith("prog_")))
def test_progressive_attribute_surface(name):
    """Progressive files: same attribute surface; the MCU geometry left on the object is the LAST scan's (:591-621)."""
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    p = _parse.parse_jpeg(raw)
    last = p.scans[-1]
    assert p.scan_mode == meta["scan_mode"] == "progressive_dct"
    assert len(p.scans) == meta["scan_count"] == meta["scan_amount"] == p.scan_amount
    assert (last.mcu_width, last.mcu_height) == (meta["mcu_width"], meta["mcu_height"])
    assert (last.mcu_count_h, last.mcu_count_v, last.mcu_count) == (meta["mcu_count_h"], meta["mcu_count_v"], meta["mcu_count"])
    assert (p.array_width, p.array_height, p.array_depth) == (meta["array_width"], meta["array_height"], meta["array_depth"])
    assert {str(k): dict(v.tree) for k, v in p.huffman.items()} == meta["huffman_tables"]       # tables of the last DHTs
    assert p.restart_interval == meta["restart_interval"] and p.file_header == meta["file_header"]
    for sc in p.scans:                                                                          # every scan ends on a marker
        assert raw[sc.entropy_end] == 0xFF and raw[sc.entropy_end + 1] not in (0x00,) + tuple(range(0xD0, 0xD8))


def test_headers_only_parse_matches_full_parse():
    """segment="gpu": the host stops at the SOS of a single-scan baseline file; everything before it is parsed as
    usual and the scan carries the end of the file as its bound."""
    from pyjpegdecoder_amd import parse_jpeg
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    for name in golden_names():
        raw, _ = load_golden(name)
        full, head = parse_jpeg(raw), parse_jpeg(raw, headers_only=True)
        if full.scan_mode != "baseline_dct" or len(full.scans) != 1:
            assert not head.headers_only
            continue
        assert head.headers_only and not full.headers_only
        assert (head.image_width, head.image_height, head.restart_interval) == (full.image_width, full.image_height, full.restart_interval)
        a, b = head.scans[0], full.scans[0]
        assert a.entropy_start == b.entropy_start and a.entropy_end == len(raw) and a.segment_offsets is None
        assert (a.mcu_count_h, a.mcu_count_v, a.component_ids) == (b.mcu_count_h, b.mcu_count_v, b.component_ids)
        prep = prepare_batch([raw], parsed=[head])
        assert prep.flags & B.MJ_FLAG_GPU_SEGMENT and prep.seg_begin.size == 1
        assert prep.seg_begin[0] == a.entropy_start and prep.seg_end[0] == len(raw)
    for name in (n for n in golden_index() if n.startswith("prog_")):
        assert not parse_jpeg(load_golden(name)[0], headers_only=True).headers_only


@pytest.mark.parametrize("name", ["c1_64x64_444_pil", "128x64_420_dri3", "50x70_grey_dri4", "prog_70x50_420_pil"])
def test_class_handlers_are_the_references_six_and_callable(name):
    """jpeg_decoder.py:112-652: `handlers` maps markers to bound methods that take the segment payload and update the
    object's attributes.  Drive them by hand over a golden file, up to (not including) EOI — no GPU involved — and
    compare the attributes with what the reference left on its object."""
    import numpy as np
    from pyjpegdecoder_amd import JpegDecoder
    from pyjpegdecoder_amd._parse import ParsedJpeg, RST, bytes_to_uint
    raw, _ = load_golden(name)
    meta = golden_index()[name]
    d = JpegDecoder.__new__(JpegDecoder)
    d._verbose = False
    d.raw_file, d.file_size, d.file_header = raw, len(raw), 2
    d._parsed, d._arr = ParsedJpeg(raw=raw, file_size=len(raw)), np.frombuffer(raw, dtype=np.uint8)
    d.scan_finished, d.scan_mode, d.huffman_tables, d.quantization_tables = False, None, {}, {}
    d.color_components, d.restart_interval, d.image_array, d.scan_count = {}, 0, None, 0
    d.handlers = {b"\xFF\xC4": d.define_huffman_table, b"\xFF\xDB": d.define_quantization_table,
                  b"\xFF\xDD": d.define_restart_interval, b"\xFF\xC0": d.start_of_frame, b"\xFF\xC2": d.start_of_frame,
                  b"\xFF\xDA": d.start_of_scan}
    seen = []
    while d.file_header < len(raw):                      # the reference's loop (:78-110), minus EOI
        if raw[d.file_header] != 0xFF:
            d.file_header += 1
            continue
        m = raw[d.file_header:d.file_header + 2]
        d.file_header += 2
        if m == b"\xFF\x00" or m in RST:
            continue
        if m == b"\xFF\xD9":
            break
        size = bytes_to_uint(raw[d.file_header:d.file_header + 2]) - 2
        d.file_header += 2
        h = d.handlers.get(m)
        if h is None:
            d.file_header += size
        else:
            seen.append(m)
            h(raw[d.file_header:d.file_header + size])
    assert b"\xFF\xDA" in seen and b"\xFF\xDB" in seen and b"\xFF\xC4" in seen
    assert d.scan_mode == meta["scan_mode"] and (d.image_width, d.image_height) == (meta["image_width"], meta["image_height"])
    assert {str(k): v for k, v in d.huffman_tables.items()} == meta["huffman_tables"]
    assert {str(k): v.tolist() for k, v in d.quantization_tables.items()} == meta["quantization_tables"]
    assert d.restart_interval == meta["restart_interval"] and d.scan_amount == meta["scan_amount"]
    for k in ("mcu_width", "mcu_height", "mcu_count_h", "mcu_count_v", "mcu_count", "array_width", "array_height", "array_depth"):
        assert getattr(d, k) == meta[k], k
    assert len(d._parsed.scans) == meta["scan_count"]
