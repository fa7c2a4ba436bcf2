"""Edge cases of the host logic and the batch API that need no GPU: empty / ragged / oversized inputs,
unsupported layouts, table de-duplication."""
import numpy as np
import pytest

from conftest import load_golden
from pyjpegdecoder_amd import CorruptedJpeg, UnsupportedJpeg, parse_jpeg
from pyjpegdecoder_amd.batch import check_supported, prepare_batch


def test_ragged_batch_descriptors():
    names = ["c1_64x64_444_pil", "70x50_420_pil_opt", "128x64_420_dri3", "100x36_420_dri7"]
    raws = [load_golden(n)[0] for n in names]
    prep = prepare_batch(raws[1:])           # three 4:2:0 files of different sizes / tables / restart intervals
    assert [tuple(s) for s in prep.shapes] == [(70, 50, 3), (128, 64, 3), (100, 36, 3)]
    segs = [prep.descs[i].n_segments for i in range(3)]
    assert segs == [1, 11, 3]                 # ceil(mcus / restart_interval)
    assert prep.seg_begin.size == sum(segs) and (prep.seg_end >= prep.seg_begin).all()
    assert (prep.file_offsets % 4 == 0).all()
    # optimised tables of the Pillow file and the Annex-K tables of the other two are distinct entries; the two
    # Annex-K users share theirs
    assert prep.n_huff == 8
    assert prep.descs[1].dc_sel[0] == prep.descs[2].dc_sel[0] and prep.descs[0].dc_sel[0] != prep.descs[1].dc_sel[0]
    for i in range(3):                        # every segment lies inside its file
        d = prep.descs[i]
        b = prep.seg_begin[d.first_segment:d.first_segment + d.n_segments]
        e = prep.seg_end[d.first_segment:d.first_segment + d.n_segments]
        assert (b >= prep.file_offsets[i]).all() and (e <= prep.file_offsets[i + 1]).all()


def test_progressive_descriptors_and_empty_file():
    raw, _ = load_golden("prog_70x50_420_pil")
    p = parse_jpeg(raw)
    assert p.scan_mode == "progressive_dct" and len(p.scans) == 10
    check_supported(p)
    prep = prepare_batch([raw, load_golden("prog_64x64_420_pil")[0]])
    assert prep.n_scans == 20 and [prep.scans[k].image for k in (0, 9, 10, 19)] == [0, 0, 1, 1]
    s0, s1 = prep.scans[0], prep.scans[1]
    assert (s0.ss, s0.se, s0.n_comp) == (0, 0, 3) and (s0.mcu_count_h, s0.mcu_count_v) == (5, 4)      # interleaved DC scan
    assert s1.n_comp == 1 and s1.ss > 0 and (s1.mcu_count_h, s1.mcu_count_v) == (9, 7)                # luma AC scan: ceil(70/8) x ceil(50/8)
    with pytest.raises(UnsupportedJpeg):                                                              # no mixing of modes in one plan
        prepare_batch([raw, load_golden("64x64_420_pil")[0]])
    # header only, no scan: the reference falls off the end of the file (:81-83); the batch API refuses
    raw, _ = load_golden("c1_64x64_444_pil")
    p = parse_jpeg(raw[:raw.index(b"\xFF\xDA")])
    assert p.scans == [] and not p.reached_eoi
    with pytest.raises(CorruptedJpeg):
        check_supported(p)


def test_restart_marker_count_is_checked():
    raw, _ = load_golden("128x64_420_dri3")
    p = parse_jpeg(raw)
    off = int(p.scans[0].segment_offsets[3])
    with pytest.raises(CorruptedJpeg):
        prepare_batch([raw[:off - 2] + raw[off:]])      # one RSTn removed


def test_missing_tables_are_corrupt():
    raw, _ = load_golden("c1_64x64_444_pil")
    i = raw.index(b"\xFF\xC4")
    n = int.from_bytes(raw[i + 2:i + 4], "big")
    with pytest.raises(CorruptedJpeg):
        prepare_batch([raw[:i] + raw[i + 2 + n:]])      # first DHT segment dropped


def test_max_dimension_geometry():
    # SOF with 65535 x 65535: geometry arithmetic must not overflow (no decode attempted)
    raw, _ = load_golden("64x64_grey_pil")
    i = raw.index(b"\xFF\xC0")
    big = bytearray(raw)
    big[i + 5:i + 9] = b"\xFF\xFF\xFF\xFF"
    p = parse_jpeg(bytes(big))
    assert (p.image_width, p.image_height) == (65535, 65535)
    assert p.scans[0].mcu_count_h == 8192 and p.scans[0].mcu_count == 8192 * 8192


def test_scan_structures_the_reference_cannot_decode_are_refused_on_the_host():
    """What stays refused after round 3, and why: the reference itself fails on these files (tools/craft_jpeg.py writes them;
    /root/reference run on them where it lives: IndexError / ValueError).  Everything else the crafter can write is decoded
    (tests/test_gpu_parity.py: the crafted files)."""
    from tools.craft_jpeg import craft_progressive
    y420 = ((2, 2), (1, 1), (1, 1))
    # a single-component DC scan of a subsampled component: :993-994 step its blocks by the component's MCU size -> IndexError
    raw = craft_progressive(40, 40, y420, seed=8, script=[((0,), 0, 0, 0, 0), ((1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 0)])
    with pytest.raises(UnsupportedJpeg):
        check_supported(parse_jpeg(raw))
    # a component between 1x1 and the full resolution in a progressive file: the final pass (:1345-1358) cannot place its blocks
    raw = craft_progressive(40, 40, ((2, 2), (2, 1), (1, 1)), seed=12)
    with pytest.raises(UnsupportedJpeg):
        check_supported(parse_jpeg(raw))
    # ... while these are taken: DC scans interleaved over a subset of the components, and a lone 1x1 component
    raw = craft_progressive(50, 37, y420, seed=5, script=[((0, 1), 0, 0, 0, 0), ((2,), 0, 0, 0, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0),
                                                          ((2,), 1, 63, 0, 0)])
    p = parse_jpeg(raw)
    check_supported(p)
    prep = prepare_batch([raw])
    assert prep.n_scans == 5 and prep.scans[0].n_comp == 2 and prep.scans[1].n_comp == 1
    assert (prep.scans[0].mcu_count_h, prep.scans[0].mcu_count_v) == (4, 3)       # the frame's MCUs (:591-594, :610-611)
    assert (prep.scans[1].mcu_count_h, prep.scans[1].mcu_count_v) == (4, 3)       # Cr alone: ceil(ceil(50/2)/8) x ceil(ceil(37/2)/8)
    # unusual sampling factors (4:1:0, luma below the chroma resolution) in a progressive file
    for factors in (((4, 2), (1, 1), (1, 1)), ((1, 1), (2, 2), (2, 2))):
        check_supported(parse_jpeg(craft_progressive(40, 40, factors, seed=3)))
