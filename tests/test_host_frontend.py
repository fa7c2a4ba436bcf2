"""The native host front end (mj_host_assemble, csrc/host_frontend.cpp) against the Python host code it stands in
for (_parse.parse_jpeg(headers_only=True) + batch.prepare_batch): same arrays byte for byte when it accepts a batch,
and it declines — never guesses — whatever the Python marker loop has something to say about.  CPU only: the front
end makes no HIP call."""
import ctypes

import numpy as np
import pytest

from conftest import golden_index, load_golden
from pyjpegdecoder_amd import _binding as B
from pyjpegdecoder_amd._parse import parse_jpeg
from pyjpegdecoder_amd.batch import PreparedBatch, prepare_batch, prepare_batch_native
from tools import synth


def _py(files, layout=B.MJ_LAYOUT_XMAJOR):
    return prepare_batch(files, layout, 0, [parse_jpeg(f, headers_only=True) for f in files])


def _same(nat, py):
    assert nat is not None
    assert nat.flags == py.flags and nat.layout == py.layout and nat.shapes == py.shapes
    assert np.array_equal(nat.file_offsets, py.file_offsets)
    assert nat.blob.size == py.blob.size and np.array_equal(nat.blob, py.blob)
    n = len(py.parsed)
    assert bytes(nat.descs) == bytes(py.descs), "mj_image_desc array"
    assert np.array_equal(nat.seg_begin, py.seg_begin) and np.array_equal(nat.seg_end, py.seg_end)
    assert nat.n_huff == py.n_huff
    assert bytes(nat.huff)[:272 * py.n_huff] == bytes(py.huff)[:272 * py.n_huff], "Huffman tables, numbering included"
    assert np.array_equal(nat.qt, py.qt)
    assert len(nat.parsed) == n


def baseline_goldens():
    idx = golden_index()
    return sorted(n for n in idx if not n.startswith(("prog_", "ni_")))


@pytest.mark.parametrize("name", baseline_goldens())
def test_every_baseline_fixture_alone(name):
    raw, _ = load_golden(name)
    _same(prepare_batch_native([raw]), _py([raw]))


def test_batch_with_shared_and_distinct_tables():
    # three qualities (distinct quantisation tables), several restart intervals, odd sizes (alignment gaps in the blob)
    files = [synth.synth_jpeg(s, 72 + 8 * (s % 3), 40 + s, q, "420", ri) for s, (q, ri) in
             enumerate([(85, 1), (85, 5), (60, 7), (95, 3), (60, 2), (85, 4), (95, 9)])]
    for threads in (1, 3, 16):
        _same(prepare_batch_native(files, n_threads=threads), _py(files))
    assert _py(files).qt.shape[0] > 2
    nodri = [synth.synth_jpeg(20 + s, 72 + 8 * (s % 3), 40 + s, 85 - 10 * (s % 3), "420", 0) for s in range(5)]
    _same(prepare_batch_native(nodri), _py(nodri))


def test_rowmajor_and_flags_pass_through():
    files = [synth.synth_jpeg(1, 64, 48, 85, "444", 0)] * 2
    nat = prepare_batch_native(files, B.MJ_LAYOUT_ROWMAJOR, B.MJ_FLAG_KEEP_COEF)
    assert nat.layout == B.MJ_LAYOUT_ROWMAJOR and nat.flags == B.MJ_FLAG_KEEP_COEF | B.MJ_FLAG_GPU_SEGMENT


def test_staging_buffer_is_used_and_gaps_are_zeroed():
    files = [synth.synth_jpeg(s, 40, 24, 85, "420", 0) for s in range(3)]
    staging = np.full(1 << 16, 0xAB, dtype=np.uint8)
    nat = prepare_batch_native(files, staging=staging)
    assert nat.blob.ctypes.data == staging.ctypes.data
    _same(nat, _py(files))                       # every byte between and behind the files is zero, not 0xAB


def _declined(files):
    return prepare_batch_native(files) is None


def test_declines_what_the_python_loop_must_see():
    good = synth.synth_jpeg(0, 48, 32, 85, "420", 0)
    assert not _declined([good, good])
    prog = [n for n in golden_index() if n.startswith("prog_")][0]
    ni = [n for n in golden_index() if n.startswith("ni_")][0]
    assert _declined([good, load_golden(prog)[0]])                     # SOF2
    assert _declined([load_golden(ni)[0]])                             # one scan per component
    assert _declined([b"\x89PNG\r\n\x1a\n" + bytes(64)])               # NotJpeg
    assert _declined([good[:40]])                                      # cut inside the headers
    assert _declined([good[:2] + b"\xff\xd9"])                         # EOI before any scan
    sof = good.find(b"\xff\xc0")
    assert _declined([good[:sof + 4] + b"\x0c" + good[sof + 5:]])      # 12-bit precision
    assert _declined([good[:sof + 5] + b"\x00\x00" + good[sof + 7:]])  # height 0: DNL
    assert _declined([good[:sof + 9] + b"\x04" + good[sof + 10:]])     # 4 components
    dqt = good.find(b"\xff\xdb")
    assert _declined([good[:dqt] + good[dqt + 2 + int.from_bytes(good[dqt + 2:dqt + 4], "big"):]])   # no quantisation tables
    odd = good[:2] + b"\xff\xc8\x00\x02" + good[2:]                    # a marker the front end does not know
    assert _declined([odd])
    assert parse_jpeg(odd, headers_only=True).image_width == 48        # ... which the Python loop skips like the reference


def test_missing_table_is_declined():
    good = synth.synth_jpeg(0, 48, 32, 85, "420", 0)
    dht = good.find(b"\xff\xc4")
    # turn the first DHT segment into a comment: its tables are never defined
    assert _declined([good[:dht] + b"\xff\xfe" + good[dht + 2:]])


def test_which_file_was_declined():
    good = synth.synth_jpeg(0, 48, 32, 85, "420", 0)
    files = [good, good, b"nope" * 10, good, b"nope"]
    lib = B.load_library()
    n = len(files)
    sizes = np.array([len(f) for f in files], dtype=np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum((sizes + 3) & ~3)
    blob = np.empty(int(offs[-1]) + 1024, dtype=np.uint8)
    job = B.HostJobC()
    job.n_files = n
    ptrs = (ctypes.c_char_p * n)(*files)
    job.files = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_char_p))
    job.sizes, job.file_off, job.blob, job.blob_len = sizes.ctypes.data, offs.ctypes.data, blob.ctypes.data, blob.size
    descs = (B.ImageDescC * n)()
    sb, se = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    huff, qt = (B.HuffSpecC * (6 * n))(), np.empty((3 * n, 64), dtype=np.uint16)
    job.images = ctypes.cast(descs, ctypes.POINTER(B.ImageDescC))
    job.seg_begin, job.seg_end = sb.ctypes.data, se.ctypes.data
    job.huff, job.huff_cap, job.qt, job.qt_cap, job.n_threads = ctypes.cast(huff, ctypes.POINTER(B.HuffSpecC)), 6 * n, qt.ctypes.data, 3 * n, 1
    assert lib.mj_host_assemble(ctypes.byref(job)) == B.MJ_HOST_DECLINED and job.declined_file == 2
    job.n_files = 0
    assert lib.mj_host_assemble(ctypes.byref(job)) == B.MJ_ERR_INVALID


def test_mixed_layouts_go_to_the_python_path():
    a = synth.synth_jpeg(0, 48, 32, 85, "420", 0)
    b = synth.synth_jpeg(1, 48, 32, 85, "444", 0)
    # fine files that do not belong in one plan: the groups they fall into (sampling layout; with / without restart markers)
    assert prepare_batch_native([a, b]) == [[0], [1]] or prepare_batch_native([a, b]) == [[1], [0]]
    assert isinstance(prepare_batch_native([a, a]), PreparedBatch)
    c = synth.synth_jpeg(2, 48, 32, 85, "420", 3)
    groups = prepare_batch_native([a, c, b, a, c])
    assert sorted(map(tuple, groups)) == [(0, 3), (1, 4), (2,)]


def test_fuzzed_headers_accept_only_what_python_builds_identically():
    """Random damage in front of the scan: whenever the front end accepts, the Python path must accept too and build
    the same arrays — it may decline as often as it likes."""
    from pyjpegdecoder_amd.errors import JpegError
    rng = np.random.default_rng(7)
    bases = [synth.synth_jpeg(0, 48, 32, 85, "420", 4), synth.synth_jpeg(1, 40, 40, 70, "444", 0),
             load_golden([n for n in baseline_goldens() if "grey" in n or "gray" in n][0])[0]]
    accepted = 0
    for trial in range(600):
        raw = bytearray(bases[trial % len(bases)])
        hdr_end = raw.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            pos = int(rng.integers(2, hdr_end))
            if kind == 0:
                raw[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                del raw[pos:pos + int(rng.integers(1, 6))]
            else:
                raw[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8))
        f = bytes(raw)
        nat = prepare_batch_native([f])
        if not isinstance(nat, PreparedBatch):
            continue
        accepted += 1
        try:
            py = _py([f])
        except (JpegError, IndexError, ValueError, KeyError, ZeroDivisionError) as exc:
            raise AssertionError(f"trial {trial}: accepted natively, but the Python path raises {exc!r}")
        assert py.flags & B.MJ_FLAG_GPU_SEGMENT, f"trial {trial}: the Python path does not treat this as a one-scan baseline file"
        _same(nat, py)
    assert accepted > 50


def test_split_keeps_what_it_can():
    """split=True: files the front end does not take (progressive, one scan per component, not a JPEG) are left to the
    Python path by index, the rest is grouped by kind — a stray progressive file no longer sinks a batch."""
    a = synth.synth_jpeg(0, 48, 32, 85, "420", 0)
    b = synth.synth_jpeg(1, 48, 32, 85, "444", 0)
    c = synth.synth_jpeg(2, 48, 32, 85, "420", 3)
    prog = load_golden([n for n in golden_index() if n.startswith("prog_")][0])[0]
    files = [a, prog, c, b"junk", a, b, c, prog]
    groups, rest = prepare_batch_native(files, split=True)
    assert rest == [1, 3, 7]
    assert sorted(map(tuple, groups)) == [(0, 4), (2, 6), (5,)]
    assert prepare_batch_native(files) is None                                  # without split: all or nothing
    assert prepare_batch_native([prog, b"junk"], split=True) == ([], [0, 1])
    assert prepare_batch_native([a, a, a], split=True) == ([[0, 1, 2]], [])
    # the groups assemble like batches of their own
    for g in groups:
        sub = [files[i] for i in g]
        _same(prepare_batch_native(sub), _py(sub))


def test_oversubscribed_huffman_table_gets_the_same_answer_from_both_front_ends():
    """A DHT whose code lengths claim more codes than exist (Kraft sum > 1): the Python parser raises CorruptedJpeg
    (_parse.parse_huffman_segment: the reference would build colliding keys and fail later, :718-719); the native front
    end must not build a first-fit LUT for such a file on its own — it declines, so the batch takes the Python path and the
    same exception."""
    from pyjpegdecoder_amd import CorruptedJpeg
    raw = bytearray(synth.synth_jpeg(5, 64, 48, 85, "420", 0))
    at = raw.index(b"\xFF\xC4")                     # first DHT: BITS[16] start 5 bytes in
    bits = at + 5
    raw[bits + 1] += 3                              # three more 2-bit codes than the table had: over-subscribed ...
    # (keep the segment's length consistent: take the three extra symbols from the longest lengths)
    taken = 0
    for l in range(15, 1, -1):
        while raw[bits + l] > 0 and taken < 3:
            raw[bits + l] -= 1
            taken += 1
    assert taken == 3
    raw = bytes(raw)
    with pytest.raises(CorruptedJpeg):
        parse_jpeg(raw, headers_only=True)
    assert prepare_batch_native([raw]) is None
