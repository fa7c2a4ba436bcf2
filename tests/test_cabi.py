"""The C-ABI library loads and exports every symbol include/mijpeg.h declares; host-side pieces of it
are pinned against the reference's goldens.  No compute calls; CPU only."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from pyjpegdecoder_amd import _binding as B
    if not B.LIB_PATH.exists():
        g.build()
    return B.load_library()


def test_exports_every_declared_symbol(lib):
    from pyjpegdecoder_amd import _binding as B
    header = (ROOT / "include" / "mijpeg.h").read_text()
    declared = set(re.findall(r"\b(mj_[a-z0-9_]+)\s*\(", header))
    assert declared == set(B.EXPORTS), declared ^ set(B.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mj_version() == 1


def test_library_idct_table_is_the_reference_table(lib):
    tt = np.empty(4096, dtype=np.float64)
    lib.mj_host_idct_table(tt.ctypes.data_as(ctypes.c_void_p))
    ref = np.load(GOLDEN / "idct_table.npy")            # [x,y,u,v]
    want = ref.transpose(2, 3, 0, 1).reshape(4096)      # [u*8+v][x*8+y]
    assert np.array_equal(tt.view(np.uint64), want.view(np.uint64))


def test_upsample_taps_header_matches_captured_operator():
    txt = (ROOT / "pyjpegdecoder_amd" / "csrc" / "upsample_taps.h").read_text()
    for dst, name in (((16, 16), "UP_TAPS_16x16"), ((16, 8), "UP_TAPS_16x8"), ((8, 16), "UP_TAPS_8x16")):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S).group(1)
        words = [int(w.rstrip("u"), 16) for w in re.findall(r"0x[0-9a-f]+u", body)]
        W = np.load(GOLDEN / f"upsample_W_8x8_{dst[0]}x{dst[1]}.npy")
        assert len(words) == W.shape[0]
        for o, word in enumerate(words):
            row = np.zeros(64, dtype=int)
            for t in range(3):
                idx, w = (word >> (10 * t)) & 63, (word >> (10 * t + 6)) & 15
                row[idx] += w
            assert np.array_equal(row, W[o]), (name, o)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyjpegdecoder_amd import BackendError, BatchDecoder
    with pytest.raises(BackendError):
        BatchDecoder(device=0)


def test_product_never_touches_the_oracle():
    pkg = ROOT / "pyjpegdecoder_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        src = f.read_text()
        assert "oracle" not in src.lower() or f.name in (), f"{f} mentions the oracle"
    assert "oracle" not in (ROOT / "include" / "mijpeg.h").read_text().lower()


def test_header_is_plain_c_and_layouts_match_the_binding(tmp_path):
    """include/mijpeg.h compiles as C (the boundary is a C ABI, not C++), and every struct has the size and field
    offsets the ctypes binding assumes."""
    import ctypes
    import shutil
    import subprocess
    from pyjpegdecoder_amd import _binding as B
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = [("mj_huff_spec", B.HuffSpecC), ("mj_image_desc", B.ImageDescC), ("mj_scan_desc", B.ScanDescC),
             ("mj_batch", B.BatchC), ("mj_plan_info", B.PlanInfoC), ("mj_host_job", B.HostJobC)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mijpeg.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (cname, cls), line in zip(pairs, out):
        got = line.split()
        assert got[0] == cname
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in got[1:]] == want, cname


def test_option_values_are_validated_and_readable(lib):
    """mj_set_option refuses a value outside the documented range or word list (MJ_ERR_INVALID) and keeps what the option
    had — a probe sweep must not time the defaults under another label; mj_get_option reads it back; both refuse a name
    that is no option.  (No GPU: the option table is host state.)"""
    from pyjpegdecoder_amd import _binding as B
    try:
        B.set_option("MJ_LANES_PER_WAVE", 34)
        assert B.get_option("MJ_LANES_PER_WAVE") == "34"
        for bad in ("0", "65", "-3", "12x", "lots"):
            with pytest.raises(ValueError):
                B.set_option("MJ_LANES_PER_WAVE", bad)
            assert B.get_option("MJ_LANES_PER_WAVE") == "34"
        for name, good, bad in (("MJ_HUFFMAN", "lanes11", "lanes12"), ("MJ_SEG_ORDER", "striped", "sorted"), ("MJ_LANES_RING", "64", "96"),
                                ("MJ_SYNC_CHUNK", "1024", "1022"), ("MJ_SYNC_ROUNDS", "0", "65"), ("MJ_PROG_PARTS", "8", "9"),
                                ("MJ_PROG_SPLIT", "3", "4"), ("MJ_LANES_WAVES", "16", "17")):
            B.set_option(name, good)
            assert B.get_option(name) == good
            with pytest.raises(ValueError):
                B.set_option(name, bad)
            assert B.get_option(name) == good
            B.set_option(name, None)
            assert B.get_option(name) == ""
        with pytest.raises(B.UnknownOption):
            B.set_option("MJ_NO_SUCH_SWITCH", "1")
        with pytest.raises(B.UnknownOption):
            B.get_option("PATH")
        assert lib.mj_set_option(None, b"1") == B.MJ_ERR_INVALID
    finally:
        B.set_option("MJ_LANES_PER_WAVE", None)


def test_options_set_from_another_thread_while_being_read(lib):
    """The value handed to a reader is a copy made under the lock: a thread that keeps re-setting an option cannot pull the
    string out from under a plan-creating thread (here: 20 000 reads against 20 000 writes of values of different lengths)."""
    import threading
    from pyjpegdecoder_amd import _binding as B
    stop = threading.Event()

    def writer():
        vals = ["1", "16", "7", "12"]
        i = 0
        while not stop.is_set():
            B.set_option("MJ_LANES_WAVES", vals[i & 3])
            i += 1
    t = threading.Thread(target=writer)
    t.start()
    try:
        for _ in range(20000):
            assert B.get_option("MJ_LANES_WAVES") in ("", "1", "16", "7", "12")
    finally:
        stop.set()
        t.join()
        B.set_option("MJ_LANES_WAVES", None)
