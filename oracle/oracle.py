"""ctypes front-end of oracle/jpeg_oracle.c — the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product package never imports this module (tests/test_cabi.py
greps for it).  Parity status: pinned against vectors captured from the reference itself
(tools/make_goldens.py -> tests/golden/), checked by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path
from typing import Dict, Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "liboracle.so"
_SRC = _HERE / "jpeg_oracle.c"
_GOLDEN = _HERE.parent / "tests" / "golden"


def build(force: bool = False) -> Path:
    if force or not _SO.exists() or _SO.stat().st_mtime < _SRC.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return _SO


class OrcScan(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("hs", ctypes.c_int32 * 3), ("vs", ctypes.c_int32 * 3), ("qt_sel", ctypes.c_int32 * 3),
                ("dc_sel", ctypes.c_int32 * 3), ("ac_sel", ctypes.c_int32 * 3),
                ("restart_interval", ctypes.c_int32),
                ("mcu_count_h", ctypes.c_int32), ("mcu_count_v", ctypes.c_int32)]


class OrcProgScan(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp_frame", ctypes.c_int32),
                ("hs", ctypes.c_int32 * 3), ("vs", ctypes.c_int32 * 3), ("n_scan_comp", ctypes.c_int32),
                ("scan_comp", ctypes.c_int32 * 3), ("dc_sel", ctypes.c_int32 * 3), ("ac_sel", ctypes.c_int32 * 3),
                ("ss", ctypes.c_int32), ("se", ctypes.c_int32), ("ah", ctypes.c_int32), ("al", ctypes.c_int32),
                ("restart_interval", ctypes.c_int32), ("mcu_count_h", ctypes.c_int32), ("mcu_count_v", ctypes.c_int32)]


class OrcHuff(ctypes.Structure):
    _fields_ = [("first_code", ctypes.c_int32 * 17), ("count", ctypes.c_int32 * 17),
                ("first_sym", ctypes.c_int32 * 17), ("vals", ctypes.c_uint8 * 256)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_SO))
        L.orc_entropy_decode_baseline.restype = ctypes.c_int
        L.orc_reconstruct_baseline.restype = ctypes.c_int
        L.orc_progressive_scan.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a: np.ndarray, ty=ctypes.c_void_p):
    return a.ctypes.data_as(ty)


_T = None


def idct_table() -> np.ndarray:
    """float64 [8,8,8,8] = InverseDCT.idct_table (jpeg_decoder.py:1541-1553)."""
    global _T
    if _T is None:
        T = np.empty(4096, dtype=np.float64)
        lib().orc_idct_table(_p(T))
        _T = T.reshape(8, 8, 8, 8)
    return _T


def undo_zigzag(zz: np.ndarray) -> np.ndarray:
    zz = np.ascontiguousarray(zz, dtype=np.int16)
    out = np.empty(64, dtype=np.int16)
    lib().orc_undo_zigzag(_p(zz), _p(out))
    return out.reshape(8, 8)


def idct_xy(blocks_xy: np.ndarray) -> np.ndarray:
    """InverseDCT.__call__ on int16 [n,8,8] ([x,y]) blocks."""
    b = np.ascontiguousarray(blocks_xy, dtype=np.int16).reshape(-1, 64)
    out = np.empty_like(b)
    T = idct_table()
    L = lib()
    for i in range(b.shape[0]):
        L.orc_idct_xy(_p(b[i]), _p(T), _p(out[i]))
    return out.reshape(-1, 8, 8)


def dequant_idct(coef_zz: np.ndarray, qt_xy: np.ndarray):
    """(:869-872) zig-zag int16 [n,64] + reference-layout table -> (dequantised [n,8,8], idct [n,8,8])."""
    c = np.ascontiguousarray(coef_zz, dtype=np.int16).reshape(-1, 64)
    q = np.ascontiguousarray(qt_xy, dtype=np.int16).reshape(64)
    deq = np.empty_like(c)
    out = np.empty_like(c)
    T = idct_table()
    L = lib()
    for i in range(c.shape[0]):
        L.orc_dequant_idct(_p(c[i]), _p(q), _p(T), _p(deq[i]), _p(out[i]))
    return deq.reshape(-1, 8, 8), out.reshape(-1, 8, 8)


def operator_from_diagonals(src_shape, dst_shape, diag: np.ndarray) -> np.ndarray:
    """The upsample operator of ResizeGrid (:1588-1626) for ANY pair of shapes, as int16 numerators [n_out][n_in] over
    D = dx * dy (dx = dst_w - 1 where the width changes, else 1; dy likewise): griddata interpolates linearly inside the
    triangles of scipy's Delaunay triangulation of the SOURCE grid, and on a regular grid every triangle is half of a unit
    cell — so an output sample at (x (sw-1)/(dw-1), y (sh-1)/(dh-1)) takes barycentric weights of the three corners of the
    half it falls in.  Which diagonal cuts a cell is Qhull's choice; `diag[cx, cy]` (0: the diagonal through corner (0,0),
    1: the other one) is captured from the reference per source shape (tests/golden/upsample_diagonals.npz,
    tools/make_layout_goldens.py — which also checks this function against the operator captured for every pair of
    sampling factors 1..4)."""
    (sw, sh), (dw, dh) = src_shape, dst_shape
    dx = dw - 1 if sw != dw else 1
    dy = dh - 1 if sh != dh else 1
    D = dx * dy
    W = np.zeros((dw * dh, sw * sh), dtype=np.int16)
    for x in range(dw):
        cx, rx = (x, 0) if sw == dw else divmod(x * (sw - 1), dx)
        for y in range(dh):
            cy, ry = (y, 0) if sh == dh else divmod(y * (sh - 1), dy)
            FX, FY = rx * dy, ry * dx
            x1, y1 = min(cx + 1, sw - 1), min(cy + 1, sh - 1)
            kind = int(diag[min(cx, sw - 2), min(cy, sh - 2)]) if (rx and ry) else 0
            if kind == 0:
                if FX >= FY:
                    n = [((cx, cy), D - FX), ((x1, cy), FX - FY), ((x1, y1), FY)]
                else:
                    n = [((cx, cy), D - FY), ((cx, y1), FY - FX), ((x1, y1), FX)]
            else:
                if FX + FY <= D:
                    n = [((cx, cy), D - FX - FY), ((x1, cy), FX), ((cx, y1), FY)]
                else:
                    n = [((x1, y1), FX + FY - D), ((x1, cy), D - FY), ((cx, y1), D - FX)]
            row = W[x * dh + y]
            for (px, py), w in n:
                row[px * sh + py] += w
    return W


_DIAG = None


def load_W(src_shape, dst_shape) -> np.ndarray:
    """Upsample operator of the reference's ResizeGrid as integer numerators (a row sums to its denominator): the
    matrices captured in round 1 for the common layouts, operator_from_diagonals for every other pair of shapes."""
    global _DIAG
    f = _GOLDEN / f"upsample_W_{src_shape[0]}x{src_shape[1]}_{dst_shape[0]}x{dst_shape[1]}.npy"
    if f.exists():
        return np.load(f).astype(np.int16)
    if _DIAG is None:
        _DIAG = dict(np.load(_GOLDEN / "upsample_diagonals.npz"))
    return operator_from_diagonals(tuple(src_shape), tuple(dst_shape), _DIAG[f"{src_shape[0]}x{src_shape[1]}"])


def upsample(block: np.ndarray, dst_shape) -> np.ndarray:
    """ResizeGrid.__call__ (:1588-1626) for int16 [sw,sh] -> [dw,dh]."""
    block = np.ascontiguousarray(block, dtype=np.int16)
    W = np.ascontiguousarray(load_W(block.shape, dst_shape), dtype=np.int16)
    out = np.empty(dst_shape, dtype=np.int16)
    lib().orc_upsample(_p(block), block.size, _p(W), out.size, _p(out))
    return out


def ycbcr_to_rgb(ycc: np.ndarray) -> np.ndarray:
    ycc = np.ascontiguousarray(ycc, dtype=np.int16)
    out = np.empty(ycc.shape, dtype=np.uint8)
    lib().orc_ycbcr_to_rgb(_p(ycc), ctypes.c_long(ycc.size // 3), _p(out))
    return out


def _scan_struct(parsed, scan) -> OrcScan:
    s = OrcScan()
    s.width, s.height = parsed.image_width, parsed.image_height
    comps = [parsed.color_components[cid] for cid in scan.component_ids]
    s.ncomp = len(comps)
    for i, (cid, c) in enumerate(zip(scan.component_ids, comps)):
        s.hs[i], s.vs[i] = c.horizontal_sampling, c.vertical_sampling
        s.qt_sel[i] = c.quantization_table_id & 3
        s.dc_sel[i] = scan.huffman_tables_id[cid].dc & 3
        s.ac_sel[i] = scan.huffman_tables_id[cid].ac & 3
    s.restart_interval = scan.restart_interval
    s.mcu_count_h, s.mcu_count_v = scan.mcu_count_h, scan.mcu_count_v
    return s


def _huff_arrays(scan):
    dc = (OrcHuff * 4)()
    ac = (OrcHuff * 4)()
    L = lib()
    for dest, spec in scan.huffman.items():
        tgt = ac if dest >> 4 else dc
        vals = np.zeros(256, dtype=np.uint8)
        vals[:spec.vals.size] = spec.vals[:256]
        L.orc_build_huffman(_p(np.ascontiguousarray(spec.bits)), _p(vals), ctypes.byref(tgt[dest & 3]))
    return dc, ac


def blocks_per_mcu(parsed, scan) -> int:
    comps = [parsed.color_components[cid] for cid in scan.component_ids]
    return sum(c.repeat for c in comps) if len(comps) > 1 else 1


def entropy_decode(parsed, scan=None):
    """baseline_dct_scan's entropy part -> (coef int16 [nblocks,64] zig-zag, status, end_pos)."""
    scan = scan or parsed.scans[0]
    s = _scan_struct(parsed, scan)
    dc, ac = _huff_arrays(scan)
    nblk = scan.mcu_count * blocks_per_mcu(parsed, scan)
    coef = np.zeros((nblk, 64), dtype=np.int16)
    raw = np.frombuffer(parsed.raw, dtype=np.uint8)
    end = ctypes.c_int64(0)
    st = lib().orc_entropy_decode_baseline(_p(raw), ctypes.c_int64(raw.size), ctypes.c_int64(scan.entropy_start),
                                           ctypes.byref(s), dc, ac, _p(coef), ctypes.byref(end))
    return coef, st, end.value


def reconstruct(parsed, coef: np.ndarray, scan=None, want_idct: bool = False):
    """Dequant + IDCT + upsample + crop + colour for an interleaved baseline scan.

    Returns dict(planes=int16 (W,H,C) [golden G5], rgb=uint8 (W,H,3)|(W,H) [G6], idct=int16 [nblk,8,8] [G3]).
    """
    scan = scan or parsed.scans[0]
    s = _scan_struct(parsed, scan)
    W, H, nc = parsed.image_width, parsed.image_height, s.ncomp
    qt = np.zeros((4, 64), dtype=np.int16)
    for dest, xy in parsed.quantization_tables.items():
        qt[dest & 3] = xy.reshape(64)
    T = idct_table()
    ups = (ctypes.c_void_p * 3)()
    keep = []
    if nc > 1:
        for i, cid in enumerate(scan.component_ids):
            c = parsed.color_components[cid]
            if tuple(c.shape) != tuple(parsed.sample_shape):
                Wm = np.ascontiguousarray(load_W(c.shape, parsed.sample_shape), dtype=np.int16)
                keep.append(Wm)
                ups[i] = Wm.ctypes.data
    coef = np.ascontiguousarray(coef, dtype=np.int16)
    idct = np.empty_like(coef) if want_idct else None
    planes = np.zeros((W, H, nc), dtype=np.int16)
    rgb = np.empty((W, H, 3) if nc == 3 else (W, H), dtype=np.uint8)
    st = lib().orc_reconstruct_baseline(ctypes.byref(s), _p(coef), _p(qt), _p(T), ups,
                                        _p(idct) if want_idct else None, _p(planes), _p(rgb))
    if st:
        raise RuntimeError(f"oracle: reconstruct failed with status {st}")
    return {"planes": planes, "rgb": rgb, "idct": idct.reshape(-1, 8, 8) if want_idct else None}


class _FrameScan:
    """A pseudo scan covering all frame components interleaved (geometry of the final pass, :1319-1362)."""

    def __init__(self, parsed):
        comps = parsed.color_components
        self.component_ids = list(comps.keys())
        self.huffman_tables_id = {cid: type("T", (), {"dc": 0, "ac": 0x10})() for cid in comps}
        self.restart_interval = 0
        if len(comps) > 1:
            mw = 8 * max(c.horizontal_sampling for c in comps.values())
            mh = 8 * max(c.vertical_sampling for c in comps.values())
        else:
            mw = mh = 8
        self.mcu_count_h = -(-parsed.image_width // mw)
        self.mcu_count_v = -(-parsed.image_height // mh)
        self.mcu_count = self.mcu_count_h * self.mcu_count_v
        self.huffman = {}


def progressive_entropy_decode(parsed, upto=None):
    """All scans of a progressive file (:908-1304) -> coefficient store int16 [nblocks,64] zig-zag, blocks in the
    interleaved order of the baseline seam.  `upto` = number of scans to apply (default all)."""
    frame = _FrameScan(parsed)
    nblk = frame.mcu_count * blocks_per_mcu(parsed, frame)
    coef = np.zeros((nblk, 64), dtype=np.int16)
    raw = np.frombuffer(parsed.raw, dtype=np.uint8)
    order = {cid: i for i, cid in enumerate(parsed.color_components)}
    comps = list(parsed.color_components.values())
    status = 0
    for scan in parsed.scans[:upto]:
        ps = OrcProgScan()
        ps.width, ps.height, ps.ncomp_frame = parsed.image_width, parsed.image_height, len(comps)
        for i, c in enumerate(comps):
            ps.hs[i], ps.vs[i] = c.horizontal_sampling, c.vertical_sampling
        ps.n_scan_comp = len(scan.component_ids)
        for i, cid in enumerate(scan.component_ids):
            ps.scan_comp[i] = order[cid]
            ps.dc_sel[i] = scan.huffman_tables_id[cid].dc & 3
            ps.ac_sel[i] = scan.huffman_tables_id[cid].ac & 3
        ps.ss, ps.se, ps.ah, ps.al = scan.spectral_start, scan.spectral_end, scan.bit_high, scan.bit_low
        ps.restart_interval = scan.restart_interval
        ps.mcu_count_h, ps.mcu_count_v = scan.mcu_count_h, scan.mcu_count_v
        dc, ac = _huff_arrays(scan)
        end = ctypes.c_int64(0)
        status = lib().orc_progressive_scan(_p(raw), ctypes.c_int64(raw.size), ctypes.c_int64(scan.entropy_start),
                                            ctypes.byref(ps), dc, ac, _p(coef), ctypes.byref(end))
        if status:
            break
    return coef, status, frame


def decode(raw: bytes, want_idct: bool = False) -> Dict[str, Optional[np.ndarray]]:
    """Whole reference path for one interleaved baseline file.  Uses the product's header parser for the
    container (host logic, checked separately against the reference's attribute surface)."""
    from pyjpegdecoder_amd._parse import parse_jpeg
    parsed = parse_jpeg(raw)
    if parsed.scan_mode == "progressive_dct" or len(parsed.scans) > 1:     # scan by scan (non-interleaved baseline too)
        coef, st, frame = progressive_entropy_decode(parsed)
        if st:
            raise RuntimeError(f"oracle: progressive scan status {st}")
        out = reconstruct(parsed, coef, scan=frame, want_idct=want_idct)
        out["coef"] = coef
        out["end_pos"] = parsed.scans[-1].entropy_end
        out["parsed"] = parsed
        return out
    coef, st, end = entropy_decode(parsed)
    if st:
        raise RuntimeError(f"oracle: entropy decode status {st}")
    out = reconstruct(parsed, coef, want_idct=want_idct)
    out["coef"] = coef
    out["end_pos"] = end
    out["parsed"] = parsed
    return out
