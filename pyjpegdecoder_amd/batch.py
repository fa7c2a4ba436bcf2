"""Batched decode: many JPEG files -> one plan on one MI355X.

Python host code (header parsing + restart segmentation, `_parse.py`) prepares the arrays the C ABI
takes (`include/mijpeg.h`); all pixel work happens in libmijpeg.so's HIP kernels.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _binding as B
from ._parse import ParsedJpeg, ScanInfo, parse_jpeg
from .errors import CorruptedJpeg, UnsupportedJpeg

_STATUS_TEXT = {
    B.MJ_ST_BAD_CODE: "Failed to decode image (no Huffman code within 16 bits).",
    B.MJ_ST_OVERRUN: "Failed to decode image (a restart segment ends before its MCUs do).",
    B.MJ_ST_DESYNC: "Failed to decode image (restart markers are not where the restart interval puts them).",
}


def check_supported(p: ParsedJpeg) -> ScanInfo:
    """What the MI355X path accepts: one interleaved baseline scan covering all frame components, or a
    progressive file whose DC scans are interleaved (libjpeg's scripts) and whose AC scans are single-component."""
    if not p.scans:
        raise CorruptedJpeg("No scan found in the file.")
    if p.scan_mode == "progressive_dct":
        comps = list(p.color_components.values())
        if len(comps) > 1:
            hmax = max(c.horizontal_sampling for c in comps)
            vmax = max(c.vertical_sampling for c in comps)
            for c in comps:
                h, v = c.horizontal_sampling, c.vertical_sampling
                if (h, v) != (1, 1) and (h, v) != (hmax, vmax):
                    # the reference's final pass resizes every 8x8 block to the full MCU shape and stores it into a
                    # ratio x ratio region (jpeg_decoder.py:1345-1358): a shape mismatch for such a component (ValueError)
                    raise UnsupportedJpeg(f"Progressive files need every component at 1x1 or at the full resolution "
                                          f"(a {h}x{v} component under {hmax}x{vmax} is not supported).")
        for sc in p.scans:
            # the reference's own checks (jpeg_decoder.py:917-934, :966-967)
            if sc.spectral_start == 0 and sc.spectral_end != 0 or sc.spectral_start > sc.spectral_end:
                raise CorruptedJpeg("Progressive JPEG images cannot contain both DC and AC values in the same scan.")
            if sc.bit_high != 0 and sc.bit_high - sc.bit_low != 1:
                raise CorruptedJpeg("Progressive JPEG images cannot contain more than 1 bit for each value on a refining scan.")
            if sc.spectral_start > 0 and len(sc.component_ids) > 1:
                raise CorruptedJpeg("An AC progressive scan can only have a single color component.")
            if sc.spectral_start == 0 and len(sc.component_ids) == 1 and len(p.color_components) > 1:
                c = p.color_components[sc.component_ids[0]]
                if c.horizontal_sampling > 1 or c.vertical_sampling > 1:
                    raise UnsupportedJpeg("Single-component DC scan of a component with sampling > 1 "
                                          "(the reference steps such a scan's blocks by the component's MCU size, "
                                          "jpeg_decoder.py:993-994, and runs off its array: IndexError) is not supported.")
        return p.scans[0]
    if p.scan_mode != "baseline_dct":
        raise UnsupportedJpeg("Encoding mode not supported. Only 'Baseline DCT' and 'Progressive DCT' are supported.")
    if len(p.scans) == 1 and len(p.scans[0].component_ids) == len(p.color_components):
        return p.scans[0]
    # Non-interleaved baseline: one scan per component.  The reference only gets these right when no component is
    # subsampled (it assigns the resized MCU into an 8x8 slot otherwise, jpeg_decoder.py:882-891), so that is the scope.
    comps = p.color_components
    single = all(len(sc.component_ids) == 1 for sc in p.scans)
    once = sorted(sc.component_ids[0] for sc in p.scans) == sorted(comps) if single else False
    flat = all(c.horizontal_sampling == 1 and c.vertical_sampling == 1 for c in comps.values())
    if not (single and once and flat):
        raise UnsupportedJpeg("Baseline files with several scans are supported when every scan holds one component, "
                              "every component has one scan and no component is subsampled.")
    return p.scans[0]


def is_scan_list(p: ParsedJpeg) -> bool:
    """Files decoded scan by scan into the coefficient store (mj_batch.scans): progressive ones and baseline files
    with one scan per component."""
    return p.scan_mode == "progressive_dct" or len(p.scans) > 1


def _segments_of(scan: ScanInfo, off: int):
    """(begin[], end[]) blob offsets of a scan's restart segments; checks the marker count (:898 is count driven)."""
    so = scan.segment_offsets
    if so is None:          # headers-only parse: one byte range, the GPU segments it (MJ_FLAG_GPU_SEGMENT)
        return (np.array([scan.entropy_start + off], dtype=np.int64), np.array([scan.entropy_end + off], dtype=np.int64))
    if scan.restart_interval > 0:
        want = -(-scan.mcu_count // scan.restart_interval)
        if so.size - 1 != want:
            raise CorruptedJpeg(f"Failed to decode image ({so.size - 2} restart markers found, {want - 1} expected).")
        b = so[:-1].copy()
        e = np.concatenate([so[1:-1] - 2, so[-1:]])
    else:
        b = np.array([scan.entropy_start], dtype=np.int64)
        e = np.array([scan.entropy_end], dtype=np.int64)
    return b + off, e + off


@dataclass
class PreparedBatch:
    """Host-side arrays of one mj_batch (kept alive for the life of the plan)."""
    parsed: List[ParsedJpeg]
    blob: np.ndarray
    file_offsets: np.ndarray
    descs: ctypes.Array
    seg_begin: np.ndarray
    seg_end: np.ndarray
    huff: ctypes.Array
    n_huff: int
    qt: np.ndarray
    layout: int
    flags: int
    shapes: List[Tuple[int, int, int]] = field(default_factory=list)   # (W, H, ncomp)
    scans: Optional[ctypes.Array] = None                               # progressive batches: mj_scan_desc[]
    n_scans: int = 0

    def to_c(self, blob_device_ptr: int = 0) -> B.BatchC:
        b = B.BatchC()
        b.n_images = len(self.parsed)
        b.images = ctypes.cast(self.descs, ctypes.POINTER(B.ImageDescC))
        if blob_device_ptr:
            b.blob, b.blob_mem = blob_device_ptr, B.MJ_MEM_DEVICE
        else:
            b.blob, b.blob_mem = self.blob.ctypes.data, B.MJ_MEM_HOST
        b.blob_len = int(self.blob.size)
        b.n_segments = int(self.seg_begin.size)
        b.seg_begin, b.seg_end = self.seg_begin.ctypes.data, self.seg_end.ctypes.data
        b.n_huff = self.n_huff
        b.huff = ctypes.cast(self.huff, ctypes.POINTER(B.HuffSpecC))
        b.n_qt = self.qt.shape[0]
        b.qt = self.qt.ctypes.data
        b.layout, b.flags = self.layout, self.flags
        b.n_scans = self.n_scans
        b.scans = ctypes.cast(self.scans, ctypes.POINTER(B.ScanDescC)) if self.n_scans else None
        return b


def prepare_batch(files: Sequence[bytes], layout: int = B.MJ_LAYOUT_XMAJOR, flags: int = 0,
                  parsed: Optional[List[ParsedJpeg]] = None) -> PreparedBatch:
    """Parse every file and build the descriptor / table / segment arrays of include/mijpeg.h."""
    if parsed is None:
        parsed = [parse_jpeg(f) for f in files]
    n = len(parsed)
    sizes = np.array([len(p.raw) for p in parsed], dtype=np.int64)
    # keep every file 4-byte aligned inside the blob (stage 1 fetches aligned dwords)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum((sizes + 3) & ~3)
    # stage 1 reads up to 512 bytes ahead of a segment (mijpeg.h, mj_batch.blob_mem): keep 1 KiB of readable slack behind the last file
    blob = np.zeros(int(offs[-1]) + 1024, dtype=np.uint8)
    if parsed and any(p.headers_only for p in parsed):
        if not all(p.headers_only for p in parsed):
            raise UnsupportedJpeg("A batch is segmented either on the GPU (headers-only parse) or on the host; split it.")
        flags |= B.MJ_FLAG_GPU_SEGMENT
    descs = (B.ImageDescC * n)()
    huff_ids: Dict[bytes, int] = {}
    huff_list: List[Tuple[np.ndarray, np.ndarray]] = []
    qt_ids: Dict[bytes, int] = {}
    qt_list: List[np.ndarray] = []
    seg_b: List[np.ndarray] = []
    seg_e: List[np.ndarray] = []
    n_seg_total = 0
    shapes = []

    def huff_id(spec) -> int:
        key = spec.bits.tobytes() + spec.vals.tobytes()
        if key not in huff_ids:
            huff_ids[key] = len(huff_list)
            huff_list.append((spec.bits, spec.vals))
        return huff_ids[key]

    def qt_id(zz: np.ndarray) -> int:
        key = zz.tobytes()
        if key not in qt_ids:
            qt_ids[key] = len(qt_list)
            qt_list.append(zz.astype(np.uint16))
        return qt_ids[key]

    progressive = is_scan_list(parsed[0]) if parsed else False       # "progressive" = scan-list batch from here on
    scan_list: List[B.ScanDescC] = []
    for i, p in enumerate(parsed):
        scan = check_supported(p)
        if is_scan_list(p) != progressive:
            raise UnsupportedJpeg("A batch holds either single-scan baseline files or scan-by-scan (progressive / "
                                  "non-interleaved) files; split it.")
        blob[offs[i]:offs[i] + sizes[i]] = np.frombuffer(p.raw, dtype=np.uint8)
        d = descs[i]
        if progressive:
            comp_ids = list(p.color_components)
            d.width, d.height, d.ncomp = p.image_width, p.image_height, len(comp_ids)
            for c, cid in enumerate(comp_ids):
                comp = p.color_components[cid]
                d.hs[c], d.vs[c] = comp.horizontal_sampling, comp.vertical_sampling
                # baseline files decoded scan by scan dequantise each component with the table in force at ITS scan
                # (:869); progressive files with the tables in force at the final pass (:1348)
                qsrc = p.quantization_zz
                if p.scan_mode == "baseline_dct":
                    qsrc = next((sc.quantization_zz for sc in p.scans if cid in sc.component_ids and sc.quantization_zz), qsrc)
                if comp.quantization_table_id not in qsrc:
                    raise CorruptedJpeg("Scan uses a quantization table that the file does not define.")
                d.qt_sel[c] = qt_id(qsrc[comp.quantization_table_id])
            if d.ncomp == 1:
                d.hs[0] = d.vs[0] = 1
                mw = mh = 8
            else:
                mw, mh = 8 * max(d.hs[c] for c in range(3)), 8 * max(d.vs[c] for c in range(3))
            d.mcu_count_h, d.mcu_count_v = -(-p.image_width // mw), -(-p.image_height // mh)
            d.restart_interval, d.n_segments, d.first_segment = 0, 0, 0
            for sc in p.scans:
                sd = B.ScanDescC()
                sd.image, sd.n_comp = i, len(sc.component_ids)
                for k, cid in enumerate(sc.component_ids):
                    sd.comp[k] = comp_ids.index(cid)
                    tabs = sc.huffman_tables_id[cid]
                    need_dc, need_ac = sc.spectral_start == 0 and sc.bit_high == 0, sc.spectral_end > 0
                    if (need_dc and tabs.dc not in sc.huffman) or (need_ac and tabs.ac not in sc.huffman):
                        raise CorruptedJpeg("Scan uses a Huffman table that the file does not define.")
                    sd.dc_sel[k] = huff_id(sc.huffman[tabs.dc]) if need_dc else 0
                    sd.ac_sel[k] = huff_id(sc.huffman[tabs.ac]) if need_ac else 0
                sd.ss, sd.se, sd.ah, sd.al = sc.spectral_start, sc.spectral_end, sc.bit_high, sc.bit_low
                sd.restart_interval = sc.restart_interval
                sd.mcu_count_h, sd.mcu_count_v = sc.mcu_count_h, sc.mcu_count_v
                b_, e_ = _segments_of(sc, int(offs[i]))
                sd.n_segments, sd.first_segment = int(b_.size), n_seg_total
                n_seg_total += int(b_.size)
                seg_b.append(b_)
                seg_e.append(e_)
                scan_list.append(sd)
            shapes.append((p.image_width, p.image_height, d.ncomp))
            continue
        d.width, d.height, d.ncomp = p.image_width, p.image_height, len(scan.component_ids)
        if list(scan.component_ids) != list(p.color_components)[:len(scan.component_ids)]:
            # descriptors are filled by scan position and stage 2 takes position 0 for Y, 1 and 2 for Cb and Cr
            raise UnsupportedJpeg("The scan lists the color components in another order than the frame.")
        for c, cid in enumerate(scan.component_ids):
            comp = p.color_components[cid]
            d.hs[c], d.vs[c] = comp.horizontal_sampling, comp.vertical_sampling
            if comp.quantization_table_id not in p.quantization_zz:
                raise CorruptedJpeg("Scan uses a quantization table that the file does not define.")
            d.qt_sel[c] = qt_id(p.quantization_zz[comp.quantization_table_id])
            tabs = scan.huffman_tables_id[cid]
            if tabs.dc not in scan.huffman or tabs.ac not in scan.huffman:
                raise CorruptedJpeg("Scan uses a Huffman table that the file does not define.")
            d.dc_sel[c] = huff_id(scan.huffman[tabs.dc])
            d.ac_sel[c] = huff_id(scan.huffman[tabs.ac])
        if d.ncomp == 1:
            d.hs[0] = d.vs[0] = 1     # a single-component scan is always 8x8 MCUs (jpeg_decoder.py:595-598)
        d.restart_interval = scan.restart_interval
        d.mcu_count_h, d.mcu_count_v = scan.mcu_count_h, scan.mcu_count_v
        b, e = _segments_of(scan, int(offs[i]))
        d.n_segments = int(b.size)
        d.first_segment = n_seg_total
        n_seg_total += int(b.size)
        seg_b.append(b)
        seg_e.append(e)
        shapes.append((p.image_width, p.image_height, d.ncomp))

    huff = (B.HuffSpecC * max(1, len(huff_list)))()
    for k, (bits, vals) in enumerate(huff_list):
        huff[k].bits[:] = bits.tolist()
        v = np.zeros(256, dtype=np.uint8)
        v[:min(256, vals.size)] = vals[:256]
        huff[k].vals[:] = v.tolist()
    qt = np.ascontiguousarray(np.stack(qt_list), dtype=np.uint16)
    scans_c = (B.ScanDescC * len(scan_list))(*scan_list) if scan_list else None
    return PreparedBatch(scans=scans_c, n_scans=len(scan_list), parsed=parsed, blob=blob, file_offsets=offs, descs=descs,
                         seg_begin=np.ascontiguousarray(np.concatenate(seg_b), dtype=np.int64),
                         seg_end=np.ascontiguousarray(np.concatenate(seg_e), dtype=np.int64),
                         huff=huff, n_huff=len(huff_list), qt=qt, layout=layout, flags=flags, shapes=shapes)


# numpy view of mj_image_desc (include/mijpeg.h), to read what the native front end filled in without a Python loop
_DESC_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("ncomp", "<i4"), ("hs", "<i4", 3), ("vs", "<i4", 3),
                        ("qt_sel", "<i4", 3), ("dc_sel", "<i4", 3), ("ac_sel", "<i4", 3), ("restart_interval", "<i4"),
                        ("mcu_count_h", "<i4"), ("mcu_count_v", "<i4"), ("n_segments", "<i4"), ("first_segment", "<i8")])
assert _DESC_DTYPE.itemsize == ctypes.sizeof(B.ImageDescC)


def prepare_batch_native(files: Sequence[bytes], layout: int = B.MJ_LAYOUT_XMAJOR, flags: int = 0, n_threads: int = 0,
                         staging: Optional[np.ndarray] = None, split: bool = False):
    """`prepare_batch` for a GPU-segmented batch of everyday baseline files through libmijpeg.so's host front end
    (``mj_host_assemble``: header parse and assembly on host threads).  Returns None when the front end declines a file —
    the caller then takes the Python path, which also raises the reference's exceptions — and, when the files are fine but
    do not belong in one plan (several sampling layouts; files with and without restart markers), the groups they fall
    into as lists of indices.  ``split=True``: files the front end does not take do not sink the batch; the result is then
    ``(groups, declined)`` — index lists for the front end, grouped as above, and the indices left for the Python path.
    ``staging``: a uint8 buffer to build the blob in (reused between batches by BatchDecoder)."""
    n = len(files)
    if n == 0 or not all(type(f) is bytes for f in files):
        return None
    lib = B.load_library()
    sizes = np.fromiter(map(len, files), dtype=np.int64, count=n)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum((sizes + 3) & ~3, out=offs[1:])
    blob_len = int(offs[-1]) + 1024
    blob = staging[:blob_len] if staging is not None and staging.size >= blob_len else np.empty(blob_len, dtype=np.uint8)
    descs = (B.ImageDescC * n)()
    seg_b, seg_e = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    huff = (B.HuffSpecC * (6 * n))()
    qt = np.empty((3 * n, 64), dtype=np.uint16)
    job = B.HostJobC()
    job.n_files = n
    ptrs = (ctypes.c_char_p * n)(*files)
    job.files = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_char_p))
    job.sizes, job.file_off = sizes.ctypes.data, offs.ctypes.data
    job.blob, job.blob_len = blob.ctypes.data, blob_len
    job.images = ctypes.cast(descs, ctypes.POINTER(B.ImageDescC))
    job.seg_begin, job.seg_end = seg_b.ctypes.data, seg_e.ctypes.data
    job.huff, job.huff_cap = ctypes.cast(huff, ctypes.POINTER(B.HuffSpecC)), 6 * n
    job.qt, job.qt_cap = qt.ctypes.data, 3 * n
    if n_threads <= 0:
        import os
        n_threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    job.n_threads = n_threads
    skip = np.zeros(n, dtype=np.uint8)
    if split:
        job.skip = skip.ctypes.data
    rc = lib.mj_host_assemble(ctypes.byref(job))
    if rc == B.MJ_HOST_DECLINED:
        return ([], list(range(n))) if split else None
    if rc != B.MJ_OK:
        raise B.BackendError(f"mj_host_assemble failed ({rc})")
    m = int(job.n_accepted)
    accepted = np.flatnonzero(skip == 0)
    d = np.frombuffer(descs, dtype=_DESC_DTYPE)[:m]
    lay = np.concatenate([d["ncomp"][:, None], d["hs"], d["vs"], (d["restart_interval"] > 0)[:, None]], axis=1)
    mixed = bool((lay != lay[0]).any())
    if mixed or m < n:
        # several sampling layouts, or files with and without restart markers (only a batch without any can be cut into
        # chunks when the GPU finds the markers): one plan each — the caller gets the groups (lists of file indices)
        _, inverse = np.unique(lay, axis=0, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        groups = [accepted[np.flatnonzero(inverse == g)].tolist() for g in range(int(inverse.max()) + 1)]
        return (groups, np.flatnonzero(skip).tolist()) if split else groups
    shapes = list(zip(d["width"].tolist(), d["height"].tolist(), d["ncomp"].tolist()))
    if split:
        return ([list(range(n))], [])         # (the caller assembles group by group; this pass only sorted the files)
    return PreparedBatch(parsed=[None] * n, blob=blob, file_offsets=offs, descs=descs, seg_begin=seg_b, seg_end=seg_e,
                         huff=huff, n_huff=int(job.n_huff), qt=np.ascontiguousarray(qt[:max(1, int(job.n_qt))]),
                         layout=layout, flags=flags | B.MJ_FLAG_GPU_SEGMENT, shapes=shapes)


def raise_for_status(status: np.ndarray):
    bad = np.flatnonzero(status)
    if bad.size:
        i = int(bad[0])
        if int(status[i]) == B.MJ_ST_INTERNAL:          # not the file's fault (include/mijpeg.h)
            raise B.BackendError(f"image {i}: a fused launch gave up waiting for its decoder wavefronts (internal error)")
        raise CorruptedJpeg(f"image {i}: {_STATUS_TEXT.get(int(status[i]), 'decode failed')}")


class BatchDecoder:
    """Decode lists of baseline JPEG files on one GPU.

    >>> dec = BatchDecoder(device=0)
    >>> images = dec.decode([open(p, 'rb').read() for p in paths])      # list of uint8 (W,H,3) arrays
    """

    def __init__(self, device: int = 0, layout: str = "xmajor", exact_only: bool = False, spec_refine: bool = False,
                 segment: str = "gpu", native_host: bool = True, gpu_segment_min_files: int = 8):
        self.ctx = B.Context(device)
        # "planar" / "planar_rowmajor": the components apart, (3, W, H) / (3, H, W) per colour image
        self.layout = {"xmajor": B.MJ_LAYOUT_XMAJOR, "rowmajor": B.MJ_LAYOUT_ROWMAJOR, "planar": B.MJ_LAYOUT_PLANAR_XMAJOR,
                       "planar_rowmajor": B.MJ_LAYOUT_PLANAR_ROWMAJOR}[layout]
        # exact_only: stage 2 uses the reference's summation order for every block (slow; for A/B checks)
        # spec_refine: progressive AC refinement as ITU-T T.81 defines it instead of the reference's behaviour
        self.base_flags = (B.MJ_FLAG_EXACT_ONLY if exact_only else 0) | (B.MJ_FLAG_SPEC_REFINE if spec_refine else 0)
        # segment="gpu" (the default since round 5): the host parses headers only; restart markers and the end of each
        # baseline scan are found on the GPU — the host's NumPy marker search was 76x the GPU step for a
        # batch of 1024 files.  Files the GPU scan hands back (MJ_ST_TAIL: something other than EOI follows the scan) and
        # progressive files take the host path, as do calls with fewer than `gpu_segment_min_files` files (below).
        # segment="host": the marker loop's restart segmentation in Python for every file (jpeg_decoder.py:667-669, :898).
        if segment not in ("host", "gpu"):
            raise ValueError("segment must be 'host' or 'gpu'")
        self.gpu_segment = segment == "gpu"
        # native_host: with segment="gpu", decode_device reads headers and assembles batches in libmijpeg.so's
        # multi-threaded host front end instead of _parse.py (identical arrays; anything unusual is handed back to Python)
        self.native_host = native_host
        # segment="gpu" applies from this many files per call on (or from 4 MiB of files on): a handful of ordinary files is
        # segmented on the host, which costs ~1 ms per MB and lets files with restart markers take the chunked stage-1 form
        # (it needs the segment lengths at plan time; one 1080p file: 2.3 ms instead of 6.6)
        self.gpu_segment_min_files = gpu_segment_min_files
        self._staging: Optional[np.ndarray] = None

    def _gpu_segment_for(self, files: Sequence[bytes]) -> bool:
        """segment="gpu" applies from `gpu_segment_min_files` files per call on, or from 4 MiB of files on (see __init__)."""
        return self.gpu_segment and (len(files) >= self.gpu_segment_min_files or sum(map(len, files)) > (4 << 20))

    def plan(self, files: Sequence[bytes], flags: int = 0, blob_device_ptr: int = 0):
        """(prepared batch, plan) for files of ONE kind.  Like decode(), a handful of files is segmented on the host (the plan
        then knows the segment lengths and can take the chunked stage-1 form); from `gpu_segment_min_files` files on the GPU
        finds the markers, and a caller that executes such a plan itself handles MJ_ST_TAIL (a file the scan handed back)."""
        parsed = [parse_jpeg(f, headers_only=True) for f in files] if self._gpu_segment_for(files) else None
        if parsed is not None and not all(p.headers_only for p in parsed):
            parsed = None                       # progressive / multi-scan files: host segmentation for the whole batch
        prep = prepare_batch(files, self.layout, flags | self.base_flags, parsed)
        plan = B.Plan(self.ctx, prep.to_c(blob_device_ptr), {"prep": prep, "n_images": len(prep.parsed)})
        return prep, plan

    def _shape(self, w: int, h: int, nc: int) -> tuple:
        """Array shape of one decoded image in this decoder's layout."""
        wh = (w, h) if (self.layout & 1) == B.MJ_LAYOUT_XMAJOR else (h, w)
        if nc != 3:
            return wh
        return (3,) + wh if self.layout >= B.MJ_LAYOUT_PLANAR_XMAJOR else wh + (3,)

    def split_outputs(self, prep: PreparedBatch, flat: np.ndarray, per_pixel: int = 1) -> List[np.ndarray]:
        out, off = [], 0
        for (w, h, nc) in prep.shapes:
            n = w * h * nc * per_pixel
            out.append(flat[off:off + n].reshape(self._shape(w, h, nc)))
            off += n
        return out

    def decode(self, files: Sequence[bytes], return_seams: bool = False):
        """Decode files that may mix sampling layouts (one plan per layout)."""
        gpu_segment = self._gpu_segment_for(files)
        parsed = [parse_jpeg(f, headers_only=gpu_segment) for f in files]
        groups: Dict[tuple, List[int]] = {}
        for i, p in enumerate(parsed):
            check_supported(p)
            comps = list(p.color_components.values())
            key = (p.scan_mode, len(comps), p.headers_only, is_scan_list(p), p.headers_only and p.restart_interval > 0) + (tuple((c.horizontal_sampling, c.vertical_sampling) for c in comps) if len(comps) > 1 else ())
            groups.setdefault(key, []).append(i)
        results: List[Optional[np.ndarray]] = [None] * len(files)
        seams: List[Optional[dict]] = [None] * len(files)
        flags = ((B.MJ_FLAG_KEEP_PLANES | B.MJ_FLAG_KEEP_IDCT) if return_seams else 0) | self.base_flags
        work = [(idxs, 0) for idxs in groups.values()]
        while work:
            idxs, extra = work.pop(0)
            prep = prepare_batch([files[i] for i in idxs], self.layout, flags | extra, [parsed[i] for i in idxs])
            plan = B.Plan(self.ctx, prep.to_c(), {"prep": prep, "n_images": len(idxs)})
            try:
                plan.execute()
                plan.sync()
                out = plan.read(rgb=True, coef=return_seams, planes=return_seams, idct=return_seams)
                redo = [i for k, i in enumerate(idxs) if out["status"][k] == B.MJ_ST_TAIL]
                if redo:                        # the GPU scan met something other than EOI after the scan: host parse
                    for i in redo:
                        parsed[i] = parse_jpeg(files[i])
                        check_supported(parsed[i])
                    work.append((redo, extra))
                    out["status"][[k for k, i in enumerate(idxs) if i in redo]] = 0
                again = [i for k, i in enumerate(idxs) if out["status"][k] == B.MJ_ST_UNCONVERGED]
                if again:                       # the synchronisation rounds had not settled: the serial walk for these
                    work.append((again, extra | B.MJ_FLAG_NO_SYNC))
                    out["status"][[k for k, i in enumerate(idxs) if i in again]] = 0
                raise_for_status(out["status"])
                imgs = self.split_outputs(prep, out["rgb"])
                for k, i in enumerate(idxs):
                    if i in redo or i in again:
                        continue
                    results[i] = imgs[k]
                    if return_seams:
                        b0, _ = plan.image_offsets(k)
                        b1 = plan.image_offsets(k + 1)[0] if k + 1 < len(idxs) else plan.info.total_blocks
                        w, h, nc = prep.shapes[k]
                        po = sum(s[0] * s[1] * s[2] for s in prep.shapes[:k])
                        seams[i] = {"coef": out["coef"][b0:b1], "idct": out["idct"][b0:b1].reshape(-1, 8, 8),
                                    "planes": out["planes"][po:po + w * h * nc].reshape(w, h, nc)}
            finally:
                plan.close()
        return (results, seams) if return_seams else results

    def _staging_for(self, files: Sequence[bytes]) -> np.ndarray:
        """Host buffer the native front end builds the blob in, kept between calls (first touch of a fresh 300 MB
        allocation costs as much as the parse)."""
        need = sum(map(len, files)) + 3 * len(files) + 1024
        if self._staging is None or self._staging.size < need:
            self._staging = np.empty(need + need // 4, dtype=np.uint8)
        return self._staging

    def decode_device(self, files: Sequence[bytes], parts: Optional[int] = None):
        """Like :meth:`decode`, but the pixels stay in HBM: a list of ``torch.uint8`` tensors on this decoder's GPU,
        views into one packed buffer per plan (zero-copy for any DLPack consumer via ``tensor.__dlpack__()``).
        torch is only the allocator here; import it before this package.

        With ``segment="gpu"`` a batch of everyday baseline files never meets the Python parser: libmijpeg.so's host
        front end (``mj_host_assemble``) reads the headers and assembles the batch on host threads; whatever it declines
        takes the Python path below, which raises the reference's exceptions.  A large batch on that route goes as ``parts``
        plans of 256 files or more (up to four) through :meth:`decode_device_iter`, so that one part's upload runs under the
        assembly of the next and under the kernels of the one before — inside one call the three would otherwise add up."""
        import torch
        if parts is None:
            parts = min(4, len(files) // 256) if (self.native_host and self._gpu_segment_for(files)) else 1
        if parts > 1:
            n = len(files)
            cut = [n * i // parts for i in range(parts + 1)]
            out: List["torch.Tensor"] = []
            for part in self.decode_device_iter((files[cut[i]:cut[i + 1]] for i in range(parts)), depth=2):
                out += part
            return out
        dev = torch.device("cuda", self.ctx.device)
        results: List[Optional["torch.Tensor"]] = [None] * len(files)
        parsed: Dict[int, ParsedJpeg] = {}
        work: List[Tuple[List[int], Optional[PreparedBatch]]] = []
        rest: List[int] = []
        gpu_segment = self._gpu_segment_for(files)
        if gpu_segment and self.native_host:
            prep = prepare_batch_native(files, self.layout, self.base_flags, staging=self._staging_for(files))
            if isinstance(prep, PreparedBatch):                   # the everyday case: one pass, one plan
                work.append((list(range(len(files))), prep))
            else:
                # files of several kinds (sampling layouts; with / without restart markers), or some the front end does not
                # take (progressive, ...): it sorts them — one native assembly per kind when its turn comes (one staging
                # buffer), the Python path for the rest
                groups_n, rest = prepare_batch_native(files, self.layout, self.base_flags, staging=self._staging_for(files), split=True)
                work += [(idxs, "native") for idxs in groups_n]
        if not work or rest:
            todo = rest if work else range(len(files))          # everything, unless the front end kept some of it
            groups: Dict[tuple, List[int]] = {}
            for i in todo:
                p = parsed[i] = parse_jpeg(files[i], headers_only=gpu_segment)
                check_supported(p)
                comps = list(p.color_components.values())
                key = (p.scan_mode, len(comps), p.headers_only, is_scan_list(p), p.headers_only and p.restart_interval > 0) + (tuple((c.horizontal_sampling, c.vertical_sampling) for c in comps) if len(comps) > 1 else ())
                groups.setdefault(key, []).append(i)
            # (first in line: what the front end left over is mostly progressive files, whose decode is a long serial chain
            # the other plans can run beside)
            work = [(idxs, None) for idxs in groups.values()] + work
        # Several plans (a batch of several kinds of files): all are submitted before the first is collected, on a few
        # streams in turn, so that small plans share the GPU instead of queueing behind each other's host round trips
        # (mj_plan_sync waits for a plan's own work only).  Files handed back by the GPU scan go round again.
        streams = None
        while work:
            flying = []
            try:
                while work:
                    idxs, prep = work.pop(0)
                    if isinstance(prep, str):
                        sub = [files[i] for i in idxs]
                        prep = prepare_batch_native(sub, self.layout, self.base_flags, staging=self._staging_for(sub))
                        if not isinstance(prep, PreparedBatch):   # cannot happen for a group the front end just formed
                            prep = None
                            for i in idxs:
                                parsed[i] = parse_jpeg(files[i], headers_only=True)
                                check_supported(parsed[i])
                    if prep is None or isinstance(prep, int):
                        prep = prepare_batch([files[i] for i in idxs], self.layout, self.base_flags | (prep or 0), [parsed[i] for i in idxs])
                    if work or flying:
                        # more than one plan: keep off the null stream, whose copies would wait for the other plans' kernels
                        if streams is None:
                            streams = [torch.cuda.Stream(device=dev) for _ in range(5)]
                        with torch.cuda.stream(streams[4]):
                            d_blob = torch.from_numpy(prep.blob).to(dev)         # (pageable source: the staging buffer is free on return)
                    else:
                        d_blob = torch.from_numpy(prep.blob).to(dev)
                    plan = B.Plan(self.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(idxs)})
                    flying.append((idxs, prep, plan, None, d_blob))
                    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
                    flying[-1] = (idxs, prep, plan, d_rgb, d_blob)
                    # d_rgb comes from torch's caching allocator on torch's CURRENT stream: a block a consumer has just
                    # dropped may still be read by kernels queued there, so the stream that is about to overwrite it waits
                    # for the current stream first (the allocator only orders reuse within one stream)
                    cur = torch.cuda.current_stream(dev)
                    if streams is None:
                        ev = torch.cuda.Event()
                        ev.record(cur)
                        self.ctx.wait_event(ev.cuda_event)
                        plan.execute(0, d_rgb.data_ptr())                        # the everyday case: one plan, the context's stream
                    else:
                        # (streams only overlap when they sit on different hardware queues: the package asks the runtime for
                        # eight instead of four, see __init__.py)
                        st = streams[(len(flying) - 1) % 4]
                        st.wait_stream(streams[4])                               # the upload above
                        st.wait_stream(cur)
                        d_rgb.record_stream(st)
                        d_blob.record_stream(st)
                        plan.execute(st.cuda_stream, d_rgb.data_ptr())
                for idxs, prep, plan, d_rgb, _ in flying:
                    plan.sync()
                    status = plan.read(rgb=False)["status"]
                    redo = [i for k, i in enumerate(idxs) if status[k] == B.MJ_ST_TAIL]
                    if redo:
                        for i in redo:
                            parsed[i] = parse_jpeg(files[i])
                            check_supported(parsed[i])
                        work.append((redo, None))
                        status[[k for k, i in enumerate(idxs) if i in redo]] = 0
                    again = [i for k, i in enumerate(idxs) if status[k] == B.MJ_ST_UNCONVERGED]
                    if again:                                    # synchronisation rounds not settled: the serial walk for these
                        for i in again:
                            if i not in parsed:
                                parsed[i] = parse_jpeg(files[i])
                                check_supported(parsed[i])
                        work.append((again, B.MJ_FLAG_NO_SYNC))
                        status[[k for k, i in enumerate(idxs) if i in again]] = 0
                    raise_for_status(status)
                    off = 0
                    for k, i in enumerate(idxs):
                        w, h, nc = prep.shapes[k]
                        n = w * h * nc
                        if i not in redo and i not in again:
                            results[i] = d_rgb[off:off + n].view(self._shape(w, h, nc))
                        off += n
            finally:
                for item in flying:
                    item[2].close()
        return results

    def decode_device_iter(self, batches, depth=2):
        """Decode a stream of batches (an iterable of lists of file bytes) with the host work and the upload of the next
        batches overlapping the GPU work of the ones before; yields, per batch and in order, what :meth:`decode_device` returns.

        Per batch: the native host front end assembles the blob in one of ``depth + 1`` pinned buffers (host threads), the upload
        is queued on a copy stream, the plan is created (its buffer clears run on the context's setup stream), its kernels
        are queued on the context's stream behind the previous batch's, waiting for the upload by event — and only then is the
        batch ``depth`` places back collected (``mj_plan_sync`` waits for that plan's own work) and handed out.  With one batch
        in flight (round 5) the host waited for batch k's kernels before it started assembling batch k + 2, and the copy engine
        idled meanwhile: 512 x 1080p took the front end's 4.3 ms PLUS the upload's 6.1 ms per batch; with two the three —
        host threads, copy engine, GPU — run side by side and the batch takes what the slowest of them takes.  Batches the front
        end declines are decoded by :meth:`decode_device` in place, behind everything in flight (no overlap for those)."""
        import collections
        import torch
        dev = torch.device("cuda", self.ctx.device)
        copy_stream = torch.cuda.Stream(device=dev)
        depth = max(1, int(depth))
        pinned = [None] * (depth + 1)
        uploaded = [None] * (depth + 1)     # event behind the latest upload out of each pinned buffer
        turn = 0
        pending = collections.deque()       # ((plan, prep, d_rgb, d_blob), files) of the batches in flight, oldest first

        def finish(job):
            plan, prep, d_rgb, _ = job
            try:
                plan.sync()
                status = plan.read(rgb=False)["status"]
                again = np.flatnonzero((status == B.MJ_ST_TAIL) | (status == B.MJ_ST_UNCONVERGED))
                status = status.copy()
                status[again] = 0                                 # something behind a scan / rounds not settled: those files again, below
                raise_for_status(status)
                out, off = [], 0
                for (w, h, nc) in prep.shapes:
                    n = w * h * nc
                    out.append(d_rgb[off:off + n].view(self._shape(w, h, nc)))
                    off += n
                return out, again
            finally:
                plan.close()

        def collect(job):
            out, again = finish(job[0])
            if again.size:                                        # only the files concerned take the long way (host parse)
                redo = self.decode_device([job[1][int(i)] for i in again], parts=1)
                for i, img in zip(again, redo):
                    out[int(i)] = img
            return out

        try:
            for files in batches:
                files = list(files)
                prep = None
                if self.gpu_segment and self.native_host and files:
                    buf, turn = turn, (turn + 1) % (depth + 1)
                    need = sum(map(len, files)) + 3 * len(files) + 1024
                    if uploaded[buf] is not None:
                        uploaded[buf].synchronize()               # depth + 1 batches ago: long done
                    if pinned[buf] is None or pinned[buf].numel() < need:
                        pinned[buf] = torch.empty(need + need // 4, dtype=torch.uint8, pin_memory=True)
                    prep = prepare_batch_native(files, self.layout, self.base_flags, staging=pinned[buf].numpy())
                if not isinstance(prep, PreparedBatch):           # declined, or several plans' worth: the one-call path sorts it out
                    prep = None
                if prep is None:
                    while pending:
                        yield collect(pending.popleft())
                    yield self.decode_device(files, parts=1)
                    continue
                with torch.cuda.stream(copy_stream):
                    d_blob = pinned[buf][:prep.blob.size].to(dev, non_blocking=True)
                    uploaded[buf] = torch.cuda.Event()
                    uploaded[buf].record(copy_stream)
                plan = B.Plan(self.ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": len(files)})
                try:
                    # (both tensors outlive the kernels that touch them: they stay in `pending` until the plan has been collected)
                    d_rgb = torch.empty(plan.info.rgb_bytes, dtype=torch.uint8, device=dev)
                    self.ctx.wait_event(uploaded[buf].cuda_event)
                    ev = torch.cuda.Event()                        # see decode_device: the context's stream waits for whatever
                    ev.record(torch.cuda.current_stream(dev))      # the current stream still does with a recycled block
                    self.ctx.wait_event(ev.cuda_event)
                    plan.execute(0, d_rgb.data_ptr())
                except BaseException:
                    plan.close()
                    raise
                pending.append(((plan, prep, d_rgb, d_blob), files))
                while len(pending) > depth:
                    yield collect(pending.popleft())
            while pending:
                yield collect(pending.popleft())
        finally:
            while pending:                                        # (an error, or a consumer that stopped early: nothing stays open)
                pending.popleft()[0][0].close()

    def close(self):
        self.ctx.close()
