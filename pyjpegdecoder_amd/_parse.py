"""Host-side container parsing: marker loop, SOF/DHT/DQT/DRI/SOS headers, scan segmentation.

This is the Python host code that sits ABOVE the drop-in boundary: it prepares exactly
what the reference's ``start_of_scan`` (jpeg_decoder.py:505-650) has prepared when it calls
``baseline_dct_scan`` — table selectors, Huffman BITS/HUFFVAL, quantisation tables, restart interval,
MCU geometry, the offset of the first entropy-coded byte — plus the restart-segment offsets the
MI355X stage-1 kernel needs (one wavefront per restart segment).

Semantics follow the reference's marker loop (jpeg_decoder.py:78-110) and handlers (:112-503); the
citations next to each function say which lines.  Nothing here touches the GPU.
"""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass, field
from math import ceil
from typing import Dict, List, Optional, Tuple

import numpy as np

from .errors import CorruptedJpeg, NotJpeg, UnsupportedJpeg

# Marker bytes (jpeg_decoder.py:9-21)
SOI, SOF0, SOF2, DHT, DQT, DRI, SOS, DNL, EOI = (
    b"\xFF\xD8", b"\xFF\xC0", b"\xFF\xC2", b"\xFF\xC4", b"\xFF\xDB", b"\xFF\xDD", b"\xFF\xDA", b"\xFF\xDC", b"\xFF\xD9")
RST = tuple(bytes((0xFF, m)) for m in range(0xD0, 0xD8))

# Same field names/order as the reference's namedtuples (jpeg_decoder.py:24-25)
ColorComponent = namedtuple(
    "ColorComponent", "name order vertical_sampling horizontal_sampling quantization_table_id repeat shape")
HuffmanTable = namedtuple("HuffmanTable", "dc ac")

# Zig-zag index grid, rows = vertical frequency (jpeg_decoder.py:430-437, :1652-1660)
ZZ_GRID = np.array([
    [0, 1, 5, 6, 14, 15, 27, 28],
    [2, 4, 7, 13, 16, 26, 29, 42],
    [3, 8, 12, 17, 25, 30, 41, 43],
    [9, 11, 18, 24, 31, 40, 44, 53],
    [10, 19, 23, 32, 39, 45, 52, 54],
    [20, 22, 33, 38, 46, 51, 55, 60],
    [21, 34, 37, 47, 50, 56, 59, 61],
    [35, 36, 48, 49, 57, 58, 62, 63]], dtype=np.int64)


def undo_zigzag(block: np.ndarray) -> np.ndarray:
    """64 zig-zag values -> 8x8 ``[x, y]`` array (jpeg_decoder.py:1648-1662; note the transpose)."""
    return np.asarray(block)[ZZ_GRID].T.copy()


def bytes_to_uint(b: bytes) -> int:
    return int.from_bytes(b, byteorder="big", signed=False)


@dataclass
class HuffSpec:
    """One DHT table as BITS[16] + HUFFVAL (what the GPU LUT builder consumes) and as the reference's
    ``{codeword-string: value}`` dict (jpeg_decoder.py:366-377)."""
    bits: np.ndarray          # uint8[16]
    vals: np.ndarray          # uint8[n]
    _tree: Optional[Dict[str, int]] = None

    @property
    def tree(self) -> Dict[str, int]:
        """The reference's ``{zero-padded binary codeword: value}`` dict, built on first use (only the
        ``JpegDecoder`` class surface shows it; batch decoding never needs the strings)."""
        if self._tree is None:
            tree: Dict[str, int] = {}
            code, k = 0, 0
            for bit_length, count in enumerate(self.bits.tolist(), start=1):
                code <<= 1
                for _ in range(count):
                    if k < self.vals.size:
                        tree[bin(code)[2:].rjust(bit_length, "0")] = int(self.vals[k])
                    k += 1
                    code += 1
            self._tree = tree
        return self._tree


@dataclass
class ScanInfo:
    """Everything ``start_of_scan`` hands to the scan decoder (jpeg_decoder.py:529-650)."""
    component_ids: List[int]
    huffman_tables_id: Dict[int, HuffmanTable]     # component id -> (dc dest, ac dest); ac dest has 0x10 set
    spectral_start: int = 0
    spectral_end: int = 63
    bit_high: int = 0
    bit_low: int = 0
    entropy_start: int = 0        # file offset of the first entropy-coded byte (:572)
    entropy_end: int = 0          # file offset of the marker that terminates the entropy-coded data
    restart_interval: int = 0     # value in force for this scan (:501-502)
    mcu_width: int = 8
    mcu_height: int = 8
    mcu_count_h: int = 0
    mcu_count_v: int = 0
    huffman: Dict[int, HuffSpec] = field(default_factory=dict)   # snapshot of the tables in force
    quantization_zz: Dict[int, np.ndarray] = field(default_factory=dict)   # ... and of the quantisation tables (zig-zag)
    segment_offsets: Optional[np.ndarray] = None   # int64[n_seg+1]: file offsets of restart-segment starts, then entropy_end

    @property
    def mcu_count(self) -> int:
        return self.mcu_count_h * self.mcu_count_v


def parse_huffman_segment(data: bytes) -> Dict[int, HuffSpec]:
    """DHT payload -> tables (jpeg_decoder.py:293-377)."""
    out: Dict[int, HuffSpec] = {}
    size, pos = len(data), 0
    while pos < size:
        dest = data[pos]
        pos += 1
        bits = np.zeros(16, dtype=np.uint8)
        raw = data[pos:pos + 16]
        bits[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        pos += 16
        vals_by_len = []
        for count in bits.tolist():
            vals_by_len.append(data[pos:pos + count])
            pos += count
        if pos > size:
            raise CorruptedJpeg("Failed to parse Huffman tables.")
        # more than 256 symbols, or more codes of some length than that length has left (the reference would build a
        # dict with colliding / overlong keys and fail later on an unknown code, :718-719): corrupt either way, and the
        # device-side canonical search indexes vals[] by these counts
        kraft = 0                                 # codes used so far, in units of 2^-16
        for length, count in enumerate(bits.tolist(), start=1):
            kraft += count << (16 - length)
        if int(bits.sum()) > 256 or kraft > (1 << 16):
            raise CorruptedJpeg("Failed to parse Huffman tables.")
        vals = np.frombuffer(b"".join(vals_by_len), dtype=np.uint8).copy()
        out[dest] = HuffSpec(bits=bits, vals=vals)
    return out


def parse_quantization_segment(data: bytes) -> Dict[int, Tuple[np.ndarray, np.ndarray]]:
    """DQT payload -> {dest: (zig-zag uint8-valued int16[64], reference-layout int16[8,8])} (:442-462)."""
    out = {}
    size, pos = len(data), 0
    while pos < size:
        dest = data[pos]
        pos += 1
        raw = data[pos:pos + 64]
        if len(raw) != 64:
            # undo_zigzag on a short array raises IndexError in the reference (uncaught there); treat as corrupt
            raise CorruptedJpeg("Failed to parse quantization tables.")
        zz = np.frombuffer(raw, dtype=np.uint8).astype(np.int16)
        out[dest] = (zz, undo_zigzag(zz))
        pos += 64
    return out


def find_entropy_end(raw: np.ndarray, start: int) -> int:
    """Offset of the first marker after ``start`` that is neither a stuffed ``FF 00`` nor ``RSTn``.

    For well-formed files this is where the reference's ``file_header`` rests after the scan: its bit
    reader (:654-695) stops on the byte boundary after the last MCU and the marker loop (:78-110) then
    finds the next marker there.
    """
    ff = np.flatnonzero(raw[start:-1] == 0xFF) + start
    if ff.size:
        nxt = raw[ff + 1]
        is_marker = (nxt != 0x00) & ((nxt < 0xD0) | (nxt > 0xD7)) & (nxt != 0xFF)
        hits = ff[is_marker]
        if hits.size:
            return int(hits[0])
    return int(raw.size)


def find_restart_segments(raw: np.ndarray, start: int, end: int) -> np.ndarray:
    """int64 offsets of the first byte of every restart segment in ``[start, end)``, then ``end``.

    The reference never looks at the RSTn bytes: after ``restart_interval`` MCUs it drops its queued bits
    and skips two bytes (:667-669, :898-900).  In a well-formed stream those two bytes are the ``FF Dn``
    found here, so segment k starts two bytes after the k-th RSTn.  The stage-1 kernel reports a
    per-segment desync if the bits it consumes do not end where the next marker starts.
    """
    view = raw[start:end]
    ff = np.flatnonzero(view[:-1] == 0xFF) if view.size > 1 else np.zeros(0, dtype=np.int64)
    if ff.size:
        nxt = view[ff + 1]
        ff = ff[(nxt >= 0xD0) & (nxt <= 0xD7)]
    out = np.empty(ff.size + 2, dtype=np.int64)
    out[0] = start
    out[1:-1] = ff + start + 2
    out[-1] = end
    return out


@dataclass
class ParsedJpeg:
    """State the reference keeps on ``self`` while walking the file (jpeg_decoder.py:54-66)."""
    raw: bytes
    file_size: int
    scan_mode: Optional[str] = None
    image_width: int = 0
    image_height: int = 0
    color_components: Dict[int, ColorComponent] = field(default_factory=dict)
    sample_shape: Tuple[int, int] = ()
    huffman: Dict[int, HuffSpec] = field(default_factory=dict)
    quantization_zz: Dict[int, np.ndarray] = field(default_factory=dict)
    quantization_tables: Dict[int, np.ndarray] = field(default_factory=dict)
    restart_interval: int = 0
    scans: List[ScanInfo] = field(default_factory=list)
    scan_amount: int = 0
    array_width: int = 0
    array_height: int = 0
    array_depth: int = 0
    file_header: int = 2
    reached_eoi: bool = False
    headers_only: bool = False       # parse stopped at the SOS (see parse_jpeg)
    log: List[str] = field(default_factory=list)


def parse_jpeg(raw: bytes, headers_only: bool = False) -> ParsedJpeg:
    """Walk the file like the reference's constructor does (jpeg_decoder.py:29-110), but instead of
    decoding each scan in place, record a :class:`ScanInfo` and jump to the marker that ends it.

    ``headers_only``: stop at the SOS of a baseline frame whose scan holds every component, without
    looking at the entropy-coded bytes at all — the GPU finds the restart markers and the end of the
    scan (MJ_FLAG_GPU_SEGMENT).  The scan then carries ``entropy_end = len(raw)`` as a bound and no
    ``segment_offsets``; progressive and multi-scan files are parsed in full regardless."""
    if not raw.startswith(SOI + b"\xFF"):
        raise NotJpeg("File is not a JPEG image.")
    p = ParsedJpeg(raw=raw, file_size=len(raw))
    arr = np.frombuffer(raw, dtype=np.uint8)
    say = p.log.append
    pos = 2
    n = len(raw)

    while not p.reached_eoi:
        if pos >= n:                       # IndexError branch (:81-83)
            break
        if raw[pos] != 0xFF:
            pos += 1
            continue
        marker = raw[pos:pos + 2]
        pos += 2
        if marker == b"\xFF\x00" or marker in RST:
            continue
        size = bytes_to_uint(raw[pos:pos + 2]) - 2
        pos += 2

        if marker == EOI:                  # end_of_image (:1368)
            p.reached_eoi = True
            break
        data = raw[pos:pos + size] if size > 0 else b""

        if marker in (SOF0, SOF2):         # start_of_frame (:112-247)
            _start_of_frame(p, marker, data, say)
            pos += len(data)
        elif marker == DHT:                # define_huffman_table (:249-390)
            for dest, spec in parse_huffman_segment(data).items():
                p.huffman[dest] = spec
                say(f"Parsed Huffman table - ID: {dest & 0x0F} ({'DC' if dest >> 4 == 0 else 'AC'})")
            pos += len(data)
        elif marker == DQT:                # define_quantization_table (:392-472)
            for dest, (zz, xy) in parse_quantization_segment(data).items():
                p.quantization_zz[dest] = zz
                p.quantization_tables[dest] = xy
                say(f"Parsed quantization table - ID: {dest}")
            pos += len(data)
        elif marker == DRI:                # define_restart_interval (:474-478) — advances by 2, not by size
            p.restart_interval = bytes_to_uint(data[:2])
            say(f"Restart interval: {p.restart_interval}")
            pos += 2
        elif marker == SOS:                # start_of_scan (:505-650)
            scan = _start_of_scan(p, data, pos, arr, say, headers_only)
            p.scans.append(scan)
            if scan.segment_offsets is None:       # headers_only: the rest of the file is the GPU's to look at
                p.headers_only = True
                pos = scan.entropy_start
                break
            pos = scan.entropy_end
        else:
            pos += size                    # unknown segment: skipped (:104-106)
    p.file_header = pos
    return p


def _start_of_frame(p: ParsedJpeg, marker: bytes, data: bytes, say) -> None:
    if marker == SOF0:
        p.scan_mode = "baseline_dct"
        say("Scan mode: Sequential")
    else:
        p.scan_mode = "progressive_dct"
        say("Scan mode: Progressive")
    try:
        precision = data[0]
    except IndexError:
        raise CorruptedJpeg("Failed to parse the start of frame.")
    if precision != 8:
        raise UnsupportedJpeg("Unsupported color depth. Only 8-bit greyscale and 24-bit RGB are supported.")
    p.image_height = bytes_to_uint(data[1:3])
    p.image_width = bytes_to_uint(data[3:5])
    say(f"Image dimensions: {p.image_width} x {p.image_height}")
    if p.image_width == 0:
        raise CorruptedJpeg("Image width cannot be zero.")
    try:
        components_amount = data[5]
    except IndexError:
        raise CorruptedJpeg("Failed to parse the start of frame.")
    if components_amount not in (1, 3):
        if components_amount == 4:
            raise UnsupportedJpeg("CMYK color space is not supported. Only RGB and greyscale are supported.")
        raise UnsupportedJpeg("Unsupported color space. Only RGB and greyscale are supported.")
    say("Color space: YCbCr" if components_amount == 3 else "Color space: greyscale")
    h = 6
    try:
        for count, name in enumerate(("Y", "Cb", "Cr"), start=1):
            my_id = data[h]
            sample = data[h + 1]
            hs, vs = sample >> 4, sample & 0x0F
            if hs == 0 or vs == 0:
                raise CorruptedJpeg("Failed to parse the start of frame.")     # the reference divides by these (:596-619)
            qt = data[h + 2]
            h += 3
            p.color_components[my_id] = ColorComponent(
                name=name, order=count - 1, horizontal_sampling=hs, vertical_sampling=vs,
                quantization_table_id=qt, repeat=hs * vs, shape=(8 * hs, 8 * vs))
            if count == components_amount:
                break
    except IndexError:
        raise CorruptedJpeg("Failed to parse the start of frame.")
    p.sample_shape = (max(c.shape[0] for c in p.color_components.values()),
                      max(c.shape[1] for c in p.color_components.values()))
    say("Horizontal sampling: " + " x ".join(str(c.horizontal_sampling) for c in p.color_components.values()))
    say("Vertical sampling  : " + " x ".join(str(c.vertical_sampling) for c in p.color_components.values()))


def _start_of_scan(p: ParsedJpeg, data: bytes, data_pos: int, arr: np.ndarray, say, headers_only: bool = False) -> ScanInfo:
    if p.scan_mode is None:
        raise CorruptedJpeg("Start of scan before start of frame.")
    h = 0
    ids: List[int] = []
    tabs: Dict[int, HuffmanTable] = {}
    try:                                    # a truncated header is an IndexError in the reference too; here it has a name
        components_amount = data[h]
        h += 1
        for _ in range(components_amount):
            cid = data[h]
            t = data[h + 1]
            h += 2
            if cid not in p.color_components:
                raise CorruptedJpeg("Scan refers to a color component that the frame does not define.")
            ids.append(cid)
            tabs[cid] = HuffmanTable(dc=t >> 4, ac=(t & 0x0F) | 0x10)       # :543-544
        scan = ScanInfo(component_ids=ids, huffman_tables_id=tabs)
        if p.scan_mode == "progressive_dct":
            scan.spectral_start, scan.spectral_end = data[h], data[h + 1]
            scan.bit_high, scan.bit_low = data[h + 2] >> 4, data[h + 2] & 0x0F
    except IndexError:
        raise CorruptedJpeg("Failed to parse the start of scan.")
    if not ids:
        raise CorruptedJpeg("Failed to parse the start of scan.")
    scan.entropy_start = data_pos + len(data)                             # :572

    if p.image_height == 0:                                               # DNL (:575-581)
        idx = p.raw[scan.entropy_start:].find(DNL)
        if idx == -1:
            raise CorruptedJpeg("Image height cannot be zero.")
        idx += scan.entropy_start
        p.image_height = bytes_to_uint(p.raw[idx + 4: idx + 6])

    comps = p.color_components
    if components_amount > 1:                                             # :591-611
        scan.mcu_width = 8 * max(c.horizontal_sampling for c in comps.values())
        scan.mcu_height = 8 * max(c.vertical_sampling for c in comps.values())
        scan.mcu_count_h = (p.image_width // scan.mcu_width) + (0 if p.image_width % scan.mcu_width == 0 else 1)
        scan.mcu_count_v = (p.image_height // scan.mcu_height) + (0 if p.image_height % scan.mcu_height == 0 else 1)
    else:                                                                 # :612-619
        comp = comps[ids[-1]]
        ratio_h = p.sample_shape[0] / comp.shape[0]
        ratio_v = p.sample_shape[1] / comp.shape[1]
        scan.mcu_width = scan.mcu_height = 8
        scan.mcu_count_h = ceil((p.image_width / ratio_h) / 8)
        scan.mcu_count_v = ceil((p.image_height / ratio_v) / 8)

    if not p.scans:                                                       # :624-637
        sw, sh = p.sample_shape
        count_h = (p.image_width // sw) + (0 if p.image_width % sw == 0 else 1)
        count_v = (p.image_height // sh) + (0 if p.image_height % sh == 0 else 1)
        p.array_width, p.array_height, p.array_depth = sw * count_h, sh * count_v, len(comps)
        gpu_scan = headers_only and p.scan_mode == "baseline_dct" and components_amount == len(comps) and p.image_height > 0
        first_scan = not gpu_scan
    else:
        gpu_scan = first_scan = False

    scan.restart_interval = p.restart_interval
    scan.huffman = dict(p.huffman)
    scan.quantization_zz = dict(p.quantization_zz)
    if gpu_scan:
        p.scan_amount = 1
        scan.entropy_end = len(p.raw)
        scan.segment_offsets = None
        return scan
    # One pass over the 0xFF bytes behind the scan header gives the number of scans (the first scan announces it, :624-637: the
    # SOS markers from here on), the end of the entropy-coded data (find_entropy_end) and the restart segments
    # (find_restart_segments) — three passes and a copy of the file's tail before round 5, a fifth of JpegDecoder(path)'s time.
    start = scan.entropy_start
    ff = np.flatnonzero(arr[start:-1] == 0xFF) + start
    nxt = arr[ff + 1] if ff.size else ff
    if first_scan:
        p.scan_amount = int(np.count_nonzero(nxt == 0xDA)) + 1
        say(f"Number of scans: {p.scan_amount}")
    end = int(arr.size)
    if ff.size:
        hits = ff[(nxt != 0x00) & ((nxt < 0xD0) | (nxt > 0xD7)) & (nxt != 0xFF)]
        if hits.size:
            end = int(hits[0])
    scan.entropy_end = end
    rst = ff[(nxt >= 0xD0) & (nxt <= 0xD7) & (ff < end - 1)] if ff.size else ff
    offs = np.empty(rst.size + 2, dtype=np.int64)
    offs[0] = start
    offs[1:-1] = rst + 2
    offs[-1] = end
    scan.segment_offsets = offs
    return scan
