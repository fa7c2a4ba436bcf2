"""``JpegDecoder`` — the reference's class surface (jpeg_decoder.py:27-110) over the MI355X path.

Constructing the object *is* the decode, exactly as in the reference: afterwards the same attributes are
set and ``image_array`` holds the ``uint8 (W, H, 3)`` (or ``(W, H)``) x-major result.
What differs, deliberately:

* the per-MCU hot path — ``baseline_dct_scan`` (:697-906), ``InverseDCT`` (:1535-1573), ``ResizeGrid``
  (:1580-1626), ``YCbCr_to_RGB`` (:1683-1700) — runs in libmijpeg.so's HIP kernels, not in Python;
* console chatter is off unless ``verbose=True`` and no viewer opens unless ``show=True``
  (the reference prints per MCU and calls ``self.show()`` at EOI, :1389);
* malformed restart data raises ``CorruptedJpeg`` instead of decoding garbage.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

import numpy as np

from . import _binding as B
from ._parse import (ColorComponent, HuffmanTable, ParsedJpeg, parse_jpeg, DHT, DQT, DRI, SOF0, SOF2, SOS, EOI, RST, bytes_to_uint,
                     parse_huffman_segment, parse_quantization_segment, _start_of_frame, _start_of_scan)
from .batch import check_supported, prepare_batch, raise_for_status
from .errors import CorruptedJpeg, JpegError, NotJpeg, UnsupportedJpeg  # noqa: F401  (re-exported like the reference)

_contexts = {}


def _context(device: int) -> B.Context:
    if device not in _contexts:
        _contexts[device] = B.Context(device)
    return _contexts[device]


class JpegDecoder():

    def __init__(self, file: Path, *, device: int = 0, verbose: bool = False, show: bool = False) -> None:
        # Open file (:31-35)
        with open(file, "rb") as image:
            self.raw_file = image.read()
        self.file_size = len(self.raw_file)
        self.file_path = file if isinstance(file, Path) else Path(file)
        self._verbose = verbose

        # (:39-41)  note: like the reference, a str path fails on `.name` — Path is the contract
        if not self.raw_file.startswith(b"\xFF\xD8\xFF"):
            raise NotJpeg("File is not a JPEG image.")
        self._say(f"Reading file '{file.name}' ({self.file_size:,} bytes)")

        self.handlers = {
            DHT: self.define_huffman_table, DQT: self.define_quantization_table, DRI: self.define_restart_interval,
            SOF0: self.start_of_frame, SOF2: self.start_of_frame, SOS: self.start_of_scan, EOI: self.end_of_image,
        }

        # Initialize decoding parameters (:55-66)
        self.file_header = 2
        self.scan_finished = False
        self.scan_mode = None
        self.image_width = 0
        self.image_height = 0
        self.color_components = {}
        self.sample_shape = ()
        self.huffman_tables = {}
        self.quantization_tables = {}
        self.restart_interval = 0
        self.image_array = None
        self.scan_count = 0

        # Host side: the marker loop (:78-110).  The handlers are the reference's six, with the reference's contract
        # (called with the segment's payload, `file_header` behind the length field); what differs is that
        # `start_of_scan` records the scan instead of decoding it in place and `end_of_image` sends the recorded
        # scans through the GPU path in one go.
        self._parsed = ParsedJpeg(raw=self.raw_file, file_size=self.file_size)
        self._arr = np.frombuffer(self.raw_file, dtype=np.uint8)
        self._device = device
        self._show = show
        n = self.file_size
        while not self.scan_finished:
            if self.file_header >= n:                     # the reference's IndexError branch (:81-83)
                break
            if self.raw_file[self.file_header] == 0xFF:
                my_marker = self.raw_file[self.file_header:self.file_header + 2]
                self.file_header += 2
                if my_marker != b"\xFF\x00" and my_marker not in RST:
                    my_handler = self.handlers.get(my_marker)
                    my_size = bytes_to_uint(self.raw_file[self.file_header:self.file_header + 2]) - 2
                    self.file_header += 2
                    if my_handler is not None:
                        my_handler(self.raw_file[self.file_header:self.file_header + my_size] if my_size > 0 else b"")
                    else:
                        self.file_header += my_size       # unknown segment: skipped (:104-106)
            else:
                self.file_header += 1
        if not self.scan_finished:
            # The file ended without EOI.  The reference falls off the end here (:81-83) leaving image_array as the
            # unconverted int16 array (or None when no scan was seen); the GPU path converts what it decoded.
            if self._parsed.scans:
                self._decode_scans()
            del self.raw_file
            del self._arr

    # `huffman_tables` (:61): {destination: {codeword string: value}} as the reference holds it.  Nothing on the decode path reads
    # the strings (the GPU tables are built from BITS / HUFFVAL), and writing them out costs more than the whole decode of a
    # 1080p file (0.5 of 2.2 ms), so a DHT only notes its tables and the dict is filled in — in place — when it is looked at.
    @property
    def huffman_tables(self):
        pending = self.__dict__.get("_pending_tables")
        if pending:
            for dest, spec in pending.items():
                self._huffman_tables[dest] = dict(spec.tree)
            pending.clear()
        return self._huffman_tables

    @huffman_tables.setter
    def huffman_tables(self, value):
        self._huffman_tables = value
        self._pending_tables = {}

    # -- the reference's handlers (:112-652, :1368-1390): same names, same argument, same effect on the attributes ------
    def start_of_frame(self, data: bytes) -> None:
        """SOF0 / SOF2 (:112-247): scan mode, dimensions, components, sampling, quantisation table ids."""
        p = self._parsed
        _start_of_frame(p, self.raw_file[self.file_header - 4:self.file_header - 2], data, self._say)
        self.file_header += len(data)
        self.scan_mode = p.scan_mode
        self.image_width, self.image_height = p.image_width, p.image_height
        self.color_components = dict(p.color_components)
        self.sample_shape = p.sample_shape

    def define_huffman_table(self, data: bytes) -> None:
        """DHT (:249-390): the tables as the reference's {codeword string: value} dicts; BITS/HUFFVAL kept for the GPU LUTs."""
        for dest, spec in parse_huffman_segment(data).items():
            self._parsed.huffman[dest] = spec
            self._pending_tables[dest] = spec              # the codeword strings are built when somebody looks (`huffman_tables`)
            self._say(f"Parsed Huffman table - ID: {dest & 0x0F} ({'DC' if dest >> 4 == 0 else 'AC'})")
        self.file_header += len(data)

    def define_quantization_table(self, data: bytes) -> None:
        """DQT (:392-472): int16 [x, y] tables (undo_zigzag of the 64 bytes)."""
        for dest, (zz, xy) in parse_quantization_segment(data).items():
            self._parsed.quantization_zz[dest] = zz
            self._parsed.quantization_tables[dest] = xy
            self.quantization_tables[dest] = xy
            self._say(f"Parsed quantization table - ID: {dest}")
        self.file_header += len(data)

    def define_restart_interval(self, data: bytes) -> None:
        """DRI (:474-503); like the reference the header moves on by 2, not by the segment's size."""
        self.restart_interval = self._parsed.restart_interval = bytes_to_uint(data[:2])
        self.file_header += 2
        self._say(f"Restart interval: {self.restart_interval}")

    def start_of_scan(self, data: bytes) -> None:
        """SOS (:505-652): table selectors, spectral selection, MCU geometry, restart segmentation.  The scan is
        recorded; `file_header` moves behind its entropy-coded bytes, where the reference's scan decoder leaves it."""
        p = self._parsed
        scan = _start_of_scan(p, data, self.file_header, self._arr, self._say)
        p.scans.append(scan)
        self.image_height = p.image_height            # a DNL segment may have supplied it (:575-581)
        self.scan_amount = p.scan_amount
        self.mcu_width, self.mcu_height = scan.mcu_width, scan.mcu_height
        self.mcu_shape = (scan.mcu_width, scan.mcu_height)
        self.mcu_count_h, self.mcu_count_v, self.mcu_count = scan.mcu_count_h, scan.mcu_count_v, scan.mcu_count
        self.array_width, self.array_height, self.array_depth = p.array_width, p.array_height, p.array_depth
        self._say(f"\nScan {len(p.scans)} of {self.scan_amount}")
        self._say(f"Color components: {', '.join(p.color_components[c].name for c in scan.component_ids)}")
        self._say(f"MCU count: {scan.mcu_count}")
        self.file_header = scan.entropy_end

    def end_of_image(self, data: bytes) -> None:
        """EOI (:1368-1390): decode the recorded scans on the GPU (Huffman -> dequantise -> IDCT -> upsample -> crop ->
        colour), set `image_array`, finish."""
        self._parsed.reached_eoi = True
        if self._parsed.scans:
            self._decode_scans()
        self.scan_finished = True
        if self._show and self.image_array is not None:
            self.show()
        del self.raw_file
        del self._arr

    def _decode_scans(self) -> None:
        parsed = self._parsed
        parsed.file_header = self.file_header
        check_supported(parsed)
        self._say("Decoding MCUs and performing IDCT on the GPU...")
        ctx = _context(self._device)
        for flags in (0, B.MJ_FLAG_NO_SYNC):
            prep = prepare_batch([self.raw_file], B.MJ_LAYOUT_XMAJOR, flags, [parsed])
            plan = B.Plan(ctx, prep.to_c(), {"prep": prep, "n_images": 1})
            try:
                plan.execute()
                plan.sync()
                out = plan.read(rgb=True)
            finally:
                plan.close()
            if out["status"][0] != B.MJ_ST_UNCONVERGED:      # (else: once more, one serial walk per segment)
                break
        raise_for_status(out["status"])
        self.scan_count = len(parsed.scans)
        shape = (self.image_width, self.image_height) + ((3,) if self.array_depth == 3 else ())
        self.image_array = out["rgb"].reshape(shape)

    def _say(self, text: str) -> None:
        if self._verbose:
            print(text)

    def show(self):
        """Display the decoded image (:1392-1443) — PIL viewer only; the Tk window is out of scope."""
        from PIL import Image
        img = np.swapaxes(self.image_array, 0, 1)
        Image.fromarray(img).show()

    def save(self, path=None):
        """Lossless save of the decoded image — the reference's save dialog (:1490-1532) minus the dialog: `path` stands for
        what the user would have picked (default: the source file's folder and stem, ".png").  As there, an existing file is
        never overwritten (" (1)", " (2)", ... is put behind the stem) and a suffix Pillow cannot write falls back to PNG.
        Returns the path written."""
        from pathlib import Path
        from PIL import Image
        target = Path(path) if path is not None else self.file_path.with_suffix(".png")
        picture = Image.fromarray(np.swapaxes(self.image_array, 0, 1))

        def free_name(q: Path) -> Path:
            stem, n = q.stem, 0
            while q.exists():
                n += 1
                q = q.with_stem(f"{stem} ({n})")
            return q

        target = free_name(target)
        try:
            picture.save(target)
        except ValueError:                          # no writer for that suffix
            target = free_name(target.with_suffix(".png"))
            picture.save(target, format="png")
        self._say(f"Decoded image was saved to '{target}'")
        return target
