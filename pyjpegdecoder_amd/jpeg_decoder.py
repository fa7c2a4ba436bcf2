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
from ._parse import (ColorComponent, HuffmanTable, ParsedJpeg, parse_jpeg, DHT, DQT, DRI, SOF0, SOF2, SOS, EOI)
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

        # Host side: the marker loop (:78-110) — headers, tables, scan geometry, restart segmentation
        parsed = parse_jpeg(self.raw_file)
        self._parsed = parsed
        for line in parsed.log:
            self._say(line)
        self.scan_mode = parsed.scan_mode
        self.image_width, self.image_height = parsed.image_width, parsed.image_height
        self.color_components = dict(parsed.color_components)
        self.sample_shape = parsed.sample_shape
        self.huffman_tables = {dest: dict(spec.tree) for dest, spec in parsed.huffman.items()}
        self.quantization_tables = dict(parsed.quantization_tables)
        self.restart_interval = parsed.restart_interval
        if not parsed.scans:
            # the reference falls off the end of the file here (:81-83) leaving image_array = None
            del self.raw_file
            return
        scan = check_supported(parsed)
        self.scan_amount = parsed.scan_amount
        scan = parsed.scans[-1]          # the geometry attributes are those of the last scan decoded (:591-621)
        self.mcu_width, self.mcu_height = scan.mcu_width, scan.mcu_height
        self.mcu_shape = (scan.mcu_width, scan.mcu_height)
        self.mcu_count_h, self.mcu_count_v, self.mcu_count = scan.mcu_count_h, scan.mcu_count_v, scan.mcu_count
        self.array_width, self.array_height, self.array_depth = parsed.array_width, parsed.array_height, parsed.array_depth

        # Device side: Huffman decode -> dequantise -> IDCT -> upsample -> crop -> colour  (the hot path)
        for k, sc in enumerate(parsed.scans, start=1):
            self._say(f"\nScan {k} of {self.scan_amount}")
            self._say(f"Color components: {', '.join(parsed.color_components[c].name for c in sc.component_ids)}")
            self._say(f"MCU count: {sc.mcu_count}")
        self._say("Decoding MCUs and performing IDCT on the GPU...")
        ctx = _context(device)
        prep = prepare_batch([self.raw_file], B.MJ_LAYOUT_XMAJOR, 0, [parsed])
        plan = B.Plan(ctx, prep.to_c(), {"prep": prep, "n_images": 1})
        try:
            plan.execute()
            plan.sync()
            out = plan.read(rgb=True)
        finally:
            plan.close()
        raise_for_status(out["status"])
        self.scan_count = len(parsed.scans)
        self.file_header = parsed.file_header

        if not parsed.reached_eoi:
            # The reference only crops / colour-converts in end_of_image (:1368-1390); a file without EOI
            # leaves an unconverted int16 array there.  The GPU path has already converted; keep that.
            pass
        shape = (self.image_width, self.image_height) + ((3,) if self.array_depth == 3 else ())
        self.image_array = out["rgb"].reshape(shape)
        self.scan_finished = parsed.reached_eoi
        # parsed.file_header already includes the EOI marker and the bogus 2-byte length read (:89-98)
        if show:
            self.show()
        del self.raw_file

    # -- the reference's handler names, kept so that `handlers` has the same keys/shape ----------------
    def start_of_frame(self, data: bytes) -> None:          # (:112) parsed in _parse._start_of_frame
        raise NotImplementedError("handled by pyjpegdecoder_amd._parse.parse_jpeg")

    define_huffman_table = define_quantization_table = define_restart_interval = start_of_frame
    start_of_scan = end_of_image = start_of_frame

    def _say(self, text: str) -> None:
        if self._verbose:
            print(text)

    def show(self):
        """Display the decoded image (:1392-1443) — PIL viewer only; the Tk window is out of scope."""
        from PIL import Image
        img = np.swapaxes(self.image_array, 0, 1)
        Image.fromarray(img).show()

    def save(self, path) -> None:
        """Lossless save of the decoded image (the reference's save dialog, :1490-1532, minus the GUI)."""
        from PIL import Image
        Image.fromarray(np.swapaxes(self.image_array, 0, 1)).save(path)
