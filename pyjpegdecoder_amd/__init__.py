"""MI355X-native batched JPEG block-decode behind the reference's ``JpegDecoder`` class surface.

    from pyjpegdecoder_amd import JpegDecoder          # drop-in for the reference class (one file)
    from pyjpegdecoder_amd import BatchDecoder         # many files per launch on one GPU

The pixel path lives in ``libmijpeg.so`` (hand-written HIP for gfx950, C ABI in ``include/mijpeg.h``);
importing this package does not need a GPU, decoding does — there is no CPU fallback.
"""
import os as _os

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on one
# queue run one after the other.  A decoder uses several at once (context, plan set-up, uploads, one per concurrent plan
# of a mixed batch): with four queues a progressive plan and a baseline plan "on different streams" were measured to
# serialise whenever their streams happened to share a queue.  Eight are asked for unless the caller chose otherwise;
# this only takes effect when the package is imported before the process first touches the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .errors import BackendError, CorruptedJpeg, JpegError, NotJpeg, UnsupportedJpeg  # noqa: F401
from ._parse import ColorComponent, HuffmanTable, parse_jpeg  # noqa: F401
from .jpeg_decoder import JpegDecoder  # noqa: F401
from .batch import BatchDecoder, prepare_batch  # noqa: F401

__all__ = ["JpegDecoder", "BatchDecoder", "prepare_batch", "parse_jpeg", "ColorComponent", "HuffmanTable",
           "JpegError", "NotJpeg", "CorruptedJpeg", "UnsupportedJpeg", "BackendError"]
