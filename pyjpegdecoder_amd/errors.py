"""Decoder exceptions — same names and hierarchy as the reference (jpeg_decoder.py:1714-1725)."""


class JpegError(Exception):
    """Parent of all other exceptions of this decoder."""


class NotJpeg(JpegError):
    """File is not a JPEG image."""


class CorruptedJpeg(JpegError):
    """Failed to parse the file headers / entropy-coded data."""


class UnsupportedJpeg(JpegError):
    """JPEG image is encoded in a way that the decoder does not support."""


class BackendError(RuntimeError):
    """The HIP library (libmijpeg.so) is missing, failed to load, or reported an API/HIP error.

    There is deliberately no CPU fallback: the MI355X path fails loudly instead.
    """
