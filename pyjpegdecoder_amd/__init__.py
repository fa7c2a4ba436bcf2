"""MI355X-native batched JPEG block-decode behind the reference's ``JpegDecoder`` class surface."""
from .errors import BackendError, CorruptedJpeg, JpegError, NotJpeg, UnsupportedJpeg  # noqa: F401
