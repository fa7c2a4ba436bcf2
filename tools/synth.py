"""ctypes front-end of tools/jpegenc.c — synthetic baseline-JPEG inputs for tests and bench.py.

Input-generation tooling only; nothing here is on the decode path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libjpegenc.so"
_SRC = _HERE / "jpegenc.c"

SUBSAMPLING = {"444": 0, "422": 1, "420": 2, "440": 3, "grey": 4, "444ni": 5, "411": 6}   # 444ni: one scan per component


def build(force: bool = False) -> Path:
    if force or not _SO.exists() or _SO.stat().st_mtime < _SRC.stat().st_mtime:
        cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(_SRC), "-o", str(_SO), "-lm"]
        subprocess.run(cmd, check=True, cwd=str(_HERE))
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(str(_SO))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.mjenc_synth_rgb.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_float, u8p]
        lib.mjenc_synth_rgb.restype = None
        lib.mjenc_encode_rgb.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, u8p, ctypes.c_size_t]
        lib.mjenc_encode_rgb.restype = ctypes.c_long
        lib.mjenc_synth_batch.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t,
                                          ctypes.POINTER(ctypes.c_uint64)]
        lib.mjenc_synth_batch.restype = ctypes.c_long
        lib.mjenc_synth_mixed_batch.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
        lib.mjenc_synth_mixed_batch.restype = ctypes.c_long
        _lib = lib
    return _lib


def _u8p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def synth_rgb(seed: int, width: int, height: int, noise_sigma: float = 12.0) -> np.ndarray:
    """Row-major (H, W, 3) uint8 image of the SURVEY §8d content family."""
    out = np.empty((height, width, 3), dtype=np.uint8)
    _load().mjenc_synth_rgb(seed, width, height, noise_sigma, _u8p(out))
    return out


def encode_rgb(rgb: np.ndarray, quality: int = 85, subsampling: str = "420", restart_interval: int = 0) -> bytes:
    """Encode a row-major (H, W, 3) uint8 array as a baseline JPEG file."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    cap = w * h * 12 + 65536          # quality 100 on noise: up to ~27 bits per coefficient
    buf = np.empty(cap, dtype=np.uint8)
    n = _load().mjenc_encode_rgb(_u8p(rgb), w, h, quality, SUBSAMPLING[subsampling], restart_interval, _u8p(buf), cap)
    if n < 0:
        raise RuntimeError("jpegenc: encode failed")
    return buf[:n].tobytes()


def synth_jpeg(seed: int, width: int, height: int, quality: int = 85, subsampling: str = "420",
               restart_interval: int = 0, noise_sigma: float = 12.0) -> bytes:
    return encode_rgb(synth_rgb(seed, width, height, noise_sigma), quality, subsampling, restart_interval)


def synth_batch(n: int, seed0: int, width: int, height: int, quality: int = 85, subsampling: str = "420",
                restart_interval: int = 0, noise_sigma: float = 12.0):
    """n files (seeds seed0..seed0+n-1) encoded on all host cores.

    Returns (blob uint8[total], offsets uint64[n+1]).
    """
    cap = n * (width * height + 65536)  # q<=95 files of this family stay far below 1 B/pixel
    blob = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = _load().mjenc_synth_batch(n, seed0, width, height, noise_sigma, quality, SUBSAMPLING[subsampling],
                                      restart_interval, _u8p(blob), cap,
                                      offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    if total < 0:
        raise RuntimeError("jpegenc: batch encode failed")
    return blob[:total].copy(), offs


def synth_mixed_batch(n: int, seed0: int, width: int, height: int, subsampling: str = "420", restart_interval: int = 0):
    """n files of mixed content (tools/jpegenc.c: quality 50..95, noise 0..80 above / below a random split row), encoded on
    all host cores.  Returns (blob uint8[total], offsets uint64[n+1])."""
    cap = n * (3 * width * height + 65536)
    blob = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = _load().mjenc_synth_mixed_batch(n, seed0, width, height, SUBSAMPLING[subsampling], restart_interval, _u8p(blob), cap,
                                            offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    if total < 0:
        raise RuntimeError("jpegenc: mixed batch encode failed")
    return blob[:total].copy(), offs


if __name__ == "__main__":
    import sys
    build(force=True)
    data = synth_jpeg(0, 1920, 1080, restart_interval=120)
    print(len(data), "bytes for 1080p 4:2:0 q85 DRI=120")
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_bytes(data)
