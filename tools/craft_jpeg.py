"""Write baseline JPEG files symbol by symbol, with ANY per-component sampling factors — test inputs for layouts no
encoder at hand produces (4:1:0, 1x4, luma below the chroma resolution, factors of 3 ...).

The coefficients are random (the decoders under test do not care what the picture shows); the Huffman and quantisation
tables are the Annex-K ones of tools/std_tables.h.  Input-generation tooling only; nothing here is on the decode path."""
from __future__ import annotations

import re
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent


def _std_tables():
    txt = (_HERE / "std_tables.h").read_text()
    out = {}
    for name, body in re.findall(r"(STD_\w+)\[\d+\]\s*=\s*\{([^}]*)\}", txt):
        out[name] = bytes(int(t, 16) for t in re.findall(r"0x([0-9a-fA-F]{2})", body))
    return out


_T = _std_tables()


def _codes(bits: bytes, vals: bytes):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            k += 1
            code += 1
        code <<= 1
    return out


_DC = [_codes(_T["STD_DC_LUMA_BITS"], _T["STD_DC_LUMA_VALS"]), _codes(_T["STD_DC_CHROMA_BITS"], _T["STD_DC_CHROMA_VALS"])]
_AC = [_codes(_T["STD_AC_LUMA_BITS"], _T["STD_AC_LUMA_VALS"]), _codes(_T["STD_AC_CHROMA_BITS"], _T["STD_AC_CHROMA_VALS"])]


def _seg(marker: int, payload: bytes) -> bytes:
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, length: int):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0x00)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _category(v: int) -> int:
    return int(abs(v)).bit_length()


def _value_bits(v: int, size: int) -> int:
    return v if v >= 0 else v + (1 << size) - 1


def random_block(rng, density: float, max_size: int, dc_size: int):
    """64 zig-zag coefficients; index 0 holds the DC DIFFERENCE."""
    zz = np.zeros(64, dtype=np.int64)
    s = int(rng.integers(0, dc_size + 1))
    if s:
        mag = int(rng.integers(1 << (s - 1), 1 << s))
        zz[0] = mag if rng.random() < 0.5 else -mag
    p = density * np.exp(-np.arange(1, 64) / 20.0)
    hit = rng.random(63) < p
    for k in np.nonzero(hit)[0] + 1:
        s = int(rng.integers(1, max_size + 1))
        mag = int(rng.integers(1 << (s - 1), 1 << s))
        zz[k] = mag if rng.random() < 0.5 else -mag
    return zz


def _encode_block(w: _Bits, zz, dc_codes, ac_codes):
    d = int(zz[0])
    s = _category(d)
    w.put(*dc_codes[s])
    if s:
        w.put(_value_bits(d, s), s)
    run = 0
    last = int(np.max(np.nonzero(zz[1:])[0])) + 1 if np.any(zz[1:]) else 0
    for k in range(1, last + 1):
        v = int(zz[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            w.put(*ac_codes[0xF0])
            run -= 16
        s = _category(v)
        w.put(*ac_codes[(run << 4) | s])
        w.put(_value_bits(v, s), s)
        run = 0
    if last < 63:
        w.put(*ac_codes[0x00])


def craft_baseline(width: int, height: int, factors, seed: int = 0, restart_interval: int = 0, density: float = 0.35,
                   max_size: int = 5, dc_size: int = 5) -> bytes:
    """A baseline file of `width` x `height` with one interleaved scan; `factors` = ((h, v), ...) per component (1 or 3 of them)."""
    factors = [tuple(f) for f in factors]
    ncomp = len(factors)
    assert ncomp in (1, 3)
    rng = np.random.default_rng(seed)
    hmax = max(h for h, _ in factors) if ncomp > 1 else 1
    vmax = max(v for _, v in factors) if ncomp > 1 else 1
    mcw, mch = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xFF\xD8")
    out += _seg(0xDB, b"\x00" + _T["STD_QT_LUMA_ZZ"])
    out += _seg(0xDB, b"\x01" + _T["STD_QT_CHROMA_ZZ"])
    sof = bytes([8]) + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([ncomp])
    for c, (h, v) in enumerate(factors):
        sof += bytes([c + 1, (h << 4) | v, 0 if c == 0 else 1])
    out += _seg(0xC0, sof)
    out += _seg(0xC4, b"\x00" + _T["STD_DC_LUMA_BITS"] + _T["STD_DC_LUMA_VALS"])
    out += _seg(0xC4, b"\x10" + _T["STD_AC_LUMA_BITS"] + _T["STD_AC_LUMA_VALS"])
    if ncomp > 1:
        out += _seg(0xC4, b"\x01" + _T["STD_DC_CHROMA_BITS"] + _T["STD_DC_CHROMA_VALS"])
        out += _seg(0xC4, b"\x11" + _T["STD_AC_CHROMA_BITS"] + _T["STD_AC_CHROMA_VALS"])
    if restart_interval:
        out += _seg(0xDD, restart_interval.to_bytes(2, "big"))
    sos = bytes([ncomp])
    for c in range(ncomp):
        sos += bytes([c + 1, 0x00 if c == 0 else 0x11])
    sos += bytes([0, 63, 0])
    out += _seg(0xDA, sos)
    w = _Bits()
    n_mcu = mcw * mch
    rst = 0
    for m in range(n_mcu):
        for c, (h, v) in enumerate(factors):
            rep = h * v if ncomp > 1 else 1
            t = 0 if c == 0 else 1
            for _ in range(rep):
                _encode_block(w, random_block(rng, density, max_size, dc_size), _DC[t], _AC[t])
        if restart_interval and (m + 1) % restart_interval == 0 and m + 1 != n_mcu:
            w.flush()
            w.out += bytes([0xFF, 0xD0 + (rst & 7)])
            rst += 1
    w.flush()
    out += w.out
    out += b"\xFF\xD9"
    return bytes(out)
