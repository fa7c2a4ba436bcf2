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


def wide_ac_table(seed: int = 0):
    """(bits, vals) of an AC table that holds every (run, size) symbol with sizes 0..15 and run 0..15 that fits 256 symbols —
    the reference takes whatever size a symbol names (jpeg_decoder.py:862) — with code lengths from 2 to 16 bits, so that
    finished LUT entries, entries whose value bits are taken arithmetically, and second-level tables all occur."""
    rng = np.random.default_rng(seed)
    likely = [0x00, 0x01, 0x02, 0x11, 0x03, 0x21, 0xF0, 0x04, 0x12, 0x31, 0x05, 0x41]
    rest = [(r << 4) | s for s in range(0, 16) for r in range(16) if ((r << 4) | s) not in likely and not (s == 0 and r not in (0, 15))]
    rng.shuffle(rest)
    order = likely + rest
    counts = {2: 1, 3: 1, 4: 2, 5: 4, 6: 4, 7: 8, 8: 12, 9: 20, 10: 24, 11: 30, 12: 30, 13: 30, 14: 30, 15: 24, 16: len(order) - 220}
    assert sum(counts.values()) == len(order) and counts[16] > 0
    return bytes(counts.get(l, 0) for l in range(1, 17)), bytes(order)


def craft_baseline(width: int, height: int, factors, seed: int = 0, restart_interval: int = 0, density: float = 0.35,
                   max_size: int = 5, dc_size: int = 5, tables=None, ac_tables=None) -> bytes:
    """A baseline file of `width` x `height` with one interleaved scan; `factors` = ((h, v), ...) per component (1 or 3 of them).
    `tables` = per component (dc table, ac table) as indices into the table lists (default: luma tables for the first
    component, chroma tables for the others); `ac_tables` = extra (bits, vals) AC tables behind the two Annex-K ones (index
    2..): files whose components share or swap tables, up to four tables per class."""
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
    if tables is None:
        tables = [(0, 0) if c == 0 else (1, 1) for c in range(ncomp)]
    tables = [tuple(t) for t in tables]
    ac_specs = [(_T["STD_AC_LUMA_BITS"], _T["STD_AC_LUMA_VALS"]), (_T["STD_AC_CHROMA_BITS"], _T["STD_AC_CHROMA_VALS"])] + list(ac_tables or [])
    dc_specs = [(_T["STD_DC_LUMA_BITS"], _T["STD_DC_LUMA_VALS"]), (_T["STD_DC_CHROMA_BITS"], _T["STD_DC_CHROMA_VALS"])]
    ac_codes = [_codes(b, v) for b, v in ac_specs]
    for t in range(4):
        if t in {d for d, _ in tables}:
            out += _seg(0xC4, bytes([t]) + dc_specs[t][0] + dc_specs[t][1])
        if t in {a for _, a in tables}:
            out += _seg(0xC4, bytes([0x10 | t]) + ac_specs[t][0] + ac_specs[t][1])
    if restart_interval:
        out += _seg(0xDD, restart_interval.to_bytes(2, "big"))
    sos = bytes([ncomp])
    for c in range(ncomp):
        sos += bytes([c + 1, (tables[c][0] << 4) | tables[c][1]])
    sos += bytes([0, 63, 0])
    out += _seg(0xDA, sos)
    w = _Bits()
    n_mcu = mcw * mch
    rst = 0
    for m in range(n_mcu):
        for c, (h, v) in enumerate(factors):
            rep = h * v if ncomp > 1 else 1
            for _ in range(rep):
                # (the Annex-K tables have no symbol for sizes above 10)
                _encode_block(w, random_block(rng, density, max_size if tables[c][1] >= 2 else min(max_size, 10), dc_size),
                              _DC[tables[c][0]], ac_codes[tables[c][1]])
        if restart_interval and (m + 1) % restart_interval == 0 and m + 1 != n_mcu:
            w.flush()
            w.out += bytes([0xFF, 0xD0 + (rst & 7)])
            rst += 1
    w.flush()
    out += w.out
    out += b"\xFF\xD9"
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------------------
# Progressive files, scan by scan.  Nothing is transformed: every scan's symbols are drawn at random, subject only to what
# a decoder has seen so far (a refining scan sends a correction bit for every coefficient earlier scans left non-zero and
# may place +-1 where they left zero), so any scan script is possible — DC scans over a subset of the components,
# arbitrary bands, several refinement levels, long end-of-band runs, restart intervals.  The symbol order of the
# refining scans follows T.81 G.1.2.3 / figure G.7 (zero runs count zero-history coefficients only; corrections follow
# the symbol that passes them; an end-of-band run carries the corrections of everything it skips).

def _ac_prog_table(seed: int):
    """A Huffman table holding every (run, size) symbol a progressive AC scan can use (sizes 0..10; size 0 = EOBn / ZRL), code
    lengths from 3 to 14 bits so that LUT hits, long codes and second-level tables all occur."""
    rng = np.random.default_rng(seed)
    syms = [(r << 4) | s for s in range(0, 11) for r in range(16)]
    likely = [0x00, 0x01, 0x11, 0x02, 0x21, 0x10, 0xF0, 0x31, 0x41, 0x12, 0x03, 0x20]
    rest = [s for s in syms if s not in likely]
    rng.shuffle(rest)
    order = likely + rest
    counts = {3: 2, 4: 2, 5: 4, 6: 6, 7: 10, 8: 16, 9: 24, 10: 32, 12: 40, 14: 40}
    assert sum(counts.values()) == len(order) == 176
    bits = bytes(counts.get(l, 0) for l in range(1, 17))
    return bits, bytes(order)


DEFAULT_SCRIPT = [   # libjpeg's default for three components: (components, Ss, Se, Ah, Al)
    ((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2),
    ((0,), 1, 63, 2, 1), ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0),
]


def craft_progressive(width: int, height: int, factors, seed: int = 0, script=None, restart_interval: int = 0,
                      density: float = 0.2, new_density: float = 0.08, empty_block: float = 0.3, max_size: int = 4, dc_size: int = 4) -> bytes:
    factors = [tuple(f) for f in factors]
    ncomp = len(factors)
    assert ncomp in (1, 3)
    script = script or (DEFAULT_SCRIPT if ncomp == 3 else [((0,), 0, 0, 0, 1), ((0,), 1, 63, 0, 1), ((0,), 0, 0, 1, 0), ((0,), 1, 63, 1, 0)])
    rng = np.random.default_rng(seed)
    hmax = max(h for h, _ in factors) if ncomp > 1 else 1
    vmax = max(v for _, v in factors) if ncomp > 1 else 1
    mcw, mch = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    ac_bits, ac_vals = _ac_prog_table(seed)
    ac_codes = _codes(ac_bits, ac_vals)
    out = bytearray(b"\xFF\xD8")
    out += _seg(0xDB, b"\x00" + _T["STD_QT_LUMA_ZZ"])
    out += _seg(0xDB, b"\x01" + _T["STD_QT_CHROMA_ZZ"])
    sof = bytes([8]) + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([ncomp])
    for c, (h, v) in enumerate(factors):
        sof += bytes([c + 1, (h << 4) | v, 0 if c == 0 else 1])
    out += _seg(0xC2, sof)
    out += _seg(0xC4, b"\x00" + _T["STD_DC_LUMA_BITS"] + _T["STD_DC_LUMA_VALS"])
    out += _seg(0xC4, b"\x01" + _T["STD_DC_CHROMA_BITS"] + _T["STD_DC_CHROMA_VALS"])
    out += _seg(0xC4, b"\x10" + ac_bits + ac_vals)
    out += _seg(0xC4, b"\x11" + ac_bits + ac_vals)
    if restart_interval:
        out += _seg(0xDD, restart_interval.to_bytes(2, "big"))
    # what a decoder holds so far: non-zero flags per component, [block row][block column][zig-zag index], padded to whole MCUs
    fh = [(h, v) if ncomp > 1 else (1, 1) for h, v in factors]
    nz = [np.zeros((mch * v, mcw * h, 64), dtype=bool) for h, v in fh]

    for comps, ss, se, ah, al in script:
        # (table selectors: DC table 0 / 1 and AC table 0 / 1 for luma / the others)
        sos = bytes([len(comps)]) + b"".join(bytes([c + 1, ((0 if c == 0 else 1) << 4) | (0 if c == 0 else 1)]) for c in comps)
        sos += bytes([ss, se, (ah << 4) | al])
        out += _seg(0xDA, sos)
        w = _Bits()
        rst = [0]

        def restart_marker():
            w.flush()
            w.out.extend(bytes([0xFF, 0xD0 + (rst[0] & 7)]))
            rst[0] += 1

        if ss == 0:                                                    # ---- DC scan
            assert se == 0
            if len(comps) > 1:
                n_mcu = mcw * mch
            else:
                h, v = fh[comps[0]]
                bwn = -(-(width * h) // (hmax * 8)) if ncomp > 1 else -(-width // 8)
                bhn = -(-(height * v) // (vmax * 8)) if ncomp > 1 else -(-height // 8)
                n_mcu = bwn * bhn
            for m in range(n_mcu):
                for c in comps:
                    h, v = fh[c]
                    rep = h * v if len(comps) > 1 else 1
                    for _ in range(rep):
                        if ah == 0:
                            s = int(rng.integers(0, dc_size + 1))
                            w.put(*_DC[0 if c == 0 else 1][s])
                            if s:
                                w.put(int(rng.integers(0, 1 << s)), s)
                        else:
                            w.put(int(rng.integers(0, 2)), 1)
                if restart_interval and (m + 1) % restart_interval == 0 and m + 1 != n_mcu:
                    restart_marker()
            w.flush()
            out += w.out
            continue
        # ---- AC scan of one component, its own block raster (:612-619)
        assert len(comps) == 1
        c = comps[0]
        h, v = fh[c]
        bwn = -(-(width * h) // (hmax * 8)) if ncomp > 1 else -(-width // 8)
        bhn = -(-(height * v) // (vmax * 8)) if ncomp > 1 else -(-height // 8)
        n_mcu = bwn * bhn
        codes = ac_codes
        eobrun = [0]
        be = []                                                        # correction bits waiting behind the end-of-band run

        def emit_eobrun():
            if eobrun[0] > 0:
                nb = eobrun[0].bit_length() - 1
                w.put(*codes[nb << 4])
                if nb:
                    w.put(eobrun[0] & ((1 << nb) - 1), nb)
                eobrun[0] = 0
            for b in be:
                w.put(b, 1)
            be.clear()

        for m in range(n_mcu):
            by, bx = divmod(m, bwn)
            flags = nz[c][by, bx]
            if ah == 0:                                                # first scan of the band
                r = 0
                if rng.random() >= empty_block:
                    for k in range(ss, se + 1):
                        if rng.random() >= density * np.exp(-(k - ss) / 24.0):
                            r += 1
                            continue
                        emit_eobrun()
                        while r > 15:
                            w.put(*codes[0xF0])
                            r -= 16
                        s = int(rng.integers(1, max_size + 1))
                        w.put(*codes[(r << 4) | s])
                        w.put(int(rng.integers(0, 1 << s)), s)
                        flags[k] = True
                        r = 0
                else:
                    r = se - ss + 1
                if r > 0:
                    eobrun[0] += 1
                    if eobrun[0] == 0x7FFF:
                        emit_eobrun()
            else:                                                      # refining scan
                kinds = np.zeros(64, dtype=np.int8)                    # 0 stays zero, 1 history, 2 new
                kinds[flags] = 1
                if rng.random() >= empty_block:
                    new = (~flags) & (rng.random(64) < new_density)
                    kinds[new] = 2
                band = range(ss, se + 1)
                eob_at = max([k for k in band if kinds[k] == 2], default=-1)
                r, br = 0, []
                for k in band:
                    if kinds[k] == 0:
                        r += 1
                        continue
                    while r > 15 and k <= eob_at:
                        emit_eobrun()
                        w.put(*codes[0xF0])
                        r -= 16
                        for b in br:
                            w.put(b, 1)
                        br = []
                    if kinds[k] == 1:
                        br.append(int(rng.integers(0, 2)))
                        continue
                    emit_eobrun()
                    w.put(*codes[(r << 4) | 1])
                    w.put(int(rng.integers(0, 2)), 1)
                    for b in br:
                        w.put(b, 1)
                    br = []
                    r = 0
                    flags[k] = True
                if r > 0 or br:
                    eobrun[0] += 1
                    be.extend(br)
                    if eobrun[0] == 0x7FFF or len(be) > 900:
                        emit_eobrun()
            if restart_interval and (m + 1) % restart_interval == 0 and m + 1 != n_mcu:
                emit_eobrun()
                restart_marker()
        emit_eobrun()
        w.flush()
        out += w.out
    out += b"\xFF\xD9"
    return bytes(out)


# ---- random files of both kinds (tests, tools/stress_crafted.py, tools/crosscheck_reference.py --crafted) ----------------
def random_baseline(rng, seed: int) -> bytes:
    """Any factors 1..4 per component (at most 16 blocks per MCU), 1..119 pixels a side, with and without restart intervals."""
    f = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(3)]
    while sum(h * v for h, v in f) > 16:
        f = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(3)]
    w, h = int(rng.integers(1, 120)), int(rng.integers(1, 120))
    return craft_baseline(w, h, f, seed=seed, restart_interval=int(rng.integers(0, 4)), density=float(rng.uniform(0.05, 0.6)),
                          max_size=int(rng.integers(1, 8)), dc_size=int(rng.integers(1, 8)))


def random_script(rng, factors):
    """A scan script for `factors`: the DC components in random groups (a component on its own must be 1x1: the reference
    cannot walk the other kind), bands cut at random, successive approximation up to three levels deep, the scans of the
    components interleaved at random (every scan behind the ones it refines)."""
    nc = len(factors)
    comps = list(range(nc))
    a = int(rng.integers(0, 3))
    lone_ok = [c for c in comps if factors[c] == (1, 1) or nc == 1]
    groups = [tuple(comps)]
    if nc == 3 and lone_ok and rng.random() < 0.6:
        c = int(rng.choice(lone_ok))
        groups = [tuple(x for x in comps if x != c), (c,)]
        if rng.random() < 0.5:
            groups.reverse()
    script = [(g, 0, 0, 0, a) for g in groups]
    dc_ref = [[(g, 0, 0, lvl + 1, lvl) for g in groups] for lvl in range(a - 1, -1, -1)]
    ac = {c: [] for c in comps}
    for c in comps:
        al = int(rng.integers(0, 3))
        cuts = sorted(set([1, 64] + [int(x) for x in rng.integers(2, 64, size=int(rng.integers(0, 3)))]))
        ac[c].append([((c,), cuts[j], cuts[j + 1] - 1, 0, al) for j in range(len(cuts) - 1)])
        for lvl in range(al - 1, -1, -1):
            cuts = sorted(set([1, 64] + [int(x) for x in rng.integers(2, 64, size=int(rng.integers(0, 2)))]))
            ac[c].append([((c,), cuts[j], cuts[j + 1] - 1, lvl + 1, lvl) for j in range(len(cuts) - 1)])
    stages = [dc_ref] + [ac[c] for c in comps]          # lists of stages; the stages of one list stay in order
    while any(stages):
        j = int(rng.choice([k for k, s_ in enumerate(stages) if s_]))
        script += stages[j].pop(0)
    return script


def random_progressive(rng, seed: int) -> bytes:
    """A random layout the reference can finish (every component 1x1 or at the full resolution, or greyscale), a random
    script, 1..89 pixels a side."""
    H, V = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    while H * V + 2 > 16:                                       # (16 blocks per MCU is what the MI355X path holds; T.81 allows 10)
        H, V = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    kind = int(rng.integers(0, 4))
    factors = [((H, V), (1, 1), (1, 1)), ((1, 1), (H, V), (H, V)), ((H, V), (H, V), (1, 1)), ((1, 1),)][kind]
    if sum(h * v for h, v in factors) > 16:
        factors = ((H, V), (1, 1), (1, 1))
    return craft_progressive(int(rng.integers(1, 90)), int(rng.integers(1, 90)), factors, seed=seed, script=random_script(rng, factors),
                             restart_interval=int(rng.choice([0, 0, 1, 3, 7])), density=float(rng.uniform(0.05, 0.4)),
                             new_density=float(rng.uniform(0.02, 0.25)), empty_block=float(rng.uniform(0.0, 0.95)),
                             max_size=int(rng.integers(1, 7)), dc_size=int(rng.integers(1, 7)))
