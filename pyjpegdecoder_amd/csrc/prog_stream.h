// What the progressive stage-1 walks on the stage-0 stream share (progressive_fast.hip): the per-wave ring of the
// segment's stream in LDS, the packed description of the symbol that would start at a given bit, and the windows of
// looked-up symbols.  See progressive_fast.hip for the design.
#pragma once
#include "mijpeg_internal.h"

namespace mj {
namespace progstream {

constexpr int kRingDw = 256;              // stream ring per wave (dwords)
constexpr int kPLut = 1 << kProgLutBits;
constexpr int kProgDcLutBits = 9;         // DC tables (a dozen symbols) get by with 9 bits: three of them fit where one AC table does
constexpr int kPDcLut = 1 << kProgDcLutBits;

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t from_bit(int k) { return k >= 64 ? 0 : ~(uint64_t)0 << k; }          // bits k..63
__device__ __forceinline__ uint64_t bit_range(int a, int b) { return from_bit(a) & ~from_bit(b); }        // bits a..b-1
__device__ __forceinline__ int mbcnt(uint64_t m) {                                                       // bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// v[lane l] = s (l wave-uniform).  The lane select of v_writelane shares the constant bus with the value: it goes through m0,
// which nothing else in these kernels uses (gfx950's LDS instructions do not read it).
__device__ __forceinline__ void write_lane(int &v, int s, int l) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(s), "s"(l) : "m0");
}

__constant__ uint8_t c_nat_of_zz_ps[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// One wave's view of its restart segment: stage 0 (destuff.hip) wrote the bytes the bit reader keeps as big-endian
// dwords; a 256-dword ring of them sits in LDS, topped up 64 dwords at a time with one global load in flight.
// Behind the segment's end the stream reads as zeros (jpeg_decoder.py:689-693).  All members but `pend` are wave-uniform.
struct Stream {
    uint32_t *ring;
    const uint32_t *sw;
    int n_dw, total_bits;
    int whi;              // dwords [whi - 256, whi) are in the ring; `pend` holds [whi, whi + 64)
    uint32_t pend;
    int bp;               // next bit
    int lane;

    __device__ __forceinline__ uint32_t chunk(int d0) const { const int d = d0 + lane; return (uint32_t)d < (uint32_t)n_dw ? sw[d] : 0u; }
    __device__ __forceinline__ void init(uint32_t *ring_, const uint32_t *stream, const int32_t *seg_bits, const DevProgSeg *sg, int lane_,
                                         int bp0 = 0) {
        ring = ring_; lane = lane_;
        sw = stream + (sg->begin >> 2) + sg->stream_slot;       // stage 0 wrote it at dword (begin >> 2) + segment number
        total_bits = seg_bits[sg->stream_slot];
        n_dw = (total_bits + 31) >> 5;
        bp = bp0;
        whi = (bp0 >> 5) & ~63;
        pend = chunk(whi);
        top_up();
    }
    // at least 128 dwords ahead of the next bit; call it often enough that no more than ~100 are consumed in between
    __device__ __forceinline__ void top_up() {
        while (whi - (bp >> 5) < 128) {
            ring[(whi + lane) & (kRingDw - 1)] = pend;
            whi += 64;
            pend = chunk(whi);
        }
    }
    // the 32 stream bits from bit q on (q per lane)
    __device__ __forceinline__ uint32_t bits_at(int q) const {
        const int d = q >> 5, sh = q & 31;
        const uint32_t a = ring[d & (kRingDw - 1)], b = ring[(d + 1) & (kRingDw - 1)];
        return (uint32_t)((((uint64_t)a << 32 | b) << sh) >> 32);
    }
    // ... in two steps, so that the LDS reads can be left in flight: the two ring dwords around bit q, and what they hold from q on
    __device__ __forceinline__ void raw_at(int q, uint32_t &a, uint32_t &b) const {
        const int d = q >> 5;
        a = ring[d & (kRingDw - 1)];
        b = ring[(d + 1) & (kRingDw - 1)];
    }
    static __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b, int q) {
        return (uint32_t)((((uint64_t)a << 32 | b) << (q & 31)) >> 32);
    }
    // one bit (q per lane)
    __device__ __forceinline__ int bit_at(int q) const {
        return (int)((ring[(q >> 5) & (kRingDw - 1)] >> (31 - (q & 31))) & 1u);
    }
    // OVERRUN / DESYNC at the end of the segment, as the other stage-1 kernels report them
    __device__ __forceinline__ int end_status(bool last) const {
        if (bp > total_bits) return MJ_ST_OVERRUN;                              // bits were read from behind the end
        if (!last && total_bits - bp >= 8) return MJ_ST_DESYNC;                 // a whole unread byte before the next restart marker
        return 0;
    }
};

__device__ __forceinline__ void load_lut(uint16_t *lds, const uint16_t *lut11p, int table, int lane) {
    const uint4 *src = reinterpret_cast<const uint4 *>(lut11p + (size_t)table * kPLut);
#pragma unroll
    for (int i = 0; i < kPLut * 2 / 16 / 64; ++i) reinterpret_cast<uint4 *>(lds)[i * 64 + lane] = src[i * 64 + lane];
}

// The refining scans' table (round 4): what ac_entry<true> would make of the symbol whose code fills the top of an 11-bit index,
// worked out once per wave instead of once per bit offset and window — a refining scan's symbols are (run, 0 or 1 value bit), so
// code and sign bit mostly fit the index.  16 bits: ac_entry's low half with the class copy's place (bits 9..8) taken by a value
// code — 0: zero (ZRL), 1: +(1 << Al), 2: -(1 << Al), 3: an end-of-band run of one — expanded by refine_entry() below.
// 0x0303 = "look again" (class 3): no code of <= 11 bits, code + value bit longer than the index, more than one value bit (the
// reference takes them as they come), an end-of-band run with extra bits; the walk's slow path then searches the code book from
// length 1.
__device__ __forceinline__ void load_lut_refine(uint16_t *lds, const uint16_t *lut11p, int table, int lane) {
    const uint16_t *src = lut11p + (size_t)table * kPLut;
#pragma unroll 2
    for (int i = lane; i < kPLut; i += 64) {
        const uint32_t e0 = src[i];
        const int len = (int)(e0 >> 8), r = (int)(e0 >> 4) & 15, sz = (int)e0 & 15;
        const int nbits = len + sz;
        const uint32_t bit = ((uint32_t)i >> ((kProgLutBits - 1 - len) & 15)) & 1u;         // the bit behind the code (len <= 10)
        uint32_t c = 0x0303u;
        if (len != 0) {
            if (sz == 0 && r != 15) { if (r == 0) c = 2u | (3u << 8) | ((uint32_t)len << 11); }
            else if (sz <= 1 && nbits <= kProgLutBits)
                c = ((uint32_t)(r + 1) << 2) | ((sz == 0 ? 0u : (bit ? 1u : 2u)) << 8) | ((uint32_t)nbits << 11);
        }
        lds[i] = (uint16_t)c;
    }
}
// vals = 0 | +(1 << Al) << 16 | -(1 << Al) << 32 | 1 << 48, each cut to 16 bits
__device__ __forceinline__ uint64_t refine_values(int al) {
    const uint64_t p = (uint64_t)((1u << al) & 0xFFFFu), n = (uint64_t)((0u - (1u << al)) & 0xFFFFu);
    return (p << 16) | (n << 32) | ((uint64_t)1 << 48);
}
__device__ __forceinline__ uint32_t refine_entry(uint32_t c, uint64_t vals) {
    const uint32_t vc = (c >> 8) & 3u;
    const uint32_t hi = (uint32_t)(vals >> (16 * vc)) & 0xFFFFu;
    return (c & 0xF87Fu) | ((c & 3u) << 8) | (hi << 16);
}

// the 9-bit LUT of a DC table from its 11-bit one: codes of 10 and 11 bits become misses (the long-code search finds them)
__device__ __forceinline__ void load_dc_lut(uint16_t *lds, const uint16_t *lut11p, int table, int lane) {
    const uint16_t *src = lut11p + (size_t)table * kPLut;
#pragma unroll
    for (int i = 0; i < kPDcLut / 64; ++i) {
        const uint16_t e = src[(i * 64 + lane) << (kProgLutBits - kProgDcLutBits)];
        lds[i * 64 + lane] = (e >> 8) <= kProgDcLutBits ? e : (uint16_t)0;
    }
}

// a code of `from` or more bits at the top of w (rare): canonical search (jpeg_decoder.py:712-722); len 0 = none
__device__ __forceinline__ void long_code(uint32_t w, const DevHuff *tab, int from, int &len, int &hv) {
    len = 0; hv = 0;
    for (int l = from; l <= 16; ++l) {
        const int dlt = (int)(w >> (32 - l)) - tab->first_code[l];
        if (dlt >= 0 && dlt < tab->count[l]) { hv = tab->vals[tab->first_sym[l] + dlt]; len = l; return; }
    }
}

// What a walk needs to know about the AC symbol that starts at the top bit of `w`, given its Huffman code's length and
// value (len = 0: no code of <= kProgLutBits bits matches):
//   bits 1..0   class: 0 = coefficient (size > 0), 1 = ZRL, 2 = end of band (EOBn), 3 = not in the table.
//               ZRL_IS_COEF: a ZRL is class 0 with value 0 (the refining walk: "skip r zeros, take the next zero" for
//               both, and a ZRL then places 0 into a coefficient that is 0)
//   bits 6..2   coefficients (class 0): zero run r + 1, what the walks' positions move on by; ZRL and end-of-band entries: r
//   bits 9..8   the class again (bits 7..6 are zero): the 8 bits from bit 2 on read  r + 64 * class, so a symbol loop that
//               adds them to a position of at most 63 finds every entry that is no plain coefficient behind its limit — one
//               test for "special entry" and "run past the end" (round 4: two instructions less per symbol)
//   bits 15..11 bits consumed by the code and what belongs to it (value bits / the EOB run's extra bits)
//   bits 31..16 class 0: the coefficient, extended (:1636-1646), shifted by Al and cut to int16 (:1225, :1248);
//               class 2: the length of the end-of-band run, (1 << r) + extra bits (:1160-1166)
template <bool ZRL_IS_COEF>
__device__ __forceinline__ uint32_t ac_entry(uint32_t w, int len, int hv, int al) {
    // (no branches: a window's 64 lanes hold every kind of entry, so a branch would be walked on both sides anyway)
    const int r = hv >> 4, s = hv & 15;
    const bool eob = s == 0 && r != 15;
    const int n = eob ? r : s;                                              // bits behind the code: the run's extra bits / the value
    const uint32_t raw = ((w << len) >> 1) >> (31 - n);                     // (0 for n = 0)
    const uint32_t neg = ((raw << 1) >> s) ? 0u : (1u << s) - 1u;           // top bit clear: the value is raw - (2^s - 1) (:1636-1646)
    const uint32_t val16 = (uint32_t)(uint16_t)(int16_t)((int)(raw - neg) << al);
    const uint32_t cls = eob ? 2u : (!ZRL_IS_COEF && s == 0) ? 1u : 0u;
    const uint32_t hi = eob ? (1u << r) + raw : val16;
    const uint32_t e = cls | (cls << 8) | ((uint32_t)((ZRL_IS_COEF ? !eob : cls == 0u) ? r + 1 : r) << 2) | ((uint32_t)(len + n) << 11) | (hi << 16);
    return len == 0 ? (3u | (3u << 8)) : e;
}

// ... and about a DC symbol (:1012-1029): bits 1..0 = 0, or 3 = not in the table / a size above 16; bits 11..6 the bits
// consumed (code + size <= 32); bits 31..16 the difference, extended, modulo 2^16 (the predictor wraps to int16)
__device__ __forceinline__ uint32_t dc_entry(uint32_t w, int len, int s) {
    if (len == 0 || s > 16) return 3u;
    uint32_t d16 = 0;
    if (s > 0) {
        const uint32_t raw = (uint32_t)(((uint64_t)w << len & 0xFFFFFFFFull) >> (32 - s));
        const int val = (raw >> (s - 1)) ? (int)raw : (int)raw - ((1 << s) - 1);
        d16 = (uint32_t)val & 0xFFFFu;
    }
    return ((uint32_t)(len + s) << 6) | (d16 << 16);
}

// The AC walks' look-ups, three windows of 64 bit offsets deep so that no LDS read is waited for: window 0
// [gbase, gbase + 64) is ready (ve0 = ac_entry per offset, vw0 = the 32 stream bits from there), window 1 has its bits and
// its LUT read in flight, window 2 its two ring reads.  advance() moves on by one window; the values it consumes were
// requested a whole window (~15 symbols) earlier.
template <bool ZRL_IS_COEF>
struct AcWindows {
    uint32_t ve0, vw0, w1, e1, a2, b2;
    int gbase;
    // (the refining walks' LUT holds finished entries, see load_lut_refine)
    static __device__ __forceinline__ uint32_t entry_of(uint32_t w, uint32_t e0, int al) {
        if constexpr (ZRL_IS_COEF) return refine_entry(e0, refine_values(al));
        else return ac_entry<ZRL_IS_COEF>(w, (int)(e0 >> 8), (int)(e0 & 255u), al);
    }
    __device__ __forceinline__ void start(const Stream &st, const uint16_t *lut, int al, int lane, int g) {
        gbase = g;
        vw0 = st.bits_at(g + lane);
        const uint32_t e0 = lut[vw0 >> (32 - kProgLutBits)];
        ve0 = entry_of(vw0, e0, al);
        w1 = st.bits_at(g + 64 + lane);
        e1 = lut[w1 >> (32 - kProgLutBits)];
        st.raw_at(g + 128 + lane, a2, b2);
    }
    __device__ __forceinline__ void advance(const Stream &st, const uint16_t *lut, int al, int lane) {
        gbase += 64;
        ve0 = entry_of(w1, e1, al);
        vw0 = w1;
        w1 = Stream::combine(a2, b2, gbase + 64 + lane);
        e1 = lut[w1 >> (32 - kProgLutBits)];
        st.raw_at(gbase + 128 + lane, a2, b2);
    }
    // the symbol at bit position bp lies `off` = bp - gbase >= 64 bits into the windows
    __device__ __forceinline__ void move_to(const Stream &st, const uint16_t *lut, int al, int lane, int off) {
        if (off < 128) advance(st, lut, al, lane);
        else start(st, lut, al, lane, st.bp);
    }
};

}  // namespace progstream
}  // namespace mj
