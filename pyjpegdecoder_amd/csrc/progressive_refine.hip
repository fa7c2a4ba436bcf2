// Stage 1 of progressive files, the refining AC scans (Ah > 0, Ss > 0) — jpeg_decoder.py:1122-1298 with the
// correction queue of :1100-1115.  These scans are most of a progressive file's entropy-coded bytes (the last luma
// scan alone is half of a libjpeg-default file) and they cannot be cut into independent pieces: how many correction
// bits follow a symbol depends on which coefficients of the block earlier scans left non-zero.  What is left to
// optimise is the serial chain per symbol; one wavefront walks one restart segment of one scan, and a lone
// wavefront issues one instruction every ~5 cycles whatever the instruction is, so the chain is counted in
// instructions.  progressive.hip's general walk spends ~170 of them per symbol (bit-buffer refills with the 0xFF
// rule, an LDS round trip per Huffman symbol, a loop per skipped zero, a loop per 16 correction bits).  Here:
//
//   * stage 0 (destuff.hip) has already applied the byte rules: the segment is a big-endian dword stream, a bit
//     position is one integer, and a 256-dword ring of it sits in LDS (refilled 64 dwords at a time, one global
//     load in flight);
//   * Huffman symbols are looked up 64 bit-offsets at a time: lane l decodes the symbol that WOULD start at bit
//     gbase + l (two ring dwords, one LUT read) and packs what the walk needs — class, zero run, bits consumed by
//     code and value, the value already extended and shifted — into one dword.  The walk picks its symbol with a
//     v_readlane; a 64-bit window holds ~15 symbols of a final refinement scan, and the next window is looked up
//     while this one is being consumed;
//   * per block, the zero-history positions are turned into a table once (ds_permute: ordinal -> position), so
//     "skip r zeros, then the next zero" (:1184-1215) is one v_readlane at ordinal jz + r, and the number of
//     history-non-zero coefficients passed on the way — the correction bits to skip — is a second v_readlane into the
//     prefix count of the non-zero mask.  The masks of the block's history do not change while the block is walked
//     (a coefficient placed by this scan lies behind everything later symbols look at);
//   * correction bits are not read when their symbol is decoded: every lane remembers where its bit will be
//     (one v_cmp/v_cndmask per symbol) and the whole block's corrections are fetched from the ring and applied at
//     once when the block ends; the new coefficients go into the lanes with v_writelane.
//
// ~35 instructions per coefficient symbol; blocks inside an end-of-band run cost ~40 in all.
#include "mijpeg_internal.h"

namespace mj {

namespace {

constexpr int kRingDw = 256;              // stream ring per wave
constexpr int kPLut = 1 << kProgLutBits;

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t from_bit(int k) { return k >= 64 ? 0 : ~(uint64_t)0 << k; }          // bits k..63
__device__ __forceinline__ uint64_t bit_range(int a, int b) { return from_bit(a) & ~from_bit(b); }        // bits a..b-1
__device__ __forceinline__ int mbcnt(uint64_t m) {                                                       // bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__constant__ uint8_t c_nat_of_zz_r[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// What the walk needs to know about the symbol that starts at the top bit of `w`, given its Huffman code's length and
// value (len = 0: no code of <= kProgLutBits bits matches):
//   bits 1..0   class: 0 = coefficient (size > 0) or ZRL — "skip r zeros, take the next zero" for both, a ZRL places 0
//                      into a coefficient that is 0 —, 2 = end of band (EOBn), 3 = not in the table
//   bits 5..2   zero run r (15 for ZRL)
//   bits 10..6  bits consumed by the code and what belongs to it (value bits / the EOB run's extra bits)
//   bits 31..16 class 0: the new coefficient, extended (:1636-1646), shifted by Al and cut to int16 (:1225);
//               class 2: the length of the end-of-band run, (1 << r) + extra bits (:1160-1166)
__device__ __forceinline__ uint32_t symbol_entry(uint32_t w, int len, int hv, int al) {
    const int r = hv >> 4, s = hv & 15;
    if (len == 0) return 3u;
    if (s == 0 && r != 15) {
        const uint32_t extra = r ? (w << len) >> (32 - r) : 0u;
        return 2u | ((uint32_t)r << 2) | ((uint32_t)(len + r) << 6) | (((1u << r) + extra) << 16);
    }
    uint32_t val16 = 0;
    if (s > 0) {
        const uint32_t raw = (w << len) >> (32 - s);
        const int val = (raw >> (s - 1)) ? (int)raw : (int)raw - ((1 << s) - 1);
        val16 = (uint32_t)(uint16_t)(int16_t)(val << al);
    }
    return 0u | ((uint32_t)r << 2) | ((uint32_t)(len + s) << 6) | (val16 << 16);
}

}  // namespace

__global__ __launch_bounds__(256) void k_progressive_refine(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                            const DevProgSeg *__restrict__ segs, int n_segs,
                                                            const DevProgScan *__restrict__ scans, const DevImage *__restrict__ images,
                                                            const DevHuff *__restrict__ huff, const uint16_t *__restrict__ lut11p,
                                                            int16_t *__restrict__ coef, int32_t *__restrict__ status, int spec_refine, int tr) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[4][kPLut];
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[4][kRingDw];
    const bool spec = spec_refine != 0;
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int seg_id = blockIdx.x * 4 + wave;
    if (seg_id >= n_segs) return;                              // wave-uniform; no workgroup barriers below
    const DevProgSeg *sg = segs + seg_id;
    const DevProgScan *sc = scans + sg->scan;
    const int ss = sc->ss, se = sc->se, al = sc->al;
    if (!(sc->ah != 0 && ss > 0)) return;                      // every other kind of scan: progressive.hip
    const DevImage *im = images + sc->image;
    uint16_t *lut = s_lut[wave];
    uint32_t *ring = s_ring[wave];
    const DevHuff *tab = huff + sc->ac_tab[0];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(lut11p + (size_t)sc->ac_tab[0] * kPLut);
#pragma unroll
        for (int i = 0; i < kPLut * 2 / 16 / 64; ++i) reinterpret_cast<uint4 *>(lut)[i * 64 + lane] = src[i * 64 + lane];
    }

    // ---- the segment's stream: stage 0 wrote it at dword (begin >> 2) + segment number
    const uint32_t *sw = stream + (sg->begin >> 2) + sg->stream_slot;
    const int total_bits = seg_bits[sg->stream_slot];
    const int n_dw = (total_bits + 31) >> 5;
    auto chunk = [&](int d0) -> uint32_t { const int d = d0 + lane; return d < n_dw ? sw[d] : 0u; };    // zeros behind the end (:689-693)
    int whi = 0;                              // dwords [whi - 256, whi) are in the ring; `pend` holds [whi, whi + 64)
    uint32_t pend = chunk(0);
    int bp = 0;                               // next bit
    auto top_up = [&]() {                     // at least 128 dwords ahead of the next bit: a block takes 62 at most
        while (whi - (bp >> 5) < 128) {
            ring[(whi + lane) & (kRingDw - 1)] = pend;
            whi += 64;
            pend = chunk(whi);
        }
    };
    top_up();

    // ---- symbols of 64 consecutive bit offsets (see symbol_entry)
    auto lookup64 = [&](int g, uint32_t &ve, uint32_t &vw) {
        const int q = g + lane, d = q >> 5, sh = q & 31;
        const uint32_t a = ring[d & (kRingDw - 1)], b = ring[(d + 1) & (kRingDw - 1)];
        const uint32_t w = (uint32_t)((((uint64_t)a << 32 | b) << sh) >> 32);
        const uint32_t e16 = lut[w >> (32 - kProgLutBits)];
        vw = w;
        ve = symbol_entry(w, (int)(e16 >> 8), (int)(e16 & 255u), al);
    };
    int gbase = 0;
    uint32_t ve0, vw0, ve1, vw1;              // offsets [gbase, gbase + 64) and [gbase + 64, gbase + 128)
    lookup64(0, ve0, vw0);
    lookup64(64, ve1, vw1);

    // ---- frame geometry (interleaved block order of the coefficient store), single-component scan
    const int c = sc->comp[0];
    const int hmax = im->hmax, vmax = im->vmax, bpm = im->blocks_per_mcu, fmx = im->mcu_count_h;
    const int ncf = im->ncomp;
    int16_t *cbase = coef + im->block_off * 64;
    const int h = (ncf > 1 && c == 0) ? hmax : 1, v = (ncf > 1 && c == 0) ? vmax : 1;
    const int first = c == 0 ? 0 : hmax * vmax + c - 1;
    const int smh = sc->mcu_count_h;
    const int nz_nat = c_nat_of_zz_r[lane];
    const int nat = tr ? ((nz_nat & 7) << 3 | nz_nat >> 3) : nz_nat;      // tr: blocks are kept [u][v] for the row-major stage 2
    // this lane's coefficient of the scan's blocks, one after the other in scan order (h, v are 1, 2 or 4: shifts)
    const int lh = h == 4 ? 2 : h - 1, lv = v == 4 ? 2 : v - 1;
    int nby = sg->mcu0 / smh, nbx = sg->mcu0 - nby * smh;
    auto next_elem = [&]() -> int16_t * {
        const int mx = nbx >> lh, my = nby >> lv;
        int16_t *p = cbase + ((int64_t)(my * fmx + mx) * bpm + first + ((nby - (my << lv)) << lh) + (nbx - (mx << lh))) * 64 + nat;
        if (++nbx == smh) { nbx = 0; ++nby; }
        return p;
    };
    const bool in_band = lane >= ss && lane <= se;
    const uint64_t band_from = from_bit(ss), band = bit_range(ss, se + 1);
    const int m_lo = sg->mcu0, m_hi = sg->mcu0 + sg->n_mcu;
    int err = 0, eobrun = 0;

    auto one_block = [&](int cf, int16_t *p) __attribute__((always_inline)) {
        top_up();
        const uint64_t nz0 = __ballot(cf != 0);
        const uint64_t nzb = nz0 & band_from;              // history: non-zero coefficients from Ss on
        const int rank0 = mbcnt(nzb);                      // ... how many of them below this lane
        int vbase = 0;                                     // this lane's correction bit is bit vbase + rank0 of the stream
        int kend;                                          // corrections go to the history-non-zero lanes below kend
        bool dirty = false;
        if (eobrun > 0) {                                  // inside an end-of-band run: a bit for every non-zero coefficient of the band
            vbase = bp;
            bp += __builtin_popcountll(nzb & band);
            kend = se + 1;
            --eobrun;
        } else {
            const uint64_t zb = ~nz0 & band_from;          // zero history
            const int nzeros = __builtin_popcountll(zb);
            const int zrank = mbcnt(zb);
            // position of the j-th zero: lane l sends its number to lane zrank (zeros) / behind all zeros (the others)
            const int slot = ((zb >> lane) & 1) ? zrank : nzeros + lane - zrank;
            const uint32_t zpos = (uint32_t)__builtin_amdgcn_ds_permute(slot << 2, lane);
            int k = ss, jz = 0, cnt = 0;                   // jz = zeros below k, cnt = history-non-zeros in [Ss, k)
            while (k <= se) {
                int off = bp - gbase;
                if (off >= 64) {                           // next window of looked-up symbols
                    if (off < 128) {
                        ve0 = ve1; vw0 = vw1; gbase += 64;
                    } else {
                        gbase = bp;
                        lookup64(gbase, ve0, vw0);
                    }
                    lookup64(gbase + 64, ve1, vw1);
                    off = bp - gbase;
                }
                uint32_t e = rdl(ve0, off);
                if ((e & 3u) == 3u) {                      // a code longer than the LUT's index (rare) or no code at all
                    const uint32_t w = rdl(vw0, off);
                    int len = 0, hv = 0;
                    for (int l = kProgLutBits + 1; l <= 16; ++l) {
                        const int dlt = (int)(w >> (32 - l)) - tab->first_code[l];
                        if (dlt >= 0 && dlt < tab->count[l]) { hv = tab->vals[tab->first_sym[l] + dlt]; len = l; break; }
                    }
                    if (len == 0) { err = MJ_ST_BAD_CODE; break; }
                    e = symbol_entry(w, len, hv, al);
                }
                const int adv = (int)((e >> 6) & 31u);
                if (e & 2u) {                              // end of band: this block's rest and eobrun - 1 further blocks (:1160-1166)
                    eobrun = (int)(e >> 16);
                    bp += adv;
                    break;
                }
                const int jt = jz + (int)((e >> 2) & 15u);
                if (jt >= nzeros) { err = MJ_ST_OVERRUN; break; }           // fewer zeros left than the run passes (:1190)
                const int pz = (int)rdl(zpos, jt);
                const int cn = (int)rdl((uint32_t)rank0, pz);
                asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(cf) : "s"((int)e >> 16), "s"(pz) : "m0");    // (:1225)
                const int sbase = bp + adv - cnt;          // the symbol's corrections follow its value bits (:1202, :1231)
                vbase = lane >= k ? sbase : vbase;
                bp += adv + (cn - cnt);
                cnt = cn; k = pz + 1; jz = jt + 1;
                dirty = true;
            }
            kend = k;
            if (!err && eobrun > 0) {                      // rest of the band, then the run continues in the next blocks
                const int sbase = bp - cnt;
                vbase = lane >= k ? sbase : vbase;
                bp += __builtin_popcountll(nzb & bit_range(k, se + 1));
                kend = max(k, se + 1);
                --eobrun;
            }
        }
        const uint64_t corr = nzb & ~from_bit(kend);
        if (corr != 0) {
            if ((corr >> lane) & 1) {
                const int bitpos = vbase + rank0;
                const uint32_t dw = ring[(bitpos >> 5) & (kRingDw - 1)];
                const int bit = (int)((dw >> (31 - (bitpos & 31))) & 1u);
                if (spec) cf = (int)(int16_t)(cf + (cf < 0 ? -(bit << al) : (bit << al)));   // T.81 G.1.2.3
                else cf = (int)(int16_t)(cf | (int)(int16_t)(bit << al));                    // the reference (:1114)
            }
            dirty = true;
        }
        // only this scan's band is written back: other scans of the same dependency level may be updating other
        // coefficients of the block at the same time
        if (dirty && in_band) *p = (int16_t)cf;
    };

    // four blocks' coefficients in flight: a block is ~1 us of HBM latency away and takes less than that to walk
    constexpr int D = 4;
    int cfq[D];
    int16_t *pq[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        pq[u] = cbase + nat;
        cfq[u] = 0;
        if (m_lo + u < m_hi) { pq[u] = next_elem(); cfq[u] = *pq[u]; }
    }
    for (int m = m_lo; m < m_hi && !err; m += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (m + u < m_hi && !err) {
                const int cf = cfq[u];
                int16_t *p = pq[u];
                if (m + u + D < m_hi) { pq[u] = next_elem(); cfq[u] = *pq[u]; }
                one_block(cf, p);
            }
        }
    }

    if (!err) {
        if (bp > total_bits) err = MJ_ST_OVERRUN;                                  // bits were read from behind the end
        else if (!sg->last && total_bits - bp >= 8) err = MJ_ST_DESYNC;            // a whole unread byte before the next restart marker
    }
    if (err && lane == 0) atomicMax(status + sc->image, err);
}

hipError_t launch_progressive_refine(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevProgSeg *segs,
                                     int n_segs, const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                     const uint16_t *lut11p, int16_t *coef, int32_t *status, int spec_refine, int transposed) {
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_progressive_refine, dim3((unsigned)((n_segs + 3) / 4)), dim3(256), 0, stream, dstream, seg_bits, segs,
                       n_segs, scans, images, huff, lut11p, coef, status, spec_refine, transposed);
    return hipGetLastError();
}

}  // namespace mj
