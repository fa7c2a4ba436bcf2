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
#include "prog_stream.h"

namespace mj {

using namespace progstream;

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
    load_lut(lut, lut11p, sc->ac_tab[0], lane);
    Stream st;
    st.init(ring, stream, seg_bits, sg, lane);

    // ---- symbols of 64 consecutive bit offsets (see ac_entry)
    auto lookup64 = [&](int g, uint32_t &ve, uint32_t &vw) {
        const uint32_t w = st.bits_at(g + lane);
        const uint32_t e16 = lut[w >> (32 - kProgLutBits)];
        vw = w;
        ve = ac_entry<true>(w, (int)(e16 >> 8), (int)(e16 & 255u), al);
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
    const int nz_nat = c_nat_of_zz_ps[lane];
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
        st.top_up();
        const uint64_t nz0 = __ballot(cf != 0);
        const uint64_t nzb = nz0 & band_from;              // history: non-zero coefficients from Ss on
        const int rank0 = mbcnt(nzb);                      // ... how many of them below this lane
        int vbase = 0;                                     // this lane's correction bit is bit vbase + rank0 of the stream
        int kend;                                          // corrections go to the history-non-zero lanes below kend
        bool dirty = false;
        if (eobrun > 0) {                                  // inside an end-of-band run: a bit for every non-zero coefficient of the band
            vbase = st.bp;
            st.bp += __builtin_popcountll(nzb & band);
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
            for (;;) {                                     // (Ss <= Se: a scan has at least one coefficient per block)
                int off = st.bp - gbase;
                if (__builtin_expect(off >= 64, 0)) {      // next window of looked-up symbols
                    if (off < 128) {
                        ve0 = ve1; vw0 = vw1; gbase += 64;
                    } else {
                        gbase = st.bp;
                        lookup64(gbase, ve0, vw0);
                    }
                    lookup64(gbase + 64, ve1, vw1);
                    off = st.bp - gbase;
                }
                uint32_t e = rdl(ve0, off);
                if (__builtin_expect((e & 3u) != 0u, 0)) {
                    if (e & 1u) {                          // a code longer than the LUT's index (rare) or no code at all
                        const uint32_t w = rdl(vw0, off);
                        int len, hv;
                        long_code(w, tab, len, hv);
                        e = ac_entry<true>(w, len, hv, al);
                        if (len == 0) { err = MJ_ST_BAD_CODE; break; }
                    }
                    if (e & 2u) {                          // end of band: this block's rest and eobrun - 1 further blocks (:1160-1166)
                        eobrun = (int)(e >> 16);
                        st.bp += (int)((e >> 6) & 31u);
                        break;
                    }
                }
                const int jt = jz + (int)((e >> 2) & 15u);
                if (__builtin_expect(jt >= nzeros, 0)) { err = MJ_ST_OVERRUN; break; }   // fewer zeros left than the run passes (:1190)
                const int pz = (int)rdl(zpos, jt);
                const int cn = (int)rdl((uint32_t)rank0, pz);
                write_lane(cf, (int)e >> 16, pz);         // (:1225)
                const int sbase = st.bp + (int)((e >> 6) & 31u) - cnt;      // the symbol's corrections follow its value bits (:1202, :1231)
                vbase = lane >= k ? sbase : vbase;
                st.bp = sbase + cn;
                cnt = cn; k = pz + 1; jz = jt + 1;
                if (k > se) break;
            }
            dirty = true;
            kend = k;
            if (!err && eobrun > 0) {                      // rest of the band, then the run continues in the next blocks
                const int sbase = st.bp - cnt;
                vbase = lane >= k ? sbase : vbase;
                st.bp += __builtin_popcountll(nzb & bit_range(k, se + 1));
                kend = max(k, se + 1);
                --eobrun;
            }
        }
        const uint64_t corr = nzb & ~from_bit(kend);
        if (corr != 0) {
            if ((corr >> lane) & 1) {
                const int bitpos = vbase + rank0;
                const int bit = st.bit_at(bitpos);
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

    if (!err) err = st.end_status(sg->last != 0);
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
