// Stage 1 of progressive files, the first scans of a band (Ah = 0): DC scans (jpeg_decoder.py:974-1029) and AC scans
// (:1122-1179, :1236-1250).  One wavefront walks one restart segment of one scan on the stage-0 stream, with the symbols
// looked up 64 bit offsets at a time — see progressive_refine.hip, which this file shares its stream ring and symbol
// descriptions with (prog_stream.h).  A first scan only writes:
//
//   * AC: the block's new coefficients are collected in the lanes (lane = zig-zag index, v_writelane) together with a mask
//     of the positions written, and stored once per block; blocks inside an end-of-band run are skipped without being
//     touched;
//   * DC: the predictor chain (:1018-1020) is the serial part; the values of 64 consecutive blocks of the scan are collected
//     in the lanes and stored together — the blocks of an interleaved scan are consecutive in the coefficient store, those
//     of a single-component scan one MCU apart.  An interleaved scan uses one Huffman table per component, so each window of
//     bit offsets is looked up in all of them and the walk takes the entry of the component whose turn it is.
//
// ~20 instructions per symbol instead of the ~100 of progressive.hip's general walk, which keeps the scans that are
// neither (DC refinement: one bit per block; sequential scans of non-interleaved baseline files).
#include "mijpeg_internal.h"
#include "prog_stream.h"

namespace mj {

using namespace progstream;

template <bool DC>
__global__ __launch_bounds__(256) void k_progressive_first(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                           const DevProgSeg *__restrict__ segs, int n_segs,
                                                           const DevProgScan *__restrict__ scans, const DevImage *__restrict__ images,
                                                           const DevHuff *__restrict__ huff, const uint16_t *__restrict__ lut11p,
                                                           int16_t *__restrict__ coef, int32_t *__restrict__ status, int tr) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[4][DC ? 3 : 1][kPLut];
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[4][kRingDw];
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int seg_id = blockIdx.x * 4 + wave;
    if (seg_id >= n_segs) return;                              // wave-uniform; no workgroup barriers below
    const DevProgSeg *sg = segs + seg_id;
    const DevProgScan *sc = scans + sg->scan;
    const int ss = sc->ss, se = sc->se, al = sc->al;
    const bool sequential = ss == 0 && se == 63;
    if (sc->ah != 0 || sequential || (ss == 0) != DC) return;  // refining scans: progressive_refine.hip / progressive.hip
    constexpr bool is_dc = DC;
    const DevImage *im = images + sc->image;
    const int nsc = sc->n_comp;
    uint16_t *lut = s_lut[wave][0];
    const int n_tabs = is_dc ? nsc : 1;
    for (int t = 0; t < n_tabs; ++t) load_lut(lut + t * kPLut, lut11p, is_dc ? sc->dc_tab[t] : sc->ac_tab[0], lane);
    Stream st;
    st.init(s_ring[wave], stream, seg_bits, sg, lane);

    const int hmax = im->hmax, vmax = im->vmax, bpm = im->blocks_per_mcu, fmx = im->mcu_count_h;
    const int ncf = im->ncomp;
    int16_t *cbase = coef + im->block_off * 64;
    const int m_lo = sg->mcu0, m_hi = sg->mcu0 + sg->n_mcu;
    int err = 0;
    int gbase = 0;

    if constexpr (DC) {
        // ------------------------------------------------------------ DC, first scan (:974-1029)
        // blocks per MCU of each scan component (:980-1003): an interleaved scan covers the frame's MCUs, a single-component
        // one (never of a subsampled-luma component: api.hip) one block per MCU
        const int cA = sc->comp[0], cB = nsc > 1 ? sc->comp[1] : 0, cC = nsc > 2 ? sc->comp[2] : 0;
        const int nA = (nsc > 1 && cA == 0) ? hmax * vmax : 1, nB = nsc > 1 ? ((cB == 0) ? hmax * vmax : 1) : 0;
        const int nC = nsc > 2 ? ((cC == 0) ? hmax * vmax : 1) : 0;
        const int bps = nA + nB + nC;
        auto first_of = [&](int c) { return c == 0 ? 0 : hmax * vmax + c - 1; };      // a component's first block in the frame's MCU
        uint32_t veA0, veB0 = 0, veC0 = 0, vw0, veA1, veB1 = 0, veC1 = 0, vw1;
        auto lookup64 = [&](int g, uint32_t &va, uint32_t &vb, uint32_t &vc, uint32_t &vw) {
            const uint32_t w = st.bits_at(g + lane);
            const uint32_t i11 = w >> (32 - kProgLutBits);
            vw = w;
            const uint32_t ea = lut[i11];
            va = dc_entry(w, (int)(ea >> 8), (int)(ea & 255u));
            if (nsc > 1) { const uint32_t eb = lut[kPLut + i11]; vb = dc_entry(w, (int)(eb >> 8), (int)(eb & 255u)); }
            if (nsc > 2) { const uint32_t ec = lut[2 * kPLut + i11]; vc = dc_entry(w, (int)(ec >> 8), (int)(ec & 255u)); }
        };
        lookup64(0, veA0, veB0, veC0, vw0);
        lookup64(64, veA1, veB1, veC1, vw1);
        int pred0 = 0, pred1 = 0, pred2 = 0;
        const int total = (m_hi - m_lo) * bps;
        int vdc = 0;
        int j = 0;                                                        // block of the MCU
        for (int t = 0; t < total && !err; ++t) {
            if ((t & 15) == 0) st.top_up();
            int off = st.bp - gbase;
            if (__builtin_expect(off >= 64, 0)) {
                if (off < 128) {
                    veA0 = veA1; veB0 = veB1; veC0 = veC1; vw0 = vw1; gbase += 64;
                } else {
                    gbase = st.bp;
                    lookup64(gbase, veA0, veB0, veC0, vw0);
                }
                lookup64(gbase + 64, veA1, veB1, veC1, vw1);
                off = st.bp - gbase;
            }
            const int ci = j < nA ? 0 : (j < nA + nB ? 1 : 2);            // scan component of this block
            uint32_t e = ci == 0 ? rdl(veA0, off) : (ci == 1 ? rdl(veB0, off) : rdl(veC0, off));
            if (__builtin_expect((e & 3u) != 0u, 0)) {                    // a code longer than the LUT's index, or none
                const uint32_t w = rdl(vw0, off);
                int len, s;
                long_code(w, huff + sc->dc_tab[ci], len, s);
                e = dc_entry(w, len, s);
                if (e & 3u) { err = MJ_ST_BAD_CODE; break; }
            }
            st.bp += (int)((e >> 6) & 63u);
            const int pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
            const int dcv = (int)(int16_t)((int)(e >> 16) + pred);      // (:1018-1020)
            if (ci == 0) pred0 = dcv; else if (ci == 1) pred1 = dcv; else pred2 = dcv;
            write_lane(vdc, (int)(int16_t)(dcv << al), t & 63);         // (:1029)
            if (++j == bps) j = 0;
            if ((t & 63) == 63 || t + 1 == total) {                      // 64 blocks' values at once, lane l = block t0 + l of the segment
                const int tl = (t & ~63) + lane;
                if (tl <= t) {
                    const int mm = tl / bps, jj = tl - mm * bps;
                    const int cc = jj < nA ? cA : (jj < nA + nB ? cB : cC);
                    const int rr = jj < nA ? jj : (jj < nA + nB ? jj - nA : jj - nA - nB);
                    cbase[((int64_t)(m_lo + mm) * bpm + first_of(cc) + rr) * 64] = (int16_t)vdc;
                }
            }
        }
    } else {
        // ------------------------------------------------------------ AC, first scan of the band (:1122-1179, :1236-1250)
        const int c = sc->comp[0];
        const DevHuff *tab = huff + sc->ac_tab[0];
        const int h = (ncf > 1 && c == 0) ? hmax : 1, v = (ncf > 1 && c == 0) ? vmax : 1;
        const int first = c == 0 ? 0 : hmax * vmax + c - 1;
        const int smh = sc->mcu_count_h;
        const int nz_nat = c_nat_of_zz_ps[lane];
        const int nat = tr ? ((nz_nat & 7) << 3 | nz_nat >> 3) : nz_nat;      // tr: blocks are kept [u][v] for the row-major stage 2
        const int lh = h == 4 ? 2 : h - 1, lv = v == 4 ? 2 : v - 1;           // h, v are 1, 2 or 4
        int by = m_lo / smh, bx = m_lo - by * smh;
        uint32_t ve0, vw0, ve1, vw1;
        auto lookup64 = [&](int g, uint32_t &ve, uint32_t &vw) {
            const uint32_t w = st.bits_at(g + lane);
            const uint32_t e16 = lut[w >> (32 - kProgLutBits)];
            vw = w;
            ve = ac_entry<false>(w, (int)(e16 >> 8), (int)(e16 & 255u), al);
        };
        lookup64(0, ve0, vw0);
        lookup64(64, ve1, vw1);
        int eobrun = 0;
        for (int m = m_lo; m < m_hi && !err; ++m) {
            if (eobrun > 0) {                              // the block lies in an end-of-band run: nothing of this band in it
                --eobrun;
            } else {
                st.top_up();
                int cf = 0;
                uint64_t touched = 0;
                int k = ss;
                for (;;) {
                    int off = st.bp - gbase;
                    if (__builtin_expect(off >= 64, 0)) {
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
                        if ((e & 3u) == 3u) {              // a code longer than the LUT's index (rare) or no code at all
                            const uint32_t w = rdl(vw0, off);
                            int len, hv;
                            long_code(w, tab, len, hv);
                            e = ac_entry<false>(w, len, hv, al);
                            if (len == 0) { err = MJ_ST_BAD_CODE; break; }
                        }
                        if (e & 2u) {                      // end of band: the run counts this block (:1160-1166)
                            eobrun = (int)(e >> 16) - 1;
                            st.bp += (int)((e >> 6) & 31u);
                            break;
                        }
                        if (e & 1u) {                      // ZRL: sixteen zeros (:1170)
                            k += 16;
                            st.bp += (int)((e >> 6) & 31u);
                            if (k > se) break;
                            continue;
                        }
                    }
                    k += (int)((e >> 2) & 15u);
                    if (__builtin_expect(k > 63, 0)) { err = MJ_ST_OVERRUN; break; }
                    write_lane(cf, (int)e >> 16, k);       // (:1248-1250)
                    touched |= (uint64_t)1 << k;
                    st.bp += (int)((e >> 6) & 31u);
                    if (++k > se) break;
                }
                if (touched != 0) {
                    const int mx = bx >> lh, my = by >> lv;
                    int16_t *p = cbase + ((int64_t)(my * fmx + mx) * bpm + first + ((by - (my << lv)) << lh) + (bx - (mx << lh))) * 64;
                    if ((touched >> lane) & 1) p[nat] = (int16_t)cf;
                }
            }
            if (++bx == smh) { bx = 0; ++by; }
        }
    }

    if (!err) err = st.end_status(sg->last != 0);
    if (err && lane == 0) atomicMax(status + sc->image, err);
}

// n_dc segments of DC first scans, then n_ac segments of AC first scans
hipError_t launch_progressive_first(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevProgSeg *segs,
                                    int n_dc, int n_ac, const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                    const uint16_t *lut11p, int16_t *coef, int32_t *status, int transposed) {
    if (n_dc > 0)
        hipLaunchKernelGGL(k_progressive_first<true>, dim3((unsigned)((n_dc + 3) / 4)), dim3(256), 0, stream, dstream, seg_bits, segs, n_dc,
                           scans, images, huff, lut11p, coef, status, transposed);
    if (n_ac > 0)
        hipLaunchKernelGGL(k_progressive_first<false>, dim3((unsigned)((n_ac + 3) / 4)), dim3(256), 0, stream, dstream, seg_bits, segs + n_dc,
                           n_ac, scans, images, huff, lut11p, coef, status, transposed);
    return hipGetLastError();
}

}  // namespace mj
