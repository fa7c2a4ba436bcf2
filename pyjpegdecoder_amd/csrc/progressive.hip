// Stage 1 for progressive files — one SOS (scan) of every image per launch, one wavefront per restart segment.
//
// Replaces JpegDecoder.progressive_dct_scan's entropy part (jpeg_decoder.py:908-1304).  The coefficient store
// that the reference keeps inside image_array between scans (:1029, :1225) is the same HBM array the baseline
// path uses — int16 blocks [v][u] in interleaved MCU order — so that after the last scan the ordinary stage-2
// kernel performs the reference's final pass (:1306-1362).  Scans of one image depend on each other, hence one
// launch per dependency level (scans that touch disjoint coefficients share a launch) in stream order; within a
// scan restart segments are independent.
//
// The walk is wave-uniform (same bit reader as huffman.hip); the lanes hold the 64 zig-zag coefficients of the
// block being refined, which turns the reference's element-by-element queues into mask arithmetic:
//   * "skip r zero coefficients, queueing the non-zero ones passed" (:1184-1193) = clear r low bits of the
//     zero mask above the cursor; the queue is the non-zero mask of the span;
//   * "one correction bit per queued coefficient, in order" (:1107-1115) = lane l takes bit number
//     popcount(queue below l) of the next popcount(queue) bits;
//   * the correction is `value |= bit << Al` on the int16 two's complement exactly as the reference does it,
//     i.e. NOT the spec's decrement for negative values.
#include "mijpeg_internal.h"
#include "wave_bits.h"

namespace mj {

namespace {

using namespace wavebits;

__constant__ uint8_t c_nat_of_zz_p[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ uint64_t bits_from(int k) { return k >= 64 ? 0 : ~(uint64_t)0 << k; }      // bits k..63
__device__ __forceinline__ uint64_t bits_range(int a, int b) { return bits_from(a) & ~bits_from(b); } // bits a..b-1

// take n <= 32 bits (0 allowed); the caller has refilled (>= 33 bits available)
__device__ __forceinline__ uint32_t take32(BitReader &br, int n) {
    if (n == 0) return 0;
    uint32_t v = (uint32_t)(br.bb >> (64 - n));
    br.bb <<= n;
    br.bc -= n;
    return v;
}

// one correction bit for every coefficient of `queue` (a mask over zig-zag positions), in ascending order
__device__ __forceinline__ void refine_queue(BitReader &br, uint64_t queue, int al, int lane, int &coef, bool &dirty, bool spec) {
    while (queue) {
        br.refill();
        // the queued positions of the lowest non-empty 16-position window (at most 16 bits per refill)
        const int w = __builtin_ctzll(queue) & ~15;
        const uint64_t part = queue & ((uint64_t)0xFFFF << w);
        const int n = __builtin_popcountll(part);
        queue &= ~part;
        const uint32_t bits = take32(br, n);
        if ((part >> lane) & 1) {
            const int rank = __builtin_popcountll(part & (((uint64_t)1 << lane) - 1));
            const int bit = (bits >> (n - 1 - rank)) & 1;
            if (spec) coef = (int)(int16_t)(coef + (coef < 0 ? -(bit << al) : (bit << al)));   // T.81 G.1.2.3
            else coef = (int)(int16_t)(coef | (int)(int16_t)(bit << al));                      // the reference (:1114)
        }
        dirty = true;
    }
}

}  // namespace

// KIND 1 = the refining AC scans only, KIND 0 = every other kind of scan, KIND 2 = all of them.  Two smaller kernels instead
// of one that holds all five code paths: the refining loop is scalar-register starved (23-42 spilled scalars in the combined
// kernel, 0-5 in the split ones); each launch's waves of the other kind leave at once.  The split costs a second launch
// whose kernels run one after the other, so the band-pipelined variant — whose launches are short and whose scans of all
// kinds are meant to run side by side — keeps the combined kernel.
template <bool BANDED, int KIND>
__global__ __launch_bounds__(256) void k_progressive_scan(const uint8_t *__restrict__ blob,
                                                          const DevProgSeg *__restrict__ segs, int n_segs,
                                                          const DevProgScan *__restrict__ scans,
                                                          const DevImage *__restrict__ images,
                                                          const DevHuff *__restrict__ huff, int16_t *__restrict__ coef,
                                                          int32_t *__restrict__ status, int spec_refine, int tr,
                                                          DevProgState *__restrict__ states, int step, int rows_per_band) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_lut[];   // [4 waves][3 tables][kLutSize]
    const bool spec = (spec_refine & 1) != 0;
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int seg_id = blockIdx.x * 4 + wave;
    if (seg_id >= n_segs) return;

    const DevProgSeg *sg = segs + seg_id;
    const DevProgScan *sc = scans + sg->scan;
    const DevImage *im = images + sc->image;
    uint16_t *my_lut = s_lut + (size_t)wave * 3 * kLutSize;
    const int ss = sc->ss, se = sc->se, al = sc->al;
    const bool refining = sc->ah != 0;
    const bool sequential = ss == 0 && se == 63;          // one component, DC + AC per block (baseline_dct_scan, :734-866)
    const bool is_dc = ss == 0 && !sequential;
    const int nsc = sc->n_comp;
    if (KIND != 2 && (KIND == 1) != (refining && !is_dc && !sequential)) return;
    // spec_refine bit 1: the first scans and the refining AC scans are walked elsewhere (progressive_fast.hip); what is left
    // here is DC refinement and the sequential scans
    if ((spec_refine & 2) && !sequential) return;
    const bool ac_refining = KIND == 1 ? true : (KIND == 0 ? false : refining);      // in the AC branch below

    // ---- which part of the scan this launch does.  The scans of an image are pipelined over bands of `rows_per_band`
    // frame MCU rows: launch number `step` lets a scan of dependency level L work on band step - L, so a refining scan
    // runs one band behind the scan it refines instead of waiting for all of it.  Between two of its bands a scan
    // keeps its bit reader, EOB run and DC predictors in `states`.
    // (Without BANDED the launch holds the segments of one dependency level only and each does its whole scan.)
    const int band = BANDED ? step - sc->level : 0;
    if (band < 0) return;
    int b_lo = sg->mcu0, b_hi = sg->mcu0 + sg->n_mcu;
    if (!BANDED) {
    } else if (sequential) {
        if (band != 0) return;                       // the whole scan at once (baseline scans of separate components depend on nothing)
    } else {
        const int v_scan = nsc == 1 ? im->comp_v[sc->comp[0]] : 1;   // block rows per frame MCU row
        const int64_t mpb = (int64_t)sc->mcu_count_h * rows_per_band * v_scan;
        const int64_t lo = (int64_t)band * mpb, hi = lo + mpb;
        b_lo = (int)max((int64_t)b_lo, lo);
        b_hi = (int)min((int64_t)b_hi, hi);
        if (b_lo >= b_hi) return;
    }
    const bool resume = BANDED && b_lo != sg->mcu0, finish = !BANDED || b_hi == sg->mcu0 + sg->n_mcu;
    // (read at the point of use when the scan is done in one piece: fewer scalars live through the symbol loops)
    #define m_lo (BANDED ? b_lo : sg->mcu0)
    #define m_hi (BANDED ? b_hi : sg->mcu0 + sg->n_mcu)
    DevProgState *st = states + seg_id;

    // tables of this scan: DC scans use one DC table per scan component, AC scans one AC table, sequential scans both
    const int n_tabs = sequential ? 2 : (is_dc ? (refining ? 0 : nsc) : 1);
    for (int t = 0; t < n_tabs; ++t) {
        const int gi = sequential ? (t == 0 ? sc->dc_tab[0] : sc->ac_tab[0]) : (is_dc ? sc->dc_tab[t] : sc->ac_tab[0]);
        reinterpret_cast<uint4 *>(my_lut + t * kLutSize)[lane] = reinterpret_cast<const uint4 *>(huff[gi].lut)[lane];
    }

    BitReader br;
    br.init(blob, sg->begin, sg->len, lane);
    int st_eobrun = 0, st_pred0 = 0, st_pred1 = 0, st_pred2 = 0;
    if (resume) {
        if (st->err != 0 || st->mcu_next != m_lo) return;        // the scan failed earlier (its status is set)
        br.restore(st->pos, st->bb, st->bc, st->pad);
        st_eobrun = st->eobrun; st_pred0 = st->pred[0]; st_pred1 = st->pred[1]; st_pred2 = st->pred[2];
    }

    // frame geometry (interleaved block order of the coefficient store)
    const int bpm = im->blocks_per_mcu, fmx = im->mcu_count_h;
    int16_t *cbase = coef + im->block_off * 64;
    auto block_ptr = [&](int c, int bx, int by) -> int16_t * {
        const int h = im->comp_h[c], v = im->comp_v[c], first = im->comp_first[c];   // (a lone component: 1 x 1, block 0)
        const int mx = bx / h, my = by / v;
        return cbase + ((int64_t)(my * fmx + mx) * bpm + first + (by - my * v) * h + (bx - mx * h)) * 64;
    };
    const int smh = sc->mcu_count_h;
    // tr: the plan keeps blocks transposed ([u][v]) for the row-major stage 2
    auto store_pos = [&](int z) { const int n = c_nat_of_zz_p[z]; return tr ? ((n & 7) << 3 | n >> 3) : n; };
    const int nat = store_pos(lane);
    int err = 0;
    // what a scan carries from band to band is written by the branch that owns it (kept local there: as function-wide
    // variables they cost the refining loop a dozen scalar moves per block)
    auto save_state = [&](int eob, int p0, int p1, int p2) {
        if (BANDED && !finish && lane == 0) {
            st->pos = br.pos; st->bb = br.bb; st->bc = br.bc; st->pad = br.pad;
            st->eobrun = eob; st->pred[0] = p0; st->pred[1] = p1; st->pred[2] = p2;
            st->err = err; st->mcu_next = m_hi;
        }
    };

    if (KIND != 1 && sequential) {
        // ------------------------------------------------------------ one component of a non-interleaved baseline file:
        // the scan's MCU is one 8x8 block, blocks in raster order of the component (:612-619, :771-866)
        const int c = sc->comp[0];
        int pred = 0;
        for (int m = m_lo; m < m_hi && !err; ++m) {
            const int by = m / smh, bx = m - by * smh;
            int16_t *p = block_ptr(c, bx, by);
            br.refill();
            const int s = decode_symbol(br, my_lut, huff + sc->dc_tab[0]);
            if (s < 0 || s > 16) { err = MJ_ST_BAD_CODE; break; }
            int diff = 0;
            if (s > 0) diff = extend(br.take(s), s);
            pred = (int)(int16_t)(diff + pred);                                          // (:818-820)
            if (lane == 0) p[0] = (int16_t)pred;
            int k = 1;
            while (k < 64) {                                                             // (:834-866)
                br.refill();
                const int hv = decode_symbol(br, my_lut + kLutSize, huff + sc->ac_tab[0]);
                if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                if (hv == 0) break;                                                      // end of block (:849)
                k += hv >> 4;
                if (k >= 64) break;                                                      // (:855-856): value bits stay unread
                const int n = hv & 15;
                if (n > 0) {
                    const int val = extend(br.take(n), n);
                    if (lane == 0) p[store_pos(k)] = (int16_t)val;
                }
                ++k;
            }
        }
    } else if (KIND != 1 && is_dc) {
        // ------------------------------------------------------------ DC scans (:974-1057)
        int pred0 = st_pred0, pred1 = st_pred1, pred2 = st_pred2;
        if (refining) {
            // One bit per block (:1038), blocks in scan order: lane i takes the i-th block of a group of 64, so the
            // read-modify-write of 64 DC values is one load and one store instead of 64 dependent round trips.
            int bps = 0;                                   // blocks per MCU of this scan
            for (int i = 0; i < nsc; ++i) { const int c = sc->comp[i]; bps += nsc > 1 ? im->comp_h[c] * im->comp_v[c] : 1; }
            const int total = (m_hi - sg->mcu0) * bps;
            for (int t0 = (m_lo - sg->mcu0) * bps; t0 < total; t0 += 64) {
                const int n = min(64, total - t0);
                br.refill();
                const uint32_t w0 = take32(br, min(32, n));
                br.refill();
                const uint32_t w1 = take32(br, max(0, n - 32));
                const int t = t0 + lane;
                if (lane < n) {
                    const int m = sg->mcu0 + t / bps;
                    int j = t % bps, c = 0, r = 0;
                    for (int i = 0; i < nsc; ++i) {               // which scan component / block of the MCU
                        const int ci = sc->comp[i];
                        const int cnt = nsc > 1 ? im->comp_h[ci] * im->comp_v[ci] : 1;
                        if (j < cnt) { c = ci; r = j; break; }
                        j -= cnt;
                    }
                    const int mcy = m / smh, mcx = m - mcy * smh;
                    const int h = nsc > 1 ? im->comp_h[c] : 1, v = nsc > 1 ? im->comp_v[c] : 1;
                    int16_t *p = block_ptr(c, mcx * h + r % h, mcy * v + r / h);
                    const int nlo = min(32, n);
                    const int bit = lane < 32 ? (w0 >> (nlo - 1 - lane)) & 1 : (w1 >> (n - 32 - 1 - (lane - 32))) & 1;
                    p[0] = (int16_t)(p[0] | (int16_t)(bit << al));
                }
            }
        } else
        for (int m = m_lo; m < m_hi && !err; ++m) {
            const int mcy = m / smh, mcx = m - mcy * smh;
            for (int i = 0; i < nsc && !err; ++i) {
                const int c = sc->comp[i];
                const int h = nsc > 1 ? im->comp_h[c] : 1, v = nsc > 1 ? im->comp_v[c] : 1;
                for (int r = 0; r < h * v; ++r) {
                    const int bx = mcx * h + r % h, by = mcy * v + r / h;
                    int16_t *p = block_ptr(c, bx, by);
                    br.refill();
                    if (!refining) {
                        int s = decode_symbol(br, my_lut + i * kLutSize, huff + sc->dc_tab[i]);
                        if (s < 0 || s > 16) { err = MJ_ST_BAD_CODE; break; }
                        int diff = 0;
                        if (s > 0) diff = extend(br.take(s), s);
                        const int pred = i == 0 ? pred0 : (i == 1 ? pred1 : pred2);
                        const int dcv = (int)(int16_t)(diff + pred);
                        if (i == 0) pred0 = dcv; else if (i == 1) pred1 = dcv; else pred2 = dcv;
                        if (lane == 0) p[0] = (int16_t)(dcv << al);                       // (:1029)
                    } else {
                        const int bit = (int)br.take(1);
                        if (lane == 0) p[0] = (int16_t)(p[0] | (int16_t)(bit << al));     // (:1038)
                    }
                }
            }
        }
        save_state(0, pred0, pred1, pred2);
    } else {
        // ------------------------------------------------------------ AC scans (:1060-1298)
        const int c = sc->comp[0];
        const uint16_t *lut = my_lut;
        const DevHuff *tab = huff + sc->ac_tab[0];
        const uint64_t band = bits_range(ss, se + 1);
        const int m_end = m_hi;
        int eobrun = st_eobrun;
        // refining scans read every block before they touch it: the next block's coefficients are requested while
        // this one is being worked on (a dependent load per block was most of a refining scan's time)
        int cf_next = 0;
        // block coordinates are stepped, not divided out of m for every block (two integer divisions per block were
        // some 4 % of a refining scan)
        const int by_lo = m_lo / smh, bx_lo = m_lo - by_lo * smh;
        if (ac_refining && m_lo < m_hi) cf_next = block_ptr(c, bx_lo, by_lo)[nat];
        // (only in the band-pipelined variant: in the other one the same change made the compiler's loop 5 % slower)
        for (int m = m_lo, bx = bx_lo, by = by_lo; m < m_end && !err; ++m, bx = (bx + 1 == smh ? 0 : bx + 1), by += (bx == 0)) {
            if (!BANDED) { by = m / smh; bx = m - by * smh; }
            int16_t *p = block_ptr(c, bx, by);
            const int cf_cur = cf_next;
            if (ac_refining && m + 1 < m_end) {
                int bx1 = bx + 1 == smh ? 0 : bx + 1, by1 = by + (bx1 == 0);
                if (!BANDED) { by1 = (m + 1) / smh; bx1 = (m + 1) - by1 * smh; }
                cf_next = block_ptr(c, bx1, by1)[nat];
            }
            if (!ac_refining) {
                // -------- first scan of the band: only writes (:1177-1179, :1225, :1248-1250)
                if (eobrun > 0) { --eobrun; continue; }
                int k = ss;
                while (k <= se) {
                    br.refill();
                    const int hv = decode_symbol(br, lut, tab);
                    if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                    const int r = hv >> 4, s = hv & 15;
                    if (hv == 0) { eobrun = 1; break; }
                    if (s == 0 && r != 15) { eobrun = (1 << r) + (int)take32(br, r); break; }
                    k += (hv == 0xF0) ? 16 : r;
                    if (s > 0) {
                        if (k > 63) { err = MJ_ST_OVERRUN; break; }
                        const int val = extend(br.take(s), s);
                        if (lane == 0) p[store_pos(k)] = (int16_t)(val << al);
                        ++k;
                    }
                }
                if (eobrun > 0) --eobrun;          // the band that raised the run counts as its first one
            } else {
                // -------- refining scan: coefficients of the block in the lanes (lane = zig-zag index)
                int cf = cf_cur;
                bool dirty = false;
                if (eobrun > 0) {                   // inside an EOB run: every non-zero coefficient of the band gets a bit
                    const uint64_t nz = __ballot(cf != 0);
                    refine_queue(br, nz & band, al, lane, cf, dirty, spec);
                    --eobrun;
                } else {
                    int k = ss;
                    // which coefficients are non-zero: one ballot per block, then kept up to date in scalar registers (a
                    // correction never clears a coefficient, a newly placed one is +-1 << al)
                    uint64_t nz = __ballot(cf != 0);
                    while (k <= se) {
                        br.refill();
                        const int hv = decode_symbol(br, lut, tab);
                        if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                        const int r = hv >> 4, s = hv & 15;
                        if (hv == 0) { eobrun = 1; break; }
                        if (s == 0 && r != 15) { eobrun = (1 << r) + (int)take32(br, r); break; }
                        int zr = (hv == 0xF0) ? 16 : r;
                        // pass zr zero coefficients from k on (:1184-1193)
                        uint64_t zeros = ~nz & bits_from(k);
                        int pnext = k;
                        if (zr > 0) {
                            for (int i = 1; i < zr; ++i) zeros &= zeros - 1;
                            if (zeros == 0) { err = MJ_ST_OVERRUN; break; }
                            pnext = __builtin_ctzll(zeros) + 1;
                        }
                        uint64_t queue = nz & bits_range(k, pnext);
                        k = pnext;
                        if (s > 0) {
                            br.refill();
                            const int val = extend(br.take(s), s);              // value bits precede the corrections (:1202)
                            const uint64_t z2 = ~nz & bits_from(k);             // next zero position (:1212-1215)
                            if (z2 == 0) { err = MJ_ST_OVERRUN; break; }
                            const int pz = __builtin_ctzll(z2);
                            queue |= nz & bits_range(k, pz);
                            k = pz;
                            if (lane == k) cf = (int)(int16_t)(val << al);      // (:1225)
                            nz |= (uint64_t)((int16_t)(val << al) != 0) << k;
                            dirty = true;
                            ++k;
                        }
                        refine_queue(br, queue, al, lane, cf, dirty, spec);     // (:1231-1232)
                    }
                    if (!err && eobrun > 0) {       // rest of this band, then the run continues in the next blocks
                        refine_queue(br, nz & bits_range(k, se + 1), al, lane, cf, dirty, spec);
                        --eobrun;
                    }
                }
                // only this scan's band is written back: other scans of the same dependency level may be updating
                // other coefficients of the block at the same time
                if (dirty && lane >= ss && lane <= se) p[nat] = (int16_t)cf;
            }
        }
        save_state(eobrun, 0, 0, 0);
    }

    if (!err && finish) {
        if (br.pad > 0 && br.bc < br.pad) err = MJ_ST_OVERRUN;
        else if (!sg->last && (((br.bc - br.pad) >> 3) > 0 || br.pos < br.end)) err = MJ_ST_DESYNC;
    }
    if (err && lane == 0) atomicMax(status + sc->image, err);
}

#undef m_lo
#undef m_hi

hipError_t launch_progressive_scan(hipStream_t stream, const uint8_t *blob, const DevProgSeg *segs, int n_segs,
                                   const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                   int16_t *coef, int32_t *status, int spec_refine, int transposed, DevProgState *states,
                                   int step, int rows_per_band) {
    if (n_segs == 0) return hipSuccess;
    const int blocks = (n_segs + 3) / 4;
    const size_t lds = (size_t)4 * 3 * kLutSize * sizeof(uint16_t);
    if (rows_per_band > 0) {
        hipLaunchKernelGGL((k_progressive_scan<true, 2>), dim3((unsigned)blocks), dim3(256), lds, stream, blob, segs, n_segs,
                           scans, images, huff, coef, status, spec_refine, transposed, states, step, rows_per_band);
    } else {        // every scan in one piece: `step` is the dependency level whose scans run
        hipLaunchKernelGGL((k_progressive_scan<false, 0>), dim3((unsigned)blocks), dim3(256), lds, stream, blob, segs, n_segs,
                           scans, images, huff, coef, status, spec_refine, transposed, states, step, 0);
        if (rows_per_band == 0)      // (-1: the refining AC scans are progressive_fast.hip's)
            hipLaunchKernelGGL((k_progressive_scan<false, 1>), dim3((unsigned)blocks), dim3(256), lds, stream, blob, segs, n_segs,
                               scans, images, huff, coef, status, spec_refine, transposed, states, step, 0);
    }
    return hipGetLastError();
}

}  // namespace mj
