// Stage 1 of progressive files on the stage-0 stream: one wavefront walks one restart segment of one scan.
//
// ---- the refining AC scans (Ah > 0, Ss > 0) — jpeg_decoder.py:1122-1298 with the correction queue of :1100-1115.
// These scans are most of a progressive file's entropy-coded bytes (the last luma
// scan alone is half of a libjpeg-default file) and they cannot be cut into independent pieces: how many correction
// bits follow a symbol depends on which coefficients of the block earlier scans left non-zero.  What is left to
// optimise is the serial chain per symbol; one wavefront walks one restart segment of one scan, and a lone
// wavefront issues one instruction every ~5 cycles whatever the instruction is — and fetches 32 bytes of them
// in ~15 — so the chain is counted in instructions and in bytes of code (round 4: one copy of the block's code
// instead of four unrolled ones, loops aligned to their fetch windows).
// progressive.hip's general walk spends ~170 instructions per symbol (bit-buffer refills with the 0xFF
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
//     history-non-zero coefficients passed on the way — the correction bits to skip — is a second v_readlane into a
//     table that follows from the first by arithmetic (position - Ss - ordinal).  The masks of the block's history do not change while the block is walked
//     (a coefficient placed by this scan lies behind everything later symbols look at);
//   * correction bits are not read when their symbol is decoded: every lane remembers where its bit will be
//     (one v_cmp/v_cndmask per symbol) and the whole block's corrections are fetched from the ring and applied at
//     once when the block ends; the new coefficients go into the lanes with v_writelane.
//
// 22 instructions per coefficient symbol (round 2: 28); blocks inside an end-of-band run cost ~40 in all.
//   * Split scans (round 4): where the chip has wave slots to spare, a refining scan is walked twice — a scout that follows
//     the symbol chain and the correction counts alone (14 instructions, 64 bytes per symbol) and leaves its position at the
//     first MCU of every part of a band, and one launch behind it the parts, walked side by side by the placing walk.
//
// ---- the first scans of a band (Ah = 0): DC scans (:974-1029) and AC scans (:1122-1179, :1236-1250) only write.
//   * AC: the block's new coefficients are collected in the lanes (lane = zig-zag index, v_writelane) together with a mask
//     of the positions written, and stored once per block; blocks inside an end-of-band run are skipped untouched;
//   * DC: the predictor chain (:1018-1020) is the serial part; the values of 64 consecutive blocks of the scan are
//     collected in the lanes and stored together.  An interleaved scan uses one Huffman table per component, so each
//     window of bit offsets is looked up in all of them and the walk takes the entry of the component whose turn it is.
//   12 instructions per symbol (round 4; 20 before).
//
//   * DC refinement (Ah > 0, Ss = 0) is one bit per block in scan order: block t reads bit t of the stream, no chain.
//
// progressive.hip's general walk keeps the sequential scans of non-interleaved baseline files.
//
// ---- launches.  Scans of one image depend on each other (a refining scan needs what it refines), so they are grouped into
// dependency levels.  Either one launch per level over that level's segments, or — BANDED — the scans of an image are
// pipelined over bands of frame MCU rows: launch number `step` lets a scan of level L work on band step - L, so a
// refining scan runs one band behind the scan it refines instead of waiting for all of it, and the serial chains of an
// image's ten scans overlap.  Between two of its bands a scan keeps its bit position, end-of-band run and DC predictors in
// a DevProgState.
#include "mijpeg_internal.h"
#include "prog_stream.h"
#ifdef MJ_DIAGNOSTIC
#include <algorithm>
#include <vector>
#endif

namespace mj {

using namespace progstream;

#ifdef MJ_DIAGNOSTIC   // separate diagnostic build only (make DIAG=1): where the final luma refinement's cycles go
__device__ unsigned long long g_dbg_prog[11];     // 7 phase sums, waves, walked blocks, blocks inside EOB runs, coefficients placed
void dbg_prog_report() {
    unsigned long long h[11], z[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_prog), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_prog), z, sizeof(z));
    unsigned long long tot = 0;
    for (int i = 0; i < 7; ++i) tot += h[i];
    if (tot == 0) return;
    static const char *nm[7] = {"block prologue", "symbol loop (asm)", "window look-ups", "EOB / long codes", "corrections + store", "blocks in EOB runs", "between blocks"};
    fprintf(stderr, "[diag] refining walk of the Ah=1 luma scans, %llu waves: ", h[7]);
    for (int i = 0; i < 7; ++i) fprintf(stderr, "%s %.1f%%  ", nm[i], 100.0 * (double)h[i] / (double)tot);
    fprintf(stderr, "(%.1f Mcycles per wave)\n", (double)tot / (double)(h[7] ? h[7] : 1) / 1e6);
    fprintf(stderr, "[diag]   %llu walked blocks (%.2f coefficients placed per block, %.0f cycles per block all phases but EOB-run blocks), %llu blocks inside EOB runs (%.0f cycles each)\n",
            h[8], (double)h[10] / (double)(h[8] ? h[8] : 1), (double)(tot - h[5]) / (double)(h[8] ? h[8] : 1), h[9], (double)h[5] / (double)(h[9] ? h[9] : 1));
}
// when does every wave of one band launch (MJ_DEBUG_PROG_STEP) finish, and what was it walking?
__device__ unsigned long long g_dbg_prog_waves[32768 * 3];
void dbg_prog_waves_report() {
    std::vector<unsigned long long> t(32768 * 3), z(32768 * 3, 0);
    (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_dbg_prog_waves), t.size() * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_prog_waves), z.data(), z.size() * 8);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < t.size(); i += 3) if (t[i] && t[i] < t0) t0 = t[i];
    if (t0 == ~0ull) return;
    // kinds: 0 DC first, 1 AC first, 2 AC refine luma (whole walk or scout), 3 AC refine chroma, 4 a part of a split scan's band
    std::vector<double> en[5], st[5], all;
    for (size_t i = 0; i < t.size(); i += 3) if (t[i]) {
        const int kind = (int)(t[i + 2] & 7);
        en[kind].push_back((double)(t[i + 1] - t0) * 0.01); st[kind].push_back((double)(t[i] - t0) * 0.01); all.push_back(en[kind].back());
    }
    std::sort(all.begin(), all.end());
    fprintf(stderr, "[diag prog] %zu waves that walked a band in this launch; end us: p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f\n", all.size(),
            all[(size_t)(0.1 * (all.size() - 1))], all[all.size() / 2], all[(size_t)(0.9 * (all.size() - 1))], all[(size_t)(0.99 * (all.size() - 1))], all.back());
    static const char *nm[5] = {"DC first", "AC first", "AC refine luma", "AC refine chroma", "parts that place"};
    for (int k = 0; k < 5; ++k) if (!en[k].empty()) {
        std::sort(en[k].begin(), en[k].end()); std::sort(st[k].begin(), st[k].end());
        double dur = 0; for (size_t i = 0; i < en[k].size(); ++i) dur += en[k][i];
        double sst = 0; for (size_t i = 0; i < st[k].size(); ++i) sst += st[k][i];
        fprintf(stderr, "[diag prog]   %-16s %6zu waves: start median %.0f max %.0f; end median %.0f p90 %.0f max %.0f; mean walk %.0f us\n", nm[k], en[k].size(),
                st[k][st[k].size() / 2], st[k].back(), en[k][en[k].size() / 2], en[k][(size_t)(0.9 * (en[k].size() - 1))], en[k].back(), (dur - sst) / en[k].size());
    }
}
#define PSTAMP(i) do { if (dbg_on) { uint64_t s_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s_) :: "memory"); dacc[i] += s_ - dlast; dlast = s_; } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

namespace {

// Where a symbol loop starts matters: the walks are lone wavefronts bound by instruction fetch, and the scout's 72-byte loop runs
// 7 % faster from byte 16 of a 32-byte window than from byte 0, 8 or 24 (16 x 1080p: 52.1 ms against 56.1 / 55.8 / 55.5;
// period 32 bytes).  Before the loops were aligned their speed moved by that much with whatever code happened to precede them.
// (MJ_SKEW_*: s_nop's behind the alignment; the placing loop and the first scans' loop are indifferent.)
#ifndef MJ_LOOP_ALIGN_LOG2
#define MJ_LOOP_ALIGN_LOG2 6
#endif
#define MJ_STR2(x) #x
#define MJ_STR(x) MJ_STR2(x)
#define MJ_LOOP_ALIGN MJ_STR(MJ_LOOP_ALIGN_LOG2)
#ifndef MJ_SCOUT_UNROLL
#define MJ_SCOUT_UNROLL 1
#endif
#ifndef MJ_SKEW_R
#define MJ_SKEW_R 4
#endif
#ifndef MJ_SKEW_C
#define MJ_SKEW_C 0
#endif
#ifndef MJ_SKEW_F
#define MJ_SKEW_F 4
#endif
#define MJ_LOOP_PAD(n) ".rept " MJ_STR(n) "\n s_nop 0\n .endr\n"

// Everything a walk needs; wave-uniform
struct Walk {
    const DevProgSeg *sg;
    const DevProgScan *sc;
    const DevImage *im;
    const DevHuff *huff;
    int16_t *cbase;
    int m_lo, m_hi;           // the scan MCUs this launch does
    int ss, se, al, tr, lane;
    bool spec;
    // carried from band to band
    int eobrun, pred0, pred1, pred2;
    int err;
    // a split scan's scout: where it stands at the first MCU of each part of the band (m_lo + i * sub_q) goes to sub_out[i]
    DevProgSub *sub_out;
    int sub_q;
};

// ------------------------------------------------------------------------------------------------ DC, first scan (:974-1029)
__device__ __forceinline__ void walk_dc_first(Walk &k, Stream &st, const uint16_t *lut /* [3][512] */) {
    const DevProgScan *sc = k.sc;
    const int lane = k.lane, al = k.al, nsc = sc->n_comp;
    const int bpm = k.im->blocks_per_mcu;
    const DevImage *im = k.im;
    // blocks per MCU of each scan component (:980-1003): an interleaved scan covers the frame's MCUs, a single-component
    // one (never of a subsampled-luma component: api.hip) one block per MCU
    const int cA = sc->comp[0], cB = nsc > 1 ? sc->comp[1] : 0, cC = nsc > 2 ? sc->comp[2] : 0;
    const int nA = nsc > 1 ? im->comp_h[cA] * im->comp_v[cA] : 1, nB = nsc > 1 ? im->comp_h[cB] * im->comp_v[cB] : 0;
    const int nC = nsc > 2 ? im->comp_h[cC] * im->comp_v[cC] : 0;
    const int bps = nA + nB + nC;
    auto first_of = [&](int c) { return (int)im->comp_first[c]; };      // a component's first block in the frame's MCU
    uint32_t veA0, veB0 = 0, veC0 = 0, vw0, veA1, veB1 = 0, veC1 = 0, vw1;
    auto lookup64 = [&](int g, uint32_t &va, uint32_t &vb, uint32_t &vc, uint32_t &vw) {
        const uint32_t w = st.bits_at(g + lane);
        const uint32_t i9 = w >> (32 - kProgDcLutBits);
        vw = w;
        const uint32_t ea = lut[i9];
        va = dc_entry(w, (int)(ea >> 8), (int)(ea & 255u));
        if (nsc > 1) { const uint32_t eb = lut[kPDcLut + i9]; vb = dc_entry(w, (int)(eb >> 8), (int)(eb & 255u)); }
        if (nsc > 2) { const uint32_t ec = lut[2 * kPDcLut + i9]; vc = dc_entry(w, (int)(ec >> 8), (int)(ec & 255u)); }
    };
    int gbase = st.bp;
    lookup64(gbase, veA0, veB0, veC0, vw0);
    lookup64(gbase + 64, veA1, veB1, veC1, vw1);
    int pred0 = k.pred0, pred1 = k.pred1, pred2 = k.pred2;
    const int mcu0 = k.sg->mcu0;
    const int t_lo = (k.m_lo - mcu0) * bps, t_hi = (k.m_hi - mcu0) * bps;      // blocks of the segment, in scan order
    int vdc = 0;
    int j = 0;                                                                // block of the MCU
    int err = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        if (((t - t_lo) & 15) == 0) st.top_up();
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
        const int ci = j < nA ? 0 : (j < nA + nB ? 1 : 2);                    // scan component of this block
        uint32_t e = ci == 0 ? rdl(veA0, off) : (ci == 1 ? rdl(veB0, off) : rdl(veC0, off));
        if (__builtin_expect((e & 3u) != 0u, 0)) {                            // a code longer than the LUT's index, or none
            const uint32_t w = rdl(vw0, off);
            int len, s;
            long_code(w, k.huff + sc->dc_tab[ci], kProgDcLutBits + 1, len, s);
            e = dc_entry(w, len, s);
            if (e & 3u) { err = MJ_ST_BAD_CODE; break; }
        }
        st.bp += (int)((e >> 6) & 63u);
        const int pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
        const int dcv = (int)(int16_t)((int)(e >> 16) + pred);              // (:1018-1020)
        if (ci == 0) pred0 = dcv; else if (ci == 1) pred1 = dcv; else pred2 = dcv;
        write_lane(vdc, (int)(int16_t)(dcv << al), (t - t_lo) & 63);        // (:1029)
        if (++j == bps) j = 0;
        if (((t - t_lo) & 63) == 63 || t + 1 == t_hi) {                      // 64 blocks' values at once
            const int tl = t_lo + ((t - t_lo) & ~63) + lane;                  // this lane's block of the segment
            if (tl <= t) {
                const int mm = tl / bps, jj = tl - mm * bps;
                const int cc = jj < nA ? cA : (jj < nA + nB ? cB : cC);
                const int rr = jj < nA ? jj : (jj < nA + nB ? jj - nA : jj - nA - nB);
                k.cbase[((int64_t)(mcu0 + mm) * bpm + first_of(cc) + rr) * 64] = (int16_t)vdc;
            }
        }
    }
    k.pred0 = pred0; k.pred1 = pred1; k.pred2 = pred2;
    k.err = err;
}

// ------------------------------------------------------------------------------------------------ DC, refining scan (:1036-1038)
// One bit per block, blocks in scan order: block t of the segment reads bit t of its stream — no chain at all.  64 blocks per step,
// the bit straight from the stage-0 stream in global memory.  (Until round 4 the general walk did this from the raw bytes, in a
// launch of its own behind every launch of this kernel: 70 launches x 40 us per 1024 files, all of it on the critical path.)
__device__ __forceinline__ int walk_dc_refine(Walk &k, const uint32_t *sw, int n_dw) {
    const DevProgScan *sc = k.sc;
    const DevImage *im = k.im;
    const int lane = k.lane, al = k.al, nsc = sc->n_comp;
    const int bpm = im->blocks_per_mcu;
    const int cA = sc->comp[0], cB = nsc > 1 ? sc->comp[1] : 0, cC = nsc > 2 ? sc->comp[2] : 0;
    const int nA = nsc > 1 ? im->comp_h[cA] * im->comp_v[cA] : 1, nB = nsc > 1 ? im->comp_h[cB] * im->comp_v[cB] : 0;
    const int nC = nsc > 2 ? im->comp_h[cC] * im->comp_v[cC] : 0;
    const int bps = nA + nB + nC;
    const int mcu0 = k.sg->mcu0;
    const int t_lo = (k.m_lo - mcu0) * bps, t_hi = (k.m_hi - mcu0) * bps;      // blocks of the segment, in scan order = bits of its stream
    for (int t = t_lo + lane; t < t_hi; t += 64) {
        const int d = t >> 5;
        const uint32_t w = (uint32_t)d < (uint32_t)n_dw ? sw[d] : 0u;            // behind the segment's end the stream reads as zeros
        const int bit = (int)((w >> (31 - (t & 31))) & 1u);
        const int mm = t / bps, jj = t - mm * bps;
        const int cc = jj < nA ? cA : (jj < nA + nB ? cB : cC);
        const int rr = jj < nA ? jj : (jj < nA + nB ? jj - nA : jj - nA - nB);
        int16_t *p = k.cbase + ((int64_t)(mcu0 + mm) * bpm + im->comp_first[cc] + rr) * 64;
        p[0] = (int16_t)(p[0] | (int16_t)(bit << al));                           // (:1038)
    }
    return t_hi;                                                                 // the bit position behind the band
}

// ------------------------------------------------------------------------- AC, first scan of the band (:1122-1179, :1236-1250)
__device__ __forceinline__ void walk_ac_first(Walk &k, Stream &st, const uint16_t *lut) {
    const DevProgScan *sc = k.sc;
    const int lane = k.lane, al = k.al, ss = k.ss, se = k.se;
    const int bpm = k.im->blocks_per_mcu, fmx = k.im->mcu_count_h;
    const int c = sc->comp[0];
    const DevHuff *tab = k.huff + sc->ac_tab[0];
    const int h = k.im->comp_h[c], v = k.im->comp_v[c], first = k.im->comp_first[c];   // (h, v: 1, 2 or 4 here — api.hip sends other factors to the general walk)
    const int smh = sc->mcu_count_h;
    const int nz_nat = c_nat_of_zz_ps[lane];
    const int nat = k.tr ? ((nz_nat & 7) << 3 | nz_nat >> 3) : nz_nat;        // tr: blocks are kept [u][v] for the row-major stage 2
    const int lh = h == 4 ? 2 : h - 1, lv = v == 4 ? 2 : v - 1;               // h, v are 1, 2 or 4
    int by = k.m_lo / smh, bx = k.m_lo - by * smh;
    AcWindows<false> win;
    win.start(st, lut, al, lane, st.bp);
    int eobrun = k.eobrun, err = 0;
    // first block of block row `by` (the part of a block's number that only changes with the row)
    auto row_base = [&](int y) { const int my = y >> lv; return my * fmx * bpm + first + ((y - (my << lv)) << lh); };
    int rbase = row_base(by);
    for (int m = k.m_lo; m < k.m_hi && !err;) {
        if (eobrun > 0) {                                  // blocks inside an end-of-band run hold nothing of this band: skipped in one step
            const int n = min(eobrun, k.m_hi - m);
            eobrun -= n; m += n; bx += n;
            if (bx >= smh) {
                do { bx -= smh; ++by; } while (bx >= smh);
                rbase = row_base(by);
            }
            continue;
        }
        {
            st.top_up();
            int cf = 0;                                    // the lanes take whole entries (value in the upper half) until the block ends
            int kk = ss;
            for (;;) {
                uint32_t e;
                int code, off;
                for (;;) {                                 // (the common way round on its own, as in the refining walk)
                // The run of coefficient symbols inside the current window, hand-scheduled like the refining walk's and with the same
                // trims (round 4): the entry's run field holds r + 1 and the loop keeps the last position taken; the whole entry goes
                // into the lane; positions are tested against Se once, in front — what is no plain coefficient lands past it too,
                // and so does the symbol behind a full band; two symbols per turn with the entry registers swapping roles.
                // 12 instructions per symbol (20 before).  Leaves with code 0: the band is full;  1: the next symbol starts behind
                // the window;  2: ZRL, or entry not in the LUT;  3: a run that ends behind the band;  4: end of band, eobrun set
                // (the run counts this block, :1160-1166).
                uint32_t e2;
                int t0, kk1 = kk - 1;
                asm volatile(
                    "s_sub_u32 %[off], %[bp], %[gbase]\n\t"
                    "s_cmp_gt_u32 %[off], 63\n\t"
                    "s_cbranch_scc1 Lfwin%=\n\t"
                    "v_readlane_b32 %[e], %[ve0], %[off]\n"
                    ".p2align " MJ_LOOP_ALIGN "\n" MJ_LOOP_PAD(MJ_SKEW_F)
                    "Lfsym%=:\n\t"
#define MJ_FIRST_SYMBOL(E, E2, OVER) \
                    "s_bfe_u32 %[t0], %[" E "], 0x80002\n\t"      /* run + 1 + 64 * class */ \
                    "s_add_u32 %[kk1], %[kk1], %[t0]\n\t"         \
                    "s_cmp_gt_u32 %[kk1], %[se]\n\t"              \
                    "s_cbranch_scc1 " OVER "%=\n\t"               \
                    "s_bfe_u32 %[t0], %[" E "], 0x5000b\n\t"      \
                    "s_add_u32 %[bp], %[bp], %[t0]\n\t"           \
                    "s_sub_u32 %[off], %[bp], %[gbase]\n\t"       \
                    "v_readlane_b32 %[" E2 "], %[ve0], %[off]\n\t" \
                    "s_mov_b32 m0, %[kk1]\n\t"                    \
                    "v_writelane_b32 %[cf], %[" E "], m0\n\t"     /* (:1248-1250) */ \
                    "s_cmp_le_u32 %[off], 63\n\t"
                    MJ_FIRST_SYMBOL("e", "e2", "Lfover")
                    "s_cbranch_scc0 Lfwin%=\n\t"
                    MJ_FIRST_SYMBOL("e2", "e", "Lfover2")
                    "s_cbranch_scc1 Lfsym%=\n"
                    "Lfwin%=:\n\t"
                    "s_mov_b32 %[code], 1\n\t"
                    "s_branch Lfend%=\n"
                    "Lfdone%=:\n\t"
                    "s_mov_b32 %[code], 0\n\t"
                    "s_branch Lfend%=\n"
                    "Lfover2%=:\n\t"
                    "s_mov_b32 %[e], %[e2]\n"
                    "Lfover%=:\n\t"
                    "s_sub_u32 %[kk1], %[kk1], %[t0]\n\t"        // (not taken: the position goes back)
                    "s_cmp_ge_u32 %[kk1], %[se]\n\t"             // the band's last position is taken: `e` is the next block's symbol
                    "s_cbranch_scc1 Lfdone%=\n\t"
                    "s_and_b32 %[code], %[e], 3\n\t"             // a special entry, or a run that ends behind the band
                    "s_cbranch_scc0 Lfrun%=\n\t"
                    "s_cmp_eq_u32 %[code], 2\n\t"                // EOBn found in the LUT
                    "s_cbranch_scc0 Lfother%=\n\t"
                    "s_lshr_b32 %[eob], %[e], 16\n\t"
                    "s_sub_u32 %[eob], %[eob], 1\n\t"
                    "s_bfe_u32 %[t0], %[e], 0x5000b\n\t"
                    "s_add_u32 %[bp], %[bp], %[t0]\n\t"
                    "s_mov_b32 %[code], 4\n\t"
                    "s_branch Lfend%=\n"
                    "Lfother%=:\n\t"
                    "s_mov_b32 %[code], 2\n\t"
                    "s_branch Lfend%=\n"
                    "Lfrun%=:\n\t"
                    "s_mov_b32 %[code], 3\n"
                    "Lfend%=:"
                    : [e] "=&s"(e), [e2] "=&s"(e2), [code] "=&s"(code), [t0] "=&s"(t0), [off] "=&s"(off), [bp] "+s"(st.bp), [kk1] "+s"(kk1),
                      [cf] "+v"(cf), [eob] "+s"(eobrun)
                    : [gbase] "s"(win.gbase), [ve0] "v"(win.ve0), [se] "s"(se)
                    : "scc", "m0");
                e = (uint32_t)rfl((int)e); code = rfl(code); off = rfl(off); st.bp = rfl(st.bp); kk = rfl(kk1) + 1; eobrun = rfl(eobrun);
                if (code != 1) break;
                win.move_to(st, lut, al, lane, off);
                }
                if (code == 0 || code == 4) break;
                // (code 3 — a plain coefficient whose run ends behind the band — goes through the single step below: placed up to
                // position 63 as the reference does, refused behind it)
                if ((e & 3u) == 3u) {                      // a code longer than the LUT's index (rare) or no code at all
                    const uint32_t w = rdl(win.vw0, off);
                    int len, hv;
                    long_code(w, tab, kProgLutBits + 1, len, hv);
                    e = ac_entry<false>(w, len, hv, al);
                    if (len == 0) { err = MJ_ST_BAD_CODE; break; }
                }
                if (e & 2u) {                              // end of band: the run counts this block (:1160-1166)
                    eobrun = (int)(e >> 16) - 1;
                    st.bp += (int)((e >> 11) & 31u);
                    break;
                }
                if (e & 1u) {                              // ZRL: sixteen zeros (:1170)
                    kk += 16;
                    st.bp += (int)((e >> 11) & 31u);
                    if (kk > se) break;
                    continue;
                }
                // (a long code's coefficient, or one behind the band)
                kk += (int)((e >> 2) & 31u) - 1;           // (the field holds r + 1)
                if (kk > 63) { err = MJ_ST_OVERRUN; break; }
                write_lane(cf, (int)e, kk);                // (:1248-1250; the whole entry, as the loop leaves it)
                st.bp += (int)((e >> 11) & 31u);
                if (++kk > se) break;
            }
            const uint64_t touched = __ballot(cf != 0);    // (an entry is never zero: it holds the bits it consumed)
            cf >>= 16;
            if (touched != 0) {
                const int mx = bx >> lh;
                int16_t *p = k.cbase + (int64_t)(rbase + mx * bpm + (bx - (mx << lh))) * 64;
                if ((touched >> lane) & 1) p[nat] = (int16_t)cf;
            }
        }
        ++m;
        if (++bx == smh) { bx = 0; ++by; rbase = row_base(by); }
    }
    k.eobrun = eobrun;
    k.err = err;
}

// -------------------------------------------------------------------------------------- AC, refining scan (:1122-1298, :1100-1115)
// SCOUT: the same walk without placing anything — symbols and correction counts only, the stream position and the end-of-band
// run at the first MCU of every part of the band written to k.sub_out.  A refining scan is one serial chain of ~6 000 cycles per
// block, and a batch lasts as long as the longest scan's chain; the scout's chain is about half of that, and the walks that place
// (this function without SCOUT, started from the scout's entries, kProgSub per band) run side by side one launch later.
template <bool SCOUT>
__device__ __forceinline__ void walk_ac_refine(Walk &k, Stream &st, const uint16_t *lut) {
    const DevProgScan *sc = k.sc;
    const int lane = k.lane, al = k.al, ss = k.ss, se = k.se;
    const bool spec = k.spec;
    const int bpm = k.im->blocks_per_mcu, fmx = k.im->mcu_count_h;
    const int c = sc->comp[0];
    const DevHuff *tab = k.huff + sc->ac_tab[0];
    int16_t *cbase = k.cbase;
    const int h = k.im->comp_h[c], v = k.im->comp_v[c], first = k.im->comp_first[c];   // (h, v: 1, 2 or 4 here — api.hip sends other factors to the general walk)
    const int smh = sc->mcu_count_h;
    const int nz_nat = c_nat_of_zz_ps[lane];
    const int nat = k.tr ? ((nz_nat & 7) << 3 | nz_nat >> 3) : nz_nat;        // tr: blocks are kept [u][v] for the row-major stage 2
    // this lane's coefficient of the scan's blocks, one after the other in scan order (h, v are 1, 2 or 4: shifts)
    const int lh = h == 4 ? 2 : h - 1, lv = v == 4 ? 2 : v - 1;
    // (the block's offset — wave-uniform — moves on by a constant from block to block: 1 block inside an MCU, to the component's first
    // block of the row in the next MCU otherwise; worked out in full only where a block row starts)
    int nby = k.m_lo / smh, nbx = k.m_lo - nby * smh;
    auto offset_of = [&](int bx, int by) -> int64_t {
        const int mx = bx >> lh, my = by >> lv;
        return ((int64_t)(my * fmx + mx) * bpm + first + ((by - (my << lv)) << lh) + (bx - (mx << lh))) * 64;
    };
    int64_t boff = offset_of(nbx, nby);
    const int hmask = (1 << lh) - 1, mcu_step = (bpm - hmask) * 64;
    int16_t *const lane_base = cbase + nat;
    auto next_elem = [&]() -> int16_t * {
        int16_t *p = lane_base + boff;
        if (++nbx == smh) { nbx = 0; ++nby; boff = offset_of(0, nby); }
        else boff += (nbx & hmask) ? 64 : mcu_step;
        return p;
    };
    AcWindows<true> win;                      // symbols of 64 consecutive bit offsets (see ac_entry)
    win.start(st, lut, al, lane, st.bp);
    const bool in_band = lane >= ss && lane <= se;
    const uint64_t band_from = from_bit(ss), band = bit_range(ss, se + 1);
    const int m_lo = k.m_lo, m_hi = k.m_hi;
    int err = 0, eobrun = k.eobrun;
#ifdef MJ_DIAGNOSTIC
#ifdef MJ_DIAG_SCOUT      // (the stamps on the scout instead of the placing walk)
    const bool dbg_on = SCOUT && sc->ah == 1 && c == 0;
#else
    const bool dbg_on = !SCOUT && sc->ah == 1 && c == 0;
#endif
    uint64_t dacc[7] = {0, 0, 0, 0, 0, 0, 0}, dlast = __builtin_amdgcn_s_memtime();
    unsigned long long dblocks = 0, deob = 0, dplaced = 0;
#endif

    // What a block's walk needs from its history, worked out one block ahead (the two cross-lane permutes are LDS round trips)
    struct Prep {
        uint64_t nzb;         // history: non-zero coefficients from Ss on
        int rank0;            // ... how many of them below this lane
        int nband;            // ... how many of them in the band [Ss, Se]
        int nzeros;           // zero-history positions from Ss on
        int nzband;           // ... of them inside the band [Ss, Se]: what the symbol loops may take (see there)
        uint32_t zpos;        // lane j: position of the j-th of them
        uint32_t ztab;        // lane j: history-non-zero coefficients in front of it = correction bits read up to there
    };
    auto prepare = [&](int cf, bool in_eob_run) __attribute__((always_inline)) -> Prep {
        Prep q;
        const uint64_t nz0 = __ballot(cf != 0);
        q.nzb = nz0 & band_from;
        q.rank0 = mbcnt(q.nzb);
        q.nband = __builtin_popcountll(q.nzb & band);
        q.nzeros = 0; q.nzband = 0; q.zpos = 0; q.ztab = 0;
        if (in_eob_run) return q;                          // a bit for every non-zero coefficient of the band: no zero runs to follow
        const uint64_t zb = ~nz0 & band_from;
        q.nzeros = __builtin_popcountll(zb);
        q.nzband = __builtin_popcountll(zb & band);
        const int zrank = mbcnt(zb);
        // lane l sends its number to lane zrank (zeros) / behind all zeros (the others)
        const int slot = (cf == 0 && lane >= ss) ? zrank : q.nzeros + lane - zrank;
        q.zpos = (uint32_t)__builtin_amdgcn_ds_permute(slot << 2, lane);
        if (!SCOUT) q.ztab = q.zpos - (uint32_t)(ss + lane);           // of the zpos - Ss positions in front of zero number `lane`, `lane` are zeros
        return q;
    };
    // the scout's block: the symbol chain and the counts of correction bits, nothing else
    auto scout_block = [&](const Prep &pr) __attribute__((always_inline)) {
        PSTAMP(6);
        st.top_up();
        if (eobrun > 0) { st.bp += pr.nband; --eobrun; PSTAMP(5); return; }
        const int nzeros = pr.nzeros;
        const uint32_t zpos = pr.zpos;
        int k = ss, jz = 0;
        int u = st.bp;
        for (;;) {
            uint32_t e;
            int code, off;
            for (;;) {
                // 14 instructions per symbol.  In front of zero number zl at position k1 lie k1 - Ss - zl history-non-zero
                // coefficients — the correction bits read so far — so the next symbol starts at u + k1 - Ss - zl; ur = u - gbase - Ss
                // makes that a window offset in two additions.  Codes as in the placing loop below.
#ifdef MJ_DIAGNOSTIC
                int t0, k1 = rfl(k - 1), zl = rfl(jz - 1);
#else
                int t0, k1 = k - 1, zl = jz - 1;
#endif
                const int gss = win.gbase + ss;
                int ur = u - gss;
                PSTAMP(0);
                asm volatile(
                    "s_sub_u32 %[off], %[bp], %[gbase]\n\t"
                    "s_cmp_gt_u32 %[off], 63\n\t"
                    "s_cbranch_scc1 Lcwin%=\n\t"
                    "v_readlane_b32 %[e], %[ve0], %[off]\n"
                    ".p2align " MJ_LOOP_ALIGN "\n" MJ_LOOP_PAD(MJ_SKEW_C)
                    "Lcsym%=:\n\t"
#define MJ_SCOUT_SYMBOL \
                    "s_bfe_u32 %[t0], %[e], %[krun]\n\t"             \
                    "s_add_u32 %[zl], %[zl], %[t0]\n\t"              \
                    "s_cmp_ge_u32 %[zl], %[nzeros]\n\t"              \
                    "s_cbranch_scc1 Lcover%=\n\t"                    \
                    "v_readlane_b32 %[k1], %[zpos], %[zl]\n\t"       \
                    "s_bfe_u32 %[t0], %[e], %[kbits]\n\t"            \
                    "s_add_u32 %[ur], %[ur], %[t0]\n\t"              \
                    "s_sub_u32 %[t0], %[k1], %[zl]\n\t"              \
                    "s_add_u32 %[off], %[ur], %[t0]\n\t"             \
                    "v_readlane_b32 %[e], %[ve0], %[off]\n\t"        /* (an offset past 63 reads some lane's entry, which is then not used) */ \
                    "s_cmp_le_u32 %[off], 63\n\t"
#if MJ_SCOUT_UNROLL >= 2
                    MJ_SCOUT_SYMBOL "s_cbranch_scc0 Lcwin%=\n\t"      // (two symbols per turn: one taken branch per two symbols)
#endif
                    MJ_SCOUT_SYMBOL "s_cbranch_scc1 Lcsym%=\n"
                    "Lcwin%=:\n\t"
                    "s_mov_b32 %[code], 1\n\t"
                    "s_branch Lcend%=\n"
                    "Lcdone%=:\n\t"
                    "s_mov_b32 %[code], 0\n\t"
                    "s_branch Lcend%=\n"
                    "Lcspec%=:\n\t"
                    "s_bitcmp1_b32 %[e], 0\n\t"
                    "s_cbranch_scc1 Lclong%=\n\t"
                    "s_lshr_b32 %[eob], %[e], 16\n\t"
                    "s_bfe_u32 %[t0], %[e], %[kbits]\n\t"
                    "s_add_u32 %[off], %[off], %[t0]\n\t"
                    "s_add_u32 %[ur], %[ur], %[t0]\n\t"
                    "s_mov_b32 %[code], 4\n\t"
                    "s_branch Lcend%=\n"
                    "Lclong%=:\n\t"
                    "s_mov_b32 %[code], 2\n\t"
                    "s_branch Lcend%=\n"
                    "Lcover%=:\n\t"
                    "s_sub_u32 %[zl], %[zl], %[t0]\n\t"
                    "s_cmp_ge_u32 %[k1], %[se]\n\t"               // the band's last position is taken: `e` is the next block's symbol
                    "s_cbranch_scc1 Lcdone%=\n\t"
                    "s_and_b32 %[t0], %[e], 3\n\t"
                    "s_cbranch_scc1 Lcspec%=\n\t"
                    "s_mov_b32 %[code], 3\n"
                    "Lcend%=:"
                    : [e] "=&s"(e), [code] "=&s"(code), [t0] "=&s"(t0), [off] "=&s"(off), [ur] "+s"(ur), [k1] "+s"(k1), [zl] "+s"(zl),
                      [eob] "+s"(eobrun)
                    : [bp] "s"(st.bp), [gbase] "s"(win.gbase), [ve0] "v"(win.ve0), [zpos] "v"(zpos), [se] "s"(se), [nzeros] "s"(pr.nzband),
                      [krun] "s"(0x80002), [kbits] "s"(0x5000b)      // (the two field selectors from registers: literals would make the loop 72 bytes, with them it is 64)
                    : "scc");
                k = rfl(k1) + 1; jz = rfl(zl) + 1;
                e = (uint32_t)rfl((int)e); code = rfl(code); off = rfl(off); eobrun = rfl(eobrun);
                u = rfl(ur) + gss; st.bp = win.gbase + off;
                PSTAMP(1);
                if (code != 1) break;
                win.move_to(st, lut, al, lane, off);
                PSTAMP(2);
            }
            if (code == 0 || code == 4) break;
            // (code 3: the loop takes zeros of the band only — it needs no test for the band's end per symbol that way: the symbol
            // behind the band's last zero stops it, code 0 if the band is full — and a run that ends behind the band or behind all
            // zeros is placed or refused by the single step below)
            if (e & 1u) {
                const uint32_t w = rdl(win.vw0, off);
                int len, hv;
                long_code(w, tab, 1, len, hv);              // (the LUT holds finished entries only: any length)
                e = ac_entry<true>(w, len, hv, al);
                if (len == 0) { err = MJ_ST_BAD_CODE; break; }
            }
            if (e & 2u) {
                eobrun = (int)(e >> 16);
                st.bp += (int)((e >> 11) & 31u);
                u += (int)((e >> 11) & 31u);
                break;
            }
            const int jt2 = jz + (int)((e >> 2) & 31u) - 1;
            if (jt2 >= nzeros) { err = MJ_ST_OVERRUN; break; }
            const int pz2 = (int)rdl(zpos, jt2);
            u += (int)((e >> 11) & 31u);
            st.bp = u + pz2 - ss - jt2;
            k = pz2 + 1; jz = jt2 + 1;
            if (k > se) break;
        }
        if (!err && eobrun > 0) {
            if (k <= se) st.bp = u + pr.nband;
            --eobrun;
        }
#ifdef MJ_DIAGNOSTIC
        if (dbg_on) ++dblocks;
#endif
        PSTAMP(4);
    };
    auto one_block = [&](int cf, int16_t *p, const Prep &pr) __attribute__((always_inline)) {
        if constexpr (SCOUT) { scout_block(pr); return; }
        PSTAMP(6);
        st.top_up();
        const uint64_t nzb = pr.nzb;
        const int rank0 = pr.rank0;
        const bool hist = cf != 0 && lane >= ss;           // this lane's coefficient is history-non-zero (the walk only writes zero ones)
        const int cf0 = cf;
        int vbase = 0;                                     // this lane's correction bit is bit vbase + rank0 of the stream
        int kend;                                          // corrections go to the history-non-zero lanes below kend
        bool dirty = false;
#ifdef MJ_DIAGNOSTIC
        bool dbg_eob = false;
#endif
        if (eobrun > 0) {                                  // inside an end-of-band run: a bit for every non-zero coefficient of the band
            vbase = st.bp;
            st.bp += pr.nband;
            kend = se + 1;
            --eobrun;
#ifdef MJ_DIAGNOSTIC
            dbg_eob = true;
#endif
        } else {
            const int nzeros = pr.nzeros;
            const uint32_t zpos = pr.zpos, ztab = pr.ztab;
            int k = ss, jz = 0;                            // jz = zeros below k
            int u = st.bp;                                 // the bit position without the correction bits read in this block
#ifdef MJ_DIAGNOSTIC
            bool dbg_first = true;
#endif
            for (;;) {                                     // (Ss <= Se: a scan has at least one coefficient per block)
                uint32_t e;
                int code, off;
                for (;;) {                                 // (the common way round on its own: symbols, next window, symbols ...)
                // The run of plain coefficient symbols inside the current window, hand-scheduled and software-pipelined: 22
                // instructions per symbol, two v_readlane deep, the next symbol's entry in flight during this symbol's
                // bookkeeping (the compiler's version of the same loop: ~40, a third of them branch bookkeeping).
                // Leaves with code 0: the band is full;  1: the next symbol starts behind the window;  2: entry `e` is not in the LUT;
                // 3: the zero run passes the band's last zero;  4: end of band, run length in eobrun.
                // Wait states: no v_readlane takes its lane select from a VALU-written SGPR; SALU reads of those are interlocked.
                uint32_t e2;
                int t0, cn;
                int vt;
                // (inside the loop the position is kept as k1 = k - 1, the last position taken, and the zeros passed as zl = jz - 1,
                // the ordinal of the last zero taken: the entry's run field holds r + 1, v_readlane writes the new position
                // straight into k1 — two additions per symbol less)
#ifdef MJ_DIAGNOSTIC   // (with the stamps in between, the compiler otherwise keeps these two in vector registers)
                int k1 = __builtin_amdgcn_readfirstlane(k - 1), zl = __builtin_amdgcn_readfirstlane(jz - 1);
#else
                int k1 = k - 1, zl = jz - 1;
#endif
                PSTAMP(dbg_first ? 0 : 3);
#ifdef MJ_DIAGNOSTIC
                dbg_first = false;
#endif
                asm volatile(
                    // prologue: the first symbol of the run
                    "s_sub_u32 %[off], %[bp], %[gbase]\n\t"
                    "s_cmp_gt_u32 %[off], 63\n\t"
                    "s_cbranch_scc1 Lwin%=\n\t"
                    "v_readlane_b32 %[e], %[ve0], %[off]\n"
                    ".p2align " MJ_LOOP_ALIGN "\n" MJ_LOOP_PAD(MJ_SKEW_R)
                    "Lsym%=:\n\t"
                    // (two symbols per turn, the entry registers swapping roles: no move of the next entry into the current one's place)
#define MJ_PLACE_SYMBOL(E, E2, OVER) \
                    "s_bfe_u32 %[t0], %[" E "], 0x80002\n\t"      /* run + 1 + 64 * class: what is no plain coefficient overshoots every zero count */ \
                    "s_add_u32 %[zl], %[zl], %[t0]\n\t"           \
                    "s_cmp_ge_u32 %[zl], %[nzeros]\n\t"           \
                    "s_cbranch_scc1 " OVER "%=\n\t"               \
                    "v_cmp_lt_u32 vcc, %[k1], %[vlane]\n\t"       /* the lanes from the old k on (they take this symbol's u below) */ \
                    "v_readlane_b32 %[cn], %[ztab], %[zl]\n\t"    \
                    "v_readlane_b32 %[k1], %[zpos], %[zl]\n\t"    \
                    "s_bfe_u32 %[t0], %[" E "], 0x5000b\n\t"      \
                    "s_add_u32 %[u], %[u], %[t0]\n\t"             \
                    "s_add_u32 %[bp], %[u], %[cn]\n\t"            \
                    /* the next symbol's entry is requested before this symbol's bookkeeping (a window offset past 63 reads some */ \
                    /* lane's entry, which is then not used) */     \
                    "s_sub_u32 %[off], %[bp], %[gbase]\n\t"       \
                    "v_readlane_b32 %[" E2 "], %[ve0], %[off]\n\t" \
                    "v_mov_b32 %[vt], %[u]\n\t"                   /* (v_cndmask cannot take u from its SGPR: vcc is the one constant-bus operand) */ \
                    "v_cndmask_b32 %[vbase], %[vbase], %[vt], vcc\n\t" \
                    "s_mov_b32 m0, %[k1]\n\t"                     \
                    "v_writelane_b32 %[cf], %[" E "], m0\n\t"     /* (the whole entry: its value is taken out of the lanes once per block) */ \
                    "s_cmp_le_u32 %[off], 63\n\t"
                    MJ_PLACE_SYMBOL("e", "e2", "Lover")
                    "s_cbranch_scc0 Lwin%=\n\t"
                    MJ_PLACE_SYMBOL("e2", "e", "Lover2")
                    "s_cbranch_scc1 Lsym%=\n"
                    "Lwin%=:\n\t"
                    "s_mov_b32 %[code], 1\n\t"
                    "s_branch Lend%=\n"
                    "Ldone%=:\n\t"
                    "s_mov_b32 %[code], 0\n\t"
                    "s_branch Lend%=\n"
                    "Lspec%=:\n\t"                              // EOBn found in the LUT: the run's length, the bits of code and run
                    "s_bitcmp1_b32 %[e], 0\n\t"
                    "s_cbranch_scc1 Llong%=\n\t"
                    "s_lshr_b32 %[eob], %[e], 16\n\t"
                    "s_bfe_u32 %[t0], %[e], 0x5000b\n\t"
                    "s_add_u32 %[bp], %[bp], %[t0]\n\t"
                    "s_add_u32 %[u], %[u], %[t0]\n\t"
                    "s_mov_b32 %[code], 4\n\t"
                    "s_branch Lend%=\n"
                    "Llong%=:\n\t"
                    "s_mov_b32 %[code], 2\n\t"
                    "s_branch Lend%=\n"
                    "Lover2%=:\n\t"
                    "s_mov_b32 %[e], %[e2]\n"
                    "Lover%=:\n\t"
                    "s_sub_u32 %[zl], %[zl], %[t0]\n\t"          // (not taken: the count goes back)
                    "s_cmp_ge_u32 %[k1], %[se]\n\t"               // the band's last position is taken: `e` is the next block's symbol
                    "s_cbranch_scc1 Ldone%=\n\t"
                    "s_and_b32 %[t0], %[e], 3\n\t"               // a special entry, or a zero run past the band's last zero
                    "s_cbranch_scc1 Lspec%=\n\t"
                    "s_mov_b32 %[code], 3\n"
                    "Lend%=:"
                    : [e] "=&s"(e), [e2] "=&s"(e2), [code] "=&s"(code), [t0] "=&s"(t0), [cn] "=&s"(cn),
                      [off] "=&s"(off), [vt] "=&v"(vt), [bp] "+s"(st.bp), [u] "+s"(u), [k1] "+s"(k1), [zl] "+s"(zl), [cf] "+v"(cf),
                      [vbase] "+v"(vbase), [eob] "+s"(eobrun)
                    : [gbase] "s"(win.gbase), [ve0] "v"(win.ve0), [zpos] "v"(zpos), [ztab] "v"(ztab), [vlane] "v"(lane), [se] "s"(se),
                      [nzeros] "s"(pr.nzband)
                    : "vcc", "scc", "m0");
                // (the compiler takes what an asm statement writes for lane-varying, whatever the constraint says, and does the
                // arithmetic on it with vector instructions and mask branches: readfirstlane tells it otherwise and folds away)
                k = rfl(k1) + 1; jz = rfl(zl) + 1;
                e = (uint32_t)rfl((int)e); code = rfl(code); off = rfl(off); u = rfl(u); st.bp = rfl(st.bp); eobrun = rfl(eobrun);
                PSTAMP(1);
                if (code != 1) break;
                win.move_to(st, lut, al, lane, off);       // next window of looked-up symbols
#ifdef MJ_DIAGNOSTIC
                PSTAMP(2);
#endif
                }
                if (code == 0 || code == 4) break;         // the band is done / end of band (:1160-1166)
                // (code 3: the loop takes zeros of the band only — it needs no test for the band's end per symbol that way: the
                // symbol behind the band's last zero stops it, code 0 if the band is full — and a run that ends behind the band or
                // behind all zeros is placed or refused by the single step below, :1190)
                if (e & 1u) {                              // a code longer than the LUT's index (rare) or no code at all
                    const uint32_t w = rdl(win.vw0, off);
                    int len, hv;
                    long_code(w, tab, 1, len, hv);              // (the LUT holds finished entries only: any length)
                    e = ac_entry<true>(w, len, hv, al);
                    if (len == 0) { err = MJ_ST_BAD_CODE; break; }
                }
                if (e & 2u) {                              // end of band: this block's rest and eobrun - 1 further blocks (:1160-1166)
                    eobrun = (int)(e >> 16);
                    st.bp += (int)((e >> 11) & 31u);
                    u += (int)((e >> 11) & 31u);
                    break;
                }
                // (a long code's coefficient: the same step as above, once)
                const int jt2 = jz + (int)((e >> 2) & 31u) - 1;          // (the refining walk's entries hold r + 1)
                if (jt2 >= nzeros) { err = MJ_ST_OVERRUN; break; }
                const int pz2 = (int)rdl(zpos, jt2);
                write_lane(cf, (int)e, pz2);               // (:1225; the whole entry, as the loop above leaves it)
                u += (int)((e >> 11) & 31u);                // the symbol's corrections follow its value bits (:1202, :1231)
                vbase = lane >= k ? u : vbase;
                st.bp = u + (int)rdl(ztab, jt2);
                k = pz2 + 1; jz = jt2 + 1;
                if (k > se) break;
            }
            PSTAMP(3);
            // the lanes this scan wrote hold whole entries (value in the upper half): only zero-history lanes can be among them
            cf = cf0 == 0 ? cf >> 16 : cf;
            dirty = true;
            kend = k;
            if (!err && eobrun > 0) {                      // rest of the band, then the run continues in the next blocks
                vbase = lane >= k ? u : vbase;
                // st.bp - u corrections have been read for [Ss, k): the band's other history-non-zero coefficients follow
                if (k <= se) st.bp = u + pr.nband;
                kend = max(k, se + 1);
                --eobrun;
            }
        }
        const uint64_t corr = nzb & ~from_bit(kend);
        if (corr != 0) {
            if (hist && lane < kend) {
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
#ifdef MJ_DIAGNOSTIC
        if (dbg_on) {
            if (dbg_eob) ++deob; else { ++dblocks; dplaced += __builtin_popcountll(__ballot(cf != 0 && lane >= ss) & ~nzb); }
        }
        PSTAMP(dbg_eob ? 5 : 4);
#endif
    };


    DevProgSub *sub_entry = k.sub_out;
    int sub_next = SCOUT ? m_lo : -1;
    // The next block's coefficients are asked for before this block is walked (a block is ~1 us of HBM latency away and takes
    // 1.4 us or more to walk).  One copy of the block's code, not four with four blocks in flight: the walks are instruction-fetch
    // bound as much as anything — 16 files 60.2 -> 51.9 ms, 1024: 81.1 -> 79.2, 2048: 122.6 -> 120.3 with the smaller loop.
    int16_t *pn = cbase + nat;
    int cfn = 0;
    if (m_lo < m_hi) { pn = next_elem(); cfn = *pn; }
    for (int m = m_lo; m < m_hi && !err; ++m) {
        const int cf = cfn;
        int16_t *p = pn;
        if (m + 1 < m_hi) { pn = next_elem(); cfn = *pn; }
        if (SCOUT && m == sub_next) {                      // the first MCU of a part of the band: what its walk starts from
            if (lane == 0) *sub_entry = DevProgSub{st.bp, eobrun, 0, m};
            ++sub_entry; sub_next += k.sub_q;
        }
        // (working the next block's tables out one block ahead, to hide the permute's latency, was slower: the tables
        // of two blocks alive at once cost scalar registers the symbol loop's surroundings need)
        one_block(cf, p, prepare(cf, eobrun > 0));
    }
    k.eobrun = eobrun;
    k.err = err;
#ifdef MJ_DIAGNOSTIC
    if (dbg_on && lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_dbg_prog[i], (unsigned long long)dacc[i]);
        atomicAdd(&g_dbg_prog[7], 1ull);
        atomicAdd(&g_dbg_prog[8], dblocks); atomicAdd(&g_dbg_prog[9], deob); atomicAdd(&g_dbg_prog[10], dplaced);
    }
#endif
}

}  // namespace

template <bool BANDED>
__global__ __launch_bounds__(256) void k_progressive_fast(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                          const DevProgSeg *__restrict__ segs, int n_segs,
                                                          const DevProgScan *__restrict__ scans, const DevImage *__restrict__ images,
                                                          const DevHuff *__restrict__ huff, const uint16_t *__restrict__ lut11p,
                                                          int16_t *__restrict__ coef, int32_t *__restrict__ status, int spec_refine, int tr,
                                                          DevProgState *__restrict__ states, int step, int rows_per_band,
                                                          int n_split, DevProgSub *__restrict__ subs, int parts) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[4][kPLut];        // AC: one table of 11 bits; DC: up to three of 9 bits
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[4][kRingDw];
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    // One wave per segment, the longest walks first (a launch lasts as long as its slowest wave, and holds more waves than the
    // chip has slots): the split scans' segments — the first n_split — as scouts, then their kProgSub parts of the band each, the
    // walks that place, then the other segments.  (wave-uniform; no workgroup barriers below)
    int seg_id = blockIdx.x * 4 + wave, part = -1;
    if (seg_id >= n_split) {
        const int r = seg_id - n_split;
        if (r < n_split * parts) { seg_id = r / parts; part = r - seg_id * parts; }
        else seg_id = r - n_split * (parts - 1);
    }
    if (seg_id >= n_segs) return;
#ifdef MJ_DIAGNOSTIC
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    Walk k;
    k.sg = segs + seg_id;
    k.sc = scans + k.sg->scan;
    const DevProgScan *sc = k.sc;
    k.ss = sc->ss; k.se = sc->se; k.al = sc->al; k.tr = tr & 1; k.lane = lane; k.spec = spec_refine != 0;
    const bool sequential = k.ss == 0 && k.se == 63;
    const bool is_dc = k.ss == 0;
    if (sequential) return;                                    // progressive.hip's
    k.im = images + sc->image;
    k.huff = huff;
    k.cbase = coef + k.im->block_off * 64;

    // ---- which part of the scan this launch does
    int b_lo = k.sg->mcu0, b_hi = k.sg->mcu0 + k.sg->n_mcu;
    const bool scout = BANDED && sc->split != 0 && part < 0;
    DevProgSub *sub = nullptr;
    if (BANDED) {
        const int band = step - sc->level - (part >= 0 ? 1 : 0);   // the parts follow the scout one launch behind
        if (band < 0) return;
        if (sc->split) sub = subs + ((size_t)seg_id * 2 + (band & 1)) * kProgSub;
        const int v_scan = sc->n_comp == 1 ? k.im->comp_v[sc->comp[0]] : 1;   // block rows per frame MCU row
        const int64_t mpb = (int64_t)sc->mcu_count_h * rows_per_band * v_scan;
        const int64_t lo = (int64_t)band * mpb, hi = lo + mpb;
        b_lo = (int)max((int64_t)b_lo, lo);
        b_hi = (int)min((int64_t)b_hi, hi);
        if (b_lo >= b_hi) return;
    }
    k.m_lo = b_lo; k.m_hi = b_hi;
    k.sub_out = sub; k.sub_q = (b_hi - b_lo + parts - 1) / parts;
    const bool resume = BANDED && part < 0 && b_lo != k.sg->mcu0, finish = !BANDED || (part < 0 && b_hi == k.sg->mcu0 + k.sg->n_mcu);
    if (is_dc && sc->ah != 0) {                                // DC refinement: nothing is carried from band to band
        k.lane = lane;
        const int total_bits = seg_bits[k.sg->stream_slot];
        const int bp = walk_dc_refine(k, stream + (k.sg->begin >> 2) + k.sg->stream_slot, (total_bits + 31) >> 5);
        if (finish && lane == 0) {                             // (Stream::end_status)
            const int err = bp > total_bits ? MJ_ST_OVERRUN : (k.sg->last == 0 && total_bits - bp >= 8 ? MJ_ST_DESYNC : 0);
            if (err) atomicMax(status + sc->image, err);
        }
        return;
    }
    DevProgState *ps = states + seg_id;
    int bp0 = 0;
    k.eobrun = 0; k.pred0 = k.pred1 = k.pred2 = 0; k.err = 0;
    if (part >= 0) {                                           // a part of the band: from where the scout stood at its first MCU
        const int lo = b_lo + part * k.sub_q;
        const DevProgSub e = sub[part];
        if (lo >= b_hi || e.mcu != lo || e.err != 0) return;   // (no such part; the scout did not get there: the scan's status is set)
        k.m_lo = lo; k.m_hi = min(b_hi, lo + k.sub_q);
        bp0 = e.pos; k.eobrun = e.eobrun;
    }
    if (resume) {
        if (ps->err != 0 || ps->mcu_next != b_lo) return;      // the scan failed earlier (its status is set)
        bp0 = ps->pos; k.eobrun = ps->eobrun; k.pred0 = ps->pred[0]; k.pred1 = ps->pred[1]; k.pred2 = ps->pred[2];
    }

    uint16_t *lut = s_lut[wave];
    if (is_dc) {
        for (int t = 0; t < sc->n_comp; ++t) load_dc_lut(lut + t * kPDcLut, lut11p, sc->dc_tab[t], lane);
    } else {
        if (sc->ah != 0) load_lut_refine(lut, lut11p, sc->ac_tab[0], lane);
        else load_lut(lut, lut11p, sc->ac_tab[0], lane);
    }
    Stream st;
    st.init(s_ring[wave], stream, seg_bits, k.sg, lane, bp0);

    if (is_dc) walk_dc_first(k, st, lut);
    else if (sc->ah == 0) walk_ac_first(k, st, lut);
    else if (scout) walk_ac_refine<true>(k, st, lut);
    else walk_ac_refine<false>(k, st, lut);

#ifdef MJ_DIAGNOSTIC
    const int dbg_slot = blockIdx.x * 4 + wave;
    if (lane == 0 && dbg_slot < 32768 && spec_refine >= 0 && (tr >> 8) == step + 1) {     // (tr's upper bits: the launch to record, plus one)
        unsigned long long *o = g_dbg_prog_waves + (size_t)dbg_slot * 3;
        o[0] = dbg_r0; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = part >= 0 ? 4 : is_dc ? 0 : (sc->ah == 0 ? 1 : (sc->comp[0] == 0 ? 2 : 3));
    }
#endif
    int err = k.err;
    if (BANDED && part < 0 && !finish && lane == 0) {
        ps->pos = st.bp; ps->eobrun = k.eobrun; ps->pred[0] = k.pred0; ps->pred[1] = k.pred1; ps->pred[2] = k.pred2;
        ps->err = err; ps->mcu_next = b_hi;
    }
    if (!err && finish) err = st.end_status(k.sg->last != 0);      // (a split scan's scout reads every bit the parts read)
    if (err && lane == 0) atomicMax(status + sc->image, err);
}

// rows_per_band > 0: launch number `step` of the band pipeline over all segments; otherwise the segments of one dependency level
hipError_t launch_progressive_fast(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevProgSeg *segs,
                                   int n_segs, const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                   const uint16_t *lut11p, int16_t *coef, int32_t *status, int spec_refine, int transposed,
                                   DevProgState *states, int step, int rows_per_band, int n_split, DevProgSub *subs, int parts) {
    if (n_segs == 0) return hipSuccess;
    if (rows_per_band <= 0) n_split = 0;
    parts = std::min(std::max(parts, 1), kProgSub);
    const dim3 grid((unsigned)((n_segs + n_split * parts + 3) / 4));
#ifdef MJ_DIAGNOSTIC
    if (const char *e = getenv("MJ_DEBUG_PROG_STEP")) transposed |= (atoi(e) + 1) << 8;
#endif
    if (rows_per_band > 0)
        hipLaunchKernelGGL(k_progressive_fast<true>, grid, dim3(256), 0, stream, dstream, seg_bits, segs, n_segs, scans, images, huff,
                           lut11p, coef, status, spec_refine, transposed, states, step, rows_per_band, n_split, subs, parts);
    else
        hipLaunchKernelGGL(k_progressive_fast<false>, grid, dim3(256), 0, stream, dstream, seg_bits, segs, n_segs, scans, images, huff,
                           lut11p, coef, status, spec_refine, transposed, states, 0, 0, 0, nullptr, 1);
    return hipGetLastError();
}

}  // namespace mj
