// Stage 1, lane-parallel form with RESOLVED 13-bit tables: the walk itself (one restart segment per lane), shared by the
// stage-1 kernel (huffman_lanes13.hip, whose header describes it) and by the fused kernel (fused.hip), whose producer
// wavefronts run exactly this code beside the wavefronts that reconstruct what it decodes.
//   stage()   every thread of the workgroup: the tables into LDS, the block rows cleared (a __syncthreads() must follow)
//   walk()    one wavefront: its lanes' segments from first to last MCU
// jpeg_decoder.py:654-866, :894-900 (bit reader, next_huffval, EXTEND, the entropy loop of baseline_dct_scan).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "mijpeg_internal.h"

namespace mj {
namespace lanes13 {

// (AC LUT index bits: 13 — kASlotBytes below is the table with its second-level tables)
constexpr int kASlotBytes = kLanes13SlotBytes;   // main table + second-level tables of one AC table
constexpr int kDBits = kLaneLutBits;         // DC LUTs: the 11-bit (len << 8 | symbol) tables of the other lane form
constexpr int kDSize = 1 << kDBits;
constexpr int kRow = 33;                     // dwords per lane block in LDS (32 + 1 pad)

static __constant__ uint8_t c_zz_of_nat_13[64] = {
    0,  1,  5,  6, 14, 15, 27, 28,  2,  4,  7, 13, 16, 26, 29, 42,
    3,  8, 12, 17, 25, 30, 41, 43,  9, 11, 18, 24, 31, 40, 44, 53,
   10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
   21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

typedef uint32_t __attribute__((address_space(3))) *lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)p;
}

// canonical search over code lengths l0..16 (jpeg_decoder.py:366-377 semantics); (len << 8) | symbol, or -1
static __device__ __noinline__ int canon_code(const DevHuff *t, uint32_t p16, int l0) {
    int r = -1;
#pragma unroll 1
    for (int l = l0; l <= 16; ++l) {
        const int d = (int)(p16 >> (16 - l)) - t->first_code[l];
        if (r < 0 && d >= 0 && d < t->count[l]) r = (l << 8) | t->vals[t->first_sym[l] + d];
    }
    return r;
}

__device__ __forceinline__ int extend13(uint32_t raw, int n) {   // bin_twos_complement (:1636-1646); n = 0 -> 0
    const int half = (1 << n) >> 1;
    return (int)raw - (((int)raw < half) ? ((1 << n) - 1) : 0);
}


#ifdef MJ_X_STAMP
static __device__ unsigned long long g_dbg13[16];
#endif
#ifdef MJ_DIAGNOSTIC     // when does every wave of the launch finish?  (100 MHz wall clock; diagnostic build only)
static __device__ unsigned long long g_dbg13_waves[4096 * 3];
#endif

// what the kernels pass on (the stage-1 launch's arguments, see launch_huffman_lanes13)
struct Args {
    const uint32_t *stream;        // stage 0's output
    const int32_t *seg_bits;
    const DevSegment *segs;
    int64_t n_segs;
    const DevImage *images;
    const DevHuff *huff;
    const uint16_t *lut11;         // [n_huff][kDSize]: DC tables are read from here
    const uint32_t *lut13;         // [n_ac][kASize]
    int n_ac, n_dc;
    uint64_t ac_slot_pk;           // byte t = LDS slot of table t as an AC table
    uint64_t dc_slot_pk;           // byte t = LDS slot of table t as a DC table
    uint64_t dc_tab_pk;            // byte s = table index held by DC slot s
    int16_t *coef;
    int32_t *status;
    int lpw, tr;
    const DevVSeg *vsegs;          // or null
    const int32_t *by_length;      // or null: segment numbers, longest first
    int order_mode;                // 1 = a wave takes neighbours of that list, 2 = one of every stride
    int kRing;                     // bytes of stream per lane in LDS: 128, or 64 to fit more lanes
    // fused.hip only (the stage-1 kernel keeps the constants): bytes of one AC table in LDS — its main table of 2^AB entries
    // plus the second-level tables the batch's codes need, lut13 holding the tables at this stride — and index bits of the DC
    // tables in LDS (taken out of lut11's 11-bit ones: every 2^(11 - dbits)-th entry, where its code is short enough)
    int ac_total_bytes, dbits;
    // ... the AC tables lie back to back in lut13 and in LDS: byte offset and main-level index bits (12 or 13) per LDS slot
    int ac_off[4], ac_bits[4];
    // fused.hip only: restart segments per workgroup (whole images: no multiple of the waves' lanes in general — the last
    // lanes of a workgroup's last wave then have no segment)
    int wg_segs;
    // fused.hip, segments dealt out by length (MODE 2): a wave's progress word in global memory, at [workgroup * waves + wave]
    uint32_t *progress_global;
};

// LDS bytes of `nw` waves of `lpw` lanes: tables, block rows, block addresses, stream windows — in this order from `smem`
__host__ __device__ inline size_t lds_bytes(int ac_total_bytes, int n_dc, int nw, int lpw, int kRing, int dbits = kDBits) {
    const int wstride = (lpw * kRow + 3) & ~3;
    return (((size_t)ac_total_bytes + ((size_t)n_dc << dbits) * 2 + (size_t)nw * wstride * 4 + 8 * kRow * 4 + (size_t)(nw * lpw + 8) * 8 + 127) & ~(size_t)127) +
           (size_t)nw * lpw * kRing;
}

// the tables into LDS, the rows cleared: `nthreads` threads (tid = 0 .. nthreads - 1) of the workgroup, all of them
template <bool FUSED>
__device__ __forceinline__ void stage(const Args &A, unsigned char *smem, int tid, int nthreads, int nw) {
    const int n_ac = A.n_ac, n_dc = A.n_dc;
    const int ac_total = FUSED ? A.ac_total_bytes : n_ac * kASlotBytes, dbits = FUSED ? A.dbits : kDBits, dsize = 1 << dbits;
    uint32_t *s_ac = reinterpret_cast<uint32_t *>(smem);                                         // the AC tables, back to back
    uint16_t *s_dc = reinterpret_cast<uint16_t *>(smem + (size_t)ac_total);                    // [n_dc][dsize]
    const int wstride = (A.lpw * kRow + 3) & ~3;
    unsigned char *rows0 = smem + (size_t)ac_total + (size_t)n_dc * dsize * 2;
    for (int i = tid; i < ac_total / 16; i += nthreads)
        reinterpret_cast<uint4 *>(s_ac)[i] = reinterpret_cast<const uint4 *>(A.lut13)[i];
    for (int s = 0; s < n_dc; ++s) {
        const int t = (int)((A.dc_tab_pk >> (8 * s)) & 0xFF);
        if (!FUSED || dbits == kDBits) {
            for (int i = tid; i < kDSize / 8; i += nthreads)
                reinterpret_cast<uint4 *>(s_dc + s * kDSize)[i] = reinterpret_cast<const uint4 *>(A.lut11 + (size_t)t * kDSize)[i];
        } else {        // a shorter index: the entry of the index padded with zeros, if its code fits the shorter index (else 0 = "longer")
            for (int i = tid; i < dsize; i += nthreads) {
                const uint16_t e = A.lut11[(size_t)t * kDSize + ((size_t)i << (kDBits - dbits))];
                s_dc[s * dsize + i] = (e >> 8) <= dbits ? e : (uint16_t)0;
            }
        }
    }
    for (int i = tid; i < nw * wstride; i += nthreads) reinterpret_cast<uint32_t *>(rows0)[i] = 0;
}

// One wavefront's walk: wave `wave` of the `nw` of workgroup `wg` (of `n_wg`).  MODE 0: the stage-1 kernel.  MODE 1 (fused.hip,
// whole images per workgroup): the wave reports, in the LDS word at `progress_addr`, how many MCUs of its lanes' segments are
// complete in memory (see the hook behind the AC loop) — as progress_base + that number, and progress_base + 2^20 when it is
// through: a workgroup whose images are more segments than its producers have lanes walks them in passes (`wg` is then the
// pass's "virtual workgroup"), and progress_base = pass << 20 keeps the word rising from pass to pass.
// MODE 2 (fused.hip, segments dealt out by length — any workgroup's
// consumers may need this wave's blocks): the coefficient stores are write-through (sc1) and the report goes to
// A.progress_global with an sc1 store — MI355X_MICROARCH.md's "sc1 stores, drained, then an sc1 flag" hand-off.
// Index bits of an AC table's main level: 13 — or, in a fused launch, what A.ac_bits says for its LDS slot (12: half the LDS,
// 1.5 % of the benchmark's symbols instead of 0.4 % then take the arithmetic step); its second-level tables have 2^(16 - bits) entries
template <int MODE>
__device__ __forceinline__ void walk(const Args &A, unsigned char *smem, const int lane, const int wave, const int nw, const int wg, const int n_wg,
                                     const uint32_t progress_addr, const uint32_t progress_base = 0u) {
    const uint32_t *__restrict__ stream = A.stream;
    const int32_t *__restrict__ seg_bits = A.seg_bits;
    const DevSegment *__restrict__ segs = A.segs;
    const int64_t n_segs = A.n_segs;
    const DevImage *__restrict__ images = A.images;
    const DevHuff *__restrict__ huff = A.huff;
    const int n_ac = A.n_ac, n_dc = A.n_dc;
    const uint64_t ac_slot_pk = A.ac_slot_pk, dc_slot_pk = A.dc_slot_pk;
    int16_t *__restrict__ coef = A.coef;
    int32_t *__restrict__ status = A.status;
    const int lpw = A.lpw, tr = A.tr;
    const DevVSeg *__restrict__ vsegs = A.vsegs;
    const int32_t *__restrict__ by_length = A.by_length;
    const int order_mode = A.order_mode, kRing = A.kRing;
#ifdef MJ_DIAGNOSTIC
    const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr bool FUSED = MODE != 0;
    const int ac_total = FUSED ? A.ac_total_bytes : n_ac * kASlotBytes, dbits = FUSED ? A.dbits : kDBits, dsize = 1 << dbits;
    uint32_t *s_ac = reinterpret_cast<uint32_t *>(smem);                                         // the AC tables, back to back
    uint16_t *s_dc = reinterpret_cast<uint16_t *>(smem + (size_t)ac_total);                    // [n_dc][dsize]
    // (the flush moves blocks eight at a time and may read up to seven rows and positions past a wave's last: the next wave's,
    // or the slack behind the last wave's — never stored)
    const int lpw2 = lpw;
    const int wstride = (lpw2 * kRow + 3) & ~3;                                                  // dwords per wave, 16-byte multiple
    unsigned char *rows0 = smem + (size_t)ac_total + (size_t)n_dc * dsize * 2;
    uint32_t *s_blk = reinterpret_cast<uint32_t *>(rows0) + wave * wstride;
    uint64_t *s_base = reinterpret_cast<uint64_t *>(rows0 + (size_t)nw * wstride * 4 + 8 * kRow * 4) + wave * lpw2;
    // per-lane window on the lane's stream: kRing bytes, the stream's bytes at their offsets modulo kRing (see the bit reader)
    unsigned char *rings0 = smem + (((size_t)ac_total + (size_t)n_dc * dsize * 2 + (size_t)nw * wstride * 4 + 8 * kRow * 4 + (size_t)(nw * lpw2 + 8) * 8 + 127) & ~(size_t)127);

    // Which segment a lane takes.  Without a length list: the segments in blob order.  With one (restart segments whose
    // lengths the host knows), mode 2 deals the list out one segment per wave and round, so that the long ones sit in
    // different waves, each beside short ones: a wave is as slow as the lock-step of its lanes, and a lane with a long
    // segment mostly sets its wave's pace alone.  Mode 1 (neighbours of the list share a wave) is there to be measured.
    int64_t seg_id = MODE == 1 ? (int64_t)wg * A.wg_segs + wave * lpw + lane : ((int64_t)wg * nw + wave) * lpw + lane;
    if (by_length) {
        const int64_t n_waves = (int64_t)n_wg * nw, rank = order_mode == 2 ? (int64_t)lane * n_waves + ((int64_t)wg * nw + wave) : seg_id;
        seg_id = (lane < lpw && rank < n_segs) ? by_length[rank] : n_segs;
    }
    const bool have = lane < lpw && seg_id < n_segs && (MODE != 1 || wave * lpw + lane < A.wg_segs);
    DevSegment sg = segs[(have && !vsegs) ? seg_id : 0];
    DevVSeg vs{};
    if (vsegs) {
        vs = vsegs[have ? seg_id : 0];
        sg.image = vs.image; sg.mcu0 = vs.mcu0; sg.n_mcu = vs.n_mcu; sg.last = vs.last == 1;
    }
    const DevImage *im = images + sg.image;
    const int bpm = __builtin_amdgcn_readfirstlane(im->blocks_per_mcu);     // one sampling layout per plan
    const uint64_t comp_pk = *reinterpret_cast<const uint64_t *>(im->blk_comp);
    const uint64_t comp_pk_u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(comp_pk >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)comp_pk);
    // per-lane table numbers per component: index in the batch (for the canonical search) and LDS slot
    int dcG[3], acG[3];
    {
        int seen = 0;
        for (int b = 0; b < 8 && b < im->blocks_per_mcu; ++b) {
            const int c = im->blk_comp[b];
            if (!((seen >> c) & 1)) {
                seen |= 1 << c;
                const int dt = im->tab_index[im->blk_dc_slot[b]], at = im->tab_index[im->blk_ac_slot[b]];
                if (c == 0) { dcG[0] = dt; acG[0] = at; } else if (c == 1) { dcG[1] = dt; acG[1] = at; } else { dcG[2] = dt; acG[2] = at; }
            }
        }
        if (!(seen & 2)) { dcG[1] = dcG[0]; acG[1] = acG[0]; }
        if (!(seen & 4)) { dcG[2] = dcG[0]; acG[2] = acG[0]; }
    }
    int dcS[3], acS[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dcS[c] = (int)((dc_slot_pk >> (8 * dcG[c])) & 0xFF);
        acS[c] = (int)((ac_slot_pk >> (8 * acG[c])) & 0xFF);
    }
    const int n_mcu = have ? sg.n_mcu : 0;
    int max_mcu = n_mcu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_mcu = max(max_mcu, __shfl_xor(max_mcu, o));
    max_mcu = __builtin_amdgcn_readfirstlane(max_mcu);

    // ---- per-lane bit reader: bb = bit buffer (next bit = bit 63), bc = its fill, voff = byte offset of the next dword
    // of the stream, nxtw = that dword (read one step before it can be needed).  The dwords come from the lane's window in
    // LDS — the kRing bytes of its stream up to byte pf, each at its stream offset modulo kRing — which the lane keeps
    // topped up 16 bytes at a time, well ahead of the reader: a load from the stream itself in the symbol loop would put
    // an L2 round trip (every lane walks cache lines of its own, 270 of them per CU: L1 does not hold them) on the serial
    // path of every iteration, and its wait would also wait for the previous block's coefficient stores (vmcnt is in order).
    const unsigned char *streamb = reinterpret_cast<const unsigned char *>(stream);
    const uint32_t bit_sh = vsegs ? (uint32_t)vs.bit0 & 31u : 0u;
    const uint32_t voff0 = !have ? 0u : (vsegs ? vs.voff0 + ((uint32_t)vs.bit0 >> 5) * 4u : (uint32_t)(((sg.begin >> 2) + seg_id) * 4));
    const int nbits = !have ? 0 : (vsegs ? vs.bit_end - vs.bit0 : seg_bits[seg_id]);
    const uint32_t ringbase = lds_addr(rings0 + (size_t)(wave * lpw2 + (lane < lpw2 ? lane : 0)) * kRing);
    const uint64_t ring_lanes = lpw2 >= 64 ? ~0ull : (1ull << lpw2) - 1;      // lanes that own a window
    auto ring_u32 = [&](uint32_t off) { return *(const uint32_t __attribute__((address_space(3))) *)(uintptr_t)(ringbase + (off & (kRing - 4))); };
    uint64_t bb;
    uint32_t bc, voff, nxtw;
    // (voff and pf carry a per-lane rotation of the window, `rot`, on top of the stream offset: the lanes of a wave read and
    // fill their windows at similar offsets, which without it are the same LDS banks for all of them)
    const uint32_t rot = (uint32_t)(lane & 7) * 16u;
    uint32_t pf = (voff0 & ~15u) + rot;              // the window holds the stream's bytes [pf - kRing, pf) (minus rot)
    auto top_up = [&](uint32_t want_ahead) {         // synchronous: at start, and should a lane ever run low (it does not: the loop keeps ahead)
        while (lane < lpw2 && (int)(pf - voff) < (int)want_ahead) {
            const u32x4 c = *reinterpret_cast<const u32x4 *>(streamb + (pf - rot));
            *(u32x4 __attribute__((address_space(3))) *)(uintptr_t)(ringbase + (pf & (kRing - 16))) = c;
            pf += 16;
        }
    };
    auto seek = [&](uint32_t consumed_bits) {       // position the reader `consumed_bits` behind the (virtual) segment's first bit
        const uint32_t ab = bit_sh + consumed_bits;
        const uint32_t o = voff0 + (ab >> 5) * 4u, sh = ab & 31u;
        const uint32_t d0 = *reinterpret_cast<const uint32_t *>(streamb + o), d1 = *reinterpret_cast<const uint32_t *>(streamb + o + 4);
        bb = (((uint64_t)d0 << 32) | d1) << sh;
        bc = 64u - sh;
        voff = o + 8 + rot;
        nxtw = *reinterpret_cast<const uint32_t *>(streamb + o + 8);
    };
    seek(0);
    top_up(kRing - 16);
    auto consumed = [&]() { return (int)((voff - rot - voff0) * 8u) - (int)bc - (int)bit_sh; };
    const int64_t out_off = (im->block_off + (int64_t)sg.mcu0 * bpm) * 128;
    if (lane < lpw2) s_base[lane] = (uint64_t)out_off;
    int pred0 = vs.pred[0], pred1 = vs.pred[1], pred2 = vs.pred[2];        // zero for a restart segment (:900)
    int err = 0;
    uint32_t *myblk = s_blk + (lane < lpw2 ? lane : 0) * kRow;
    int16_t *myblk16 = reinterpret_cast<int16_t *>(myblk);
    const uint32_t mybase = lds_addr(myblk);
    const uint32_t lastB = mybase + 126u, storeB = mybase + 127u;
    const uint32_t ac_base = lds_addr(s_ac);
    const uint32_t c7f = 0x7FFFFFFFu, c124 = (uint32_t)kRing - 4, c112 = (uint32_t)kRing - 16, c96 = (uint32_t)kRing - 32;

    // flush geometry: lane (slot, part) moves the 8 coefficients of natural positions 8*part .. 8*part+7 of block
    // slot + 8*it — 16 bytes; they are read from their zig-zag slots, so the block lands in HBM in the natural [v][u]
    // order stage 2 wants ([u][v] when the plan is transposed)
    const int fslot = lane >> 3, fpart = lane & 7;
    uint32_t fa[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int nat = fpart * 8 + u;
        fa[u] = lds_addr(s_blk + fslot * kRow) + 2u * c_zz_of_nat_13[tr ? ((nat & 7) << 3 | nat >> 3) : nat];
    }
    const uint64_t full_mask = lpw >= 64 ? ~0ull : (1ull << lpw) - 1;
    const uint32_t fb_addr = lds_addr(s_base + fslot);

#ifdef MJ_X_STAMP
    uint32_t dbg_d[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_iter = 0, dbg_w[3] = {0, 0, 0};
    uint64_t dbg_in = 0;
    uint32_t dbg_sym = 0, dbg_flag = 0;
    uint64_t dbg_ac = 0, dbg_fl = 0;
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int m = 0; m < max_mcu; ++m) {
        // The arbiter serves a SIMD's oldest wave first: with the (three) waves of a SIMD at one priority the first-dispatched
        // ones finish 12 % ahead of the last (wave end times 2.52 / 2.66 / 2.87 ms, profiles/r04b) and the launch lasts as long
        // as the last.  Taking turns at the priorities — every wave spends the same share of its MCUs at each — lets them
        // finish together: 2.98 -> 2.70 ms.  (Waves 4g .. 4g+3 of a workgroup are the g-th wave of their SIMDs.)
#ifndef MJ_X_NOPRIOROT     // (make XFLAGS=-DMJ_X_NOPRIOROT: the A/B build)
        if ((m & 7) == 0) {
            const int groups = (nw + 3) >> 2;
            const int turn = groups > 1 ? ((wave >> 2) + (m >> 3)) % groups : 0;
            // (fused.hip: the walk is the launch's critical path — its waves stay above the consumer waves, which run at 0)
#ifdef MJ_X_FUSED_PRIO3
            const int pr = FUSED ? 3 : turn;
#else
            const int pr = FUSED ? (turn & 1) + 2 : turn;
#endif
            if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
        }
#endif
        const bool in_mcu = m < n_mcu;
        for (int b = 0; b < bpm; ++b) {
            const int comp = (int)((comp_pk_u >> (8 * b)) & 0xFF);          // wave-uniform
            const int dcs = comp == 0 ? dcS[0] : (comp == 1 ? dcS[1] : dcS[2]);
            const int acs = comp == 0 ? acS[0] : (comp == 1 ? acS[1] : acS[2]);
            const int dcg = comp == 0 ? dcG[0] : (comp == 1 ? dcG[1] : dcG[2]);
            const int acg = comp == 0 ? acG[0] : (comp == 1 ? acG[1] : acG[2]);
            const bool act = in_mcu && err == 0;

            // ---- DC (:810-820): one symbol per lane, straight-line
            {
                const bool want = bc <= 32u;
                const uint32_t t = want ? nxtw : 0u;
                bb |= (uint64_t)t << ((32u - bc) & 63u);
                const uint32_t inc = want ? 4u : 0u;
                voff += inc;
                bc += inc * 8u;
                if (want) nxtw = ring_u32(voff);
                // (the loop below reads at most 4 bytes per iteration and brings in 16 per two: a lane cannot run its window dry,
                // but nothing is lost by looking)
                // (32, not 24: the loop's first turn has no refill in flight and can take six dwords when an unresolved entry of
                // 29+ bits — value sizes above 12, which the reference accepts — is followed by two-symbol rounds)
                if (__builtin_amdgcn_ballot_w64(lane < lpw2 && (int)(pf - voff) < 32) != 0) top_up(48);
            }
            uint32_t pB;
            {
                const uint32_t p16 = (uint32_t)(bb >> 48);
                const int e = s_dc[dcs * dsize + (p16 >> (16 - dbits))];
                int len = e >> 8, s = e & 0xFF;
                if (__builtin_amdgcn_ballot_w64(act && len == 0) != 0) {                              // code longer than 11 bits: rare
                    if (act && len == 0) {
                        const int r = canon_code(huff + dcg, p16, dbits + 1);
                        len = r < 0 ? 0 : r >> 8; s = r < 0 ? 255 : r & 0xFF;
                    }
                }
                const bool bad = act && s > 16;
                err = bad ? MJ_ST_BAD_CODE : err;
                const bool ok = act && !bad;
                const int ln = ok ? len : 0, sz = ok ? s : 0;
                const uint32_t hw = (uint32_t)(bb >> 32) << ln;             // ln + sz <= 32 <= bc
                const uint32_t rawv = (hw >> 1) >> (31 - sz);
                bb <<= ln + sz;
                bc -= (uint32_t)(ln + sz);
                const int pred = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
                const int dcv = (int)(int16_t)(extend13(rawv, sz) + pred);
                pred0 = (ok && comp == 0) ? dcv : pred0;
                pred1 = (ok && comp == 1) ? dcv : pred1;
                pred2 = (ok && comp == 2) ? dcv : pred2;
                myblk16[ok ? 0 : 64] = (int16_t)dcv;                           // 64 = the row's pad slot, never flushed
                pB = ok ? mybase : lastB;                                      // position of the last coefficient written; lastB = lane is done
            }
            // ---- AC (:833-866): until every lane is at its end of block.  See the header for the entry formats.
            // (where the block's AC table sits and how wide its main level is: wave-uniform, in registers for the asm)
            const uint32_t lutb = ac_base + (FUSED ? (uint32_t)(acs == 0 ? A.ac_off[0] : (acs == 1 ? A.ac_off[1] : (acs == 2 ? A.ac_off[2] : A.ac_off[3])))
                                                   : (uint32_t)acs * (uint32_t)kASlotBytes);
            const uint32_t ab_v = FUSED ? (uint32_t)(acs == 0 ? A.ac_bits[0] : (acs == 1 ? A.ac_bits[1] : (acs == 2 ? A.ac_bits[2] : A.ac_bits[3]))) : 13u;
            const uint32_t ash_v = 32u - ab_v, sb_v = 16u - ab_v;
            uint32_t e_last = 0xFFu;                 // the lane's latest entry; 0xFF = "nothing a correction below could use"
#ifdef MJ_X_STAMP
            const uint64_t dbg_a0 = __builtin_amdgcn_s_memtime();
#endif
            for (;;) {
                uint64_t pend, nx = nxtw, tmp64;
                uint32_t t0, t1, t2, t3, t4, t5, t6, ew;
                u32x4 chunk, chunk2;
#ifdef MJ_X_STAMP
                const uint64_t dbg_i0 = __builtin_amdgcn_s_memtime();
#endif
                // LUT address from the 13 bits on top of the buffer, and the read
#define MJ_LOOK13 \
    "v_bfe_u32 %[t0], v3, %[ash], %[abits]\n\t"         \
    "v_lshl_add_u32 %[t0], %[t0], 2, %[lutb]\n\t"       \
    "ds_read_b32 %[e], %[t0]\n\t"
                // the entry applied: position, buffer, count; exec keeps the lanes that are still inside their block AFTER this
                // symbol — they are also the ones that store it (a lane whose symbol lands on coefficient 63 leaves here and
                // stores it after the loop; an entry that is not resolved moves its lane out by 128+ and consumes nothing)
#ifdef MJ_X_STAMP
#define MJ_CNT13
#else
#define MJ_CNT13
#endif
#define MJ_CORE13 MJ_CNT13 \
    "v_mov_b32 %[ew], %[e]\n\t"                                                                                  \
    "v_add_u32_sdwa %[pB], %[pB], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
    "v_lshlrev_b64 v[2:3], %[e], v[2:3]\n\t"                                                                    \
    "v_sub_u32_sdwa %[bc], %[bc], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
    "v_cmpx_gt_u32 %[lastB], %[pB]\n\t"
#define MJ_WRITE13 "ds_write_b16_d16_hi %[pB], %[ew]\n\t"
                // refill: lanes whose buffer is at most half full take the next dword and read the one after it from their window
#define MJ_REFILL13 \
    "s_mov_b64 s[42:43], exec\n\t"                        \
    "v_cmpx_ge_u32 32, %[bc]\n\t"                         \
    "v_sub_u32 %[t0], 32, %[bc]\n\t"                      \
    "v_lshlrev_b64 v[4:5], %[t0], v[6:7]\n\t"             \
    "v_or_b32 v3, v3, v5\n\t"                             \
    "v_mov_b32 v2, v4\n\t"                                \
    "v_add_u32 %[bc], 32, %[bc]\n\t"                      \
    "v_add_u32 %[voff], 4, %[voff]\n\t"                   \
    "v_and_or_b32 %[t0], %[voff], %[c124], %[ring]\n\t"   \
    "ds_read_b32 v6, %[t0]\n\t"                           \
    "s_mov_b64 exec, s[42:43]\n\t"
                // two symbols of the lanes in exec (a resolved symbol is at most 13 bits: after a refill there are bits for
                // two); the first one's store goes out behind the second one's read
#define MJ_PAIR13 \
    MJ_LOOK13 "s_waitcnt lgkmcnt(0)\n\t" MJ_CORE13 MJ_LOOK13 MJ_WRITE13 "s_waitcnt lgkmcnt(1)\n\t" MJ_CORE13 MJ_WRITE13
                // any entry that was not resolved?  (every lane of the wave is looked at: lanes that are done keep a clean entry)
                // the window's upkeep, once per two iterations and for every lane that owns one: the 16 bytes asked for last
                // time go into the window, the next 16 are asked for if they fit (they overwrite what lies kRing behind them)
#define MJ_WINDOW_IN13 \
    "s_waitcnt vmcnt(0)\n\t"                              \
    "s_mov_b64 exec, s[52:53]\n\t"                        \
    "v_and_or_b32 %[t0], %[pf], %[c112], %[ring]\n\t"     \
    "ds_write_b128 %[t0], v[8:11]\n\t"                    \
    "v_add_u32 %[pf], 16, %[pf]\n\t"                      \
    "s_mov_b64 exec, s[56:57]\n\t"                        \
    "v_and_or_b32 %[t0], %[pf], %[c112], %[ring]\n\t"     \
    "ds_write_b128 %[t0], v[12:15]\n\t"                   \
    "v_add_u32 %[pf], 16, %[pf]\n\t"
#ifdef MJ_X_STAMP
#define MJ_T(k) "s_memtime s[58:59]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s72, s58, s60\n\ts_mov_b32 s60, s58\n\ts_add_u32 s" #k ", s" #k ", s72\n\t"
#else
#define MJ_T(k)
#endif
                asm volatile(
                    "s_mov_b64 s[40:41], exec\n\t"
#ifdef MJ_X_STAMP
                    "s_mov_b32 s62, 0\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s64, 0\n\ts_mov_b32 s65, 0\n\ts_mov_b32 s66, 0\n\ts_mov_b32 s67, 0\n\ts_mov_b32 s68, 0\n\ts_mov_b32 s69, 0\n\ts_mov_b32 s70, 0\n\ts_mov_b32 s71, 0\n\t"
                    "s_memtime s[60:61]\n\ts_waitcnt lgkmcnt(0)\n\t"
#endif
                    "s_mov_b64 %[pend], 0\n\t"
                    "s_mov_b64 s[52:53], 0\n\t"                // lanes with 16 bytes on their way
                    "s_mov_b64 s[56:57], 0\n\t"                // ... and with 16 more behind those
#ifdef MJ_X_STAMP
                    "s_mov_b32 s55, 0\n\t"
#endif
                    "v_cmpx_gt_u32 %[lastB], %[pB]\n"
                    // One turn of the loop = the window's upkeep and the look for entries that were not resolved, both on every
                    // lane of the wave, then four refill + two-symbol rounds of the lanes still inside their block
                    "L_loop%=:\n\t"
                    MJ_T(62)
                    "s_mov_b64 s[44:45], exec\n\t"             // the lanes that go on
                    "s_mov_b64 exec, %[rl]\n\t"
                    "v_cmp_gt_i16 vcc, 0, %[e]\n\t"            // (lanes that are done keep a clean entry)
                    "s_cbranch_vccnz L_open%=\n"
                    "L_back%=:\n\t"
                    "s_cmp_eq_u64 s[44:45], 0\n\t"
                    "s_cbranch_scc1 L_done%=\n\t"
                    "s_cmp_eq_u64 s[52:53], 0\n\t"             // nothing on its way (first turn): no wait — it would wait for the
                    "s_cbranch_scc1 L_ask%=\n\t"               // previous block's coefficient stores as well
                    MJ_T(69)
                    "s_waitcnt vmcnt(0)\n\t"
                    MJ_T(70)
                    MJ_WINDOW_IN13
                    "s_mov_b64 exec, %[rl]\n"
                    MJ_T(71)
                    "L_ask%=:\n\t"
                    "v_sub_u32 %[t0], %[pf], %[voff]\n\t"
                    "v_cmpx_ge_u32 %[c112], %[t0]\n\t"       // room for 16 bytes (they overwrite what lies a window behind them)
                    "v_sub_u32 %[t1], %[pf], %[rot]\n\t"
                    "global_load_dwordx4 v[8:11], %[t1], %[sbase]\n\t"
                    "s_mov_b64 s[52:53], exec\n\t"
                    "v_cmpx_ge_u32 %[c96], %[t0]\n\t"        // ... and for 16 more
                    "global_load_dwordx4 v[12:15], %[t1], %[sbase] offset:16\n\t"
                    "s_mov_b64 s[56:57], exec\n\t"
                    "s_mov_b64 exec, s[44:45]\n\t"
                    MJ_T(63)
                    MJ_REFILL13
                    MJ_T(64)
#ifdef MJ_X_STAMP
                    MJ_LOOK13 "s_waitcnt lgkmcnt(0)\n\t" MJ_T(65) MJ_CORE13 MJ_LOOK13 MJ_WRITE13 "s_waitcnt lgkmcnt(1)\n\t" MJ_T(66) MJ_CORE13 MJ_WRITE13 MJ_T(67)
#else
                    MJ_PAIR13
#endif
#define MJ_ROUND13 "s_cbranch_execz L_loop%=\n\t" MJ_REFILL13 MJ_PAIR13
#ifndef MJ_X_ROUNDS
#define MJ_X_ROUNDS 6
#endif
                    MJ_ROUND13 MJ_ROUND13 MJ_ROUND13
#if MJ_X_ROUNDS >= 6
                    MJ_ROUND13 MJ_ROUND13
#endif
#if MJ_X_ROUNDS >= 8
                    MJ_ROUND13 MJ_ROUND13
#endif
                    MJ_T(68)
#ifdef MJ_X_STAMP
                    "s_add_u32 s55, s55, 1\n\t"
#endif
                    "s_branch L_loop%=\n"
                    // ---- entries that are not resolved (0.4 % of the symbols): the lanes of vcc.  Byte 1 = 0x80 | 0x40 if the
                    // code is longer than 13 bits (then the high word is where its second-level table starts) | run + 1
                    // (0 = end of block); byte 2 = code length (0 = no such code); byte 3 = 31 - size
                    "L_open%=:\n\t"
                    "s_mov_b64 s[46:47], vcc\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "v_mov_b32 %[t5], %[e]\n\t"
                    "v_sub_u32_sdwa %[pB], %[pB], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"   // back where it was
                    "v_mov_b32 %[e], 0xff\n\t"
                    "v_cmpx_le_u32 31, %[bc]\n\t"              // bits for the longest symbol?  else again after the next refill
                    "s_andn2_b64 s[46:47], s[46:47], exec\n\t" // ... those lanes simply go on
                    "s_cbranch_execz L_hend%=\n\t"
                    "v_and_b32 %[t1], 0x4000, %[t5]\n\t"
                    "v_cmp_ne_u32 vcc, 0, %[t1]\n\t"
                    "s_cbranch_vccz L_arith%=\n\t"
                    "s_and_saveexec_b64 s[50:51], vcc\n\t"
                    "v_bfe_u32 %[t0], v3, 16, %[sbits]\n\t"   // the 16 - AB bits behind the AB of the index
                    "v_lshlrev_b32 %[t0], 2, %[t0]\n\t"
                    "v_add_u32_sdwa %[t0], %[t0], %[t5] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                    "v_add_u32 %[t0], %[t0], %[lutb]\n\t"
                    "ds_read_b32 %[t5], %[t0]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, s[50:51]\n"
                    "L_arith%=:\n\t"
                    "v_bfe_u32 %[t1], %[t5], 16, 8\n\t"        // code length
                    "v_cmp_eq_u32 vcc, 0, %[t1]\n\t"
                    "s_cbranch_vccnz L_rare%=\n\t"             // no such code (a damaged file): the canonical search says so
                    "v_bfe_u32 %[t6], %[t5], 8, 5\n\t"         // run + 1, 0 = end of block
                    "v_lshlrev_b32 %[t6], 1, %[t6]\n\t"
                    "v_cmp_eq_u32 vcc, 0, %[t6]\n\t"
                    "v_mov_b32 %[t2], 0x7f\n\t"
                    "v_cndmask_b32_e64 %[t6], %[t6], %[t2], vcc\n\t"
                    "v_lshlrev_b32 %[t2], %[t1], v3\n\t"       // value bits on top
                    "v_lshrrev_b32 %[t3], 1, %[t2]\n\t"
                    "v_lshrrev_b32_sdwa %[t3], %[t5], %[t3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"   // raw
                    "v_lshrrev_b32_sdwa %[t4], %[t5], %[c7f] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"  // 2^size - 1
                    "v_cmp_le_i32 vcc, 0, %[t2]\n\t"           // leading value bit 0: value = raw - (2^size - 1)  (:1636-1646)
                    "v_cndmask_b32_e64 %[t2], 0, %[t4], vcc\n\t"
                    "v_sub_u32 %[t3], %[t3], %[t2]\n\t"        // the coefficient
                    "v_bcnt_u32_b32 %[t4], %[t4], %[t1]\n\t"   // code + value bits
                    "v_add_u32 %[pB], %[pB], %[t6]\n\t"
                    "v_cmp_gt_u32 vcc, %[storeB], %[pB]\n\t"   // past the block: the value bits stay unread (:855-856)
                    "v_cndmask_b32_e64 %[t4], %[t1], %[t4], vcc\n\t"
                    "v_lshlrev_b64 v[2:3], %[t4], v[2:3]\n\t"
                    "v_sub_u32 %[bc], %[bc], %[t4]\n\t"
                    "v_cmpx_gt_u32 %[storeB], %[pB]\n\t"
                    "ds_write_b16 %[pB], %[t3]\n\t"
                    "v_cmpx_gt_u32 %[lastB], %[pB]\n"
                    "L_hend%=:\n\t"
                    "s_or_b64 s[44:45], s[44:45], exec\n\t"
                    "s_or_b64 s[44:45], s[44:45], s[46:47]\n\t"
                    "s_mov_b64 exec, %[rl]\n\t"
                    "s_branch L_back%=\n"
                    "L_rare%=:\n\t"
                    "s_mov_b64 %[pend], exec\n"
                    "L_done%=:\n\t"
                    MJ_WINDOW_IN13
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, s[40:41]\n\t"
#ifdef MJ_X_STAMP
                    "v_add_u32 %[di], s55, %[di]\n\t"
                    "v_add_u32 %[d0], s62, %[d0]\n\tv_add_u32 %[d1], s63, %[d1]\n\tv_add_u32 %[d2], s64, %[d2]\n\tv_add_u32 %[d3], s65, %[d3]\n\t"
                    "v_add_u32 %[d4], s66, %[d4]\n\tv_add_u32 %[d5], s67, %[d5]\n\tv_add_u32 %[d6], s68, %[d6]\n\t"
                    "v_add_u32 %[d7], s69, %[d7]\n\tv_add_u32 %[d8], s70, %[d8]\n\tv_add_u32 %[d9], s71, %[d9]\n\t"
#endif
                    : "+{v[2:3]}"(bb), "+{v[6:7]}"(nx), "=&{v[4:5]}"(tmp64), "=&{v[8:11]}"(chunk), "=&{v[12:15]}"(chunk2), [bc] "+v"(bc), [pB] "+v"(pB), [e] "+v"(e_last),
                      [voff] "+v"(voff), [pf] "+v"(pf),
                      [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
                      [ew] "=&v"(ew), [pend] "=&s"(pend)
#ifdef MJ_X_STAMP
                      , [di] "+v"(dbg_iter), [d0] "+v"(dbg_d[0]), [d1] "+v"(dbg_d[1]), [d2] "+v"(dbg_d[2]), [d3] "+v"(dbg_d[3]), [d4] "+v"(dbg_d[4]), [d5] "+v"(dbg_d[5]), [d6] "+v"(dbg_d[6]), [d7] "+v"(dbg_w[0]), [d8] "+v"(dbg_w[1]), [d9] "+v"(dbg_w[2])
#endif
                    : [lastB] "v"(lastB), [storeB] "v"(storeB), [lutb] "v"(lutb), [sbase] "s"(streamb), [c7f] "v"(c7f),
                      [ash] "v"(ash_v), [abits] "v"(ab_v), [sbits] "v"(sb_v),
                      [ring] "v"(ringbase), [c124] "v"(c124), [c112] "v"(c112), [c96] "v"(c96), [rl] "s"(ring_lanes), [rot] "v"(rot)
                    : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72");
#undef MJ_REFILL13
#undef MJ_PAIR13
#undef MJ_WINDOW_IN13
#undef MJ_LOOK13
#undef MJ_CORE13
#undef MJ_WRITE13
                nxtw = (uint32_t)nx;
#ifdef MJ_X_STAMP
                dbg_in += __builtin_amdgcn_s_memtime() - dbg_i0;
                dbg_d[7] += 1;
#endif
                if (pend == 0) break;
                // rare: no code at all in some lane (a damaged file); the lanes of `pend` — back at their positions, nothing
                // consumed — take the canonical search (jpeg_decoder.py:366-377 semantics), which says so or decodes the symbol
                if ((pend >> lane) & 1) {
                    const uint32_t hi = (uint32_t)(bb >> 32);
                    const int r = canon_code(huff + acg, hi >> 16, 1);
#ifdef MJ_X_STAMP
                    if (wg < 4 && m < 30) printf("rare: lane %d m %d b %d hi %08x r %x bc %u pB-mybase %d e_last %x\n", lane, m, b, hi, r, bc, (int)(pB - mybase), e_last);
#endif
                    if (r < 0) {
                        err = MJ_ST_BAD_CODE;
                        pB = lastB + 1u;                                          // done (and nothing to correct below)
                    } else {
                        const int ln = r >> 8, hv = r & 0xFF;
                        const uint32_t nB = pB + (hv == 0 ? 127u : 2u * (uint32_t)((hv >> 4) + 1));
                        const bool inblk = nB < storeB;
                        const int n = inblk ? (hv & 15) : 0;
                        const int tot = ln + n;                                   // <= 31 <= bc
                        const uint32_t raw = __builtin_amdgcn_ubfe(hi, (uint32_t)(32 - tot), (uint32_t)n);
                        if (inblk) myblk16[(nB - mybase) >> 1] = (int16_t)extend13(raw, n);
                        bb <<= tot;
                        bc -= (uint32_t)tot;
                        pB = nB;
                    }
                }
            }
            if constexpr (FUSED) {
                // The AC loop leaves with vmcnt = 0 (its last window upkeep waits for everything in flight): every coefficient
                // store this wave issued before it — the whole of MCU m - 1 once block 0 of MCU m is through — has been
                // acknowledged by L2.  The consumer waves of this workgroup (fused.hip) read the wave's progress from LDS
                // and then the blocks through the same CU's vector cache: workgroup scope, nothing to invalidate.
                if (b == 0) {
                    uint32_t t_prog;
                    if constexpr (MODE == 1)
                        asm volatile("v_mov_b32 %0, %1\n\tds_write_b32 %2, %0" : "=&v"(t_prog) : "s"(progress_base + (uint32_t)m), "v"(progress_addr) : "memory");
                    else        // (write-through stores that have been waited for are in memory: the flag may follow)
                        asm volatile("v_mov_b32 %0, %1\n\tglobal_store_dword %2, %0, off sc1" : "=&v"(t_prog) : "s"((uint32_t)m), "v"(A.progress_global + (wg * nw + wave)) : "memory");
                }
            }
            {
                const bool resolved = act && (e_last & 0xFFu) != 0xFFu;       // the lane's last symbol came straight out of a resolved entry
                // ... onto coefficient 63: the loop left the store to us (its lanes leave before they store)
                if (resolved && pB == lastB) myblk16[63] = (int16_t)(e_last >> 16);
                // ... past the block: the entry has consumed its value bits, which the reference leaves unread (:855-856).  The
                // final position is even then (only the end-of-block symbol moves by an odd amount).  Damaged files only.
                const bool ovf = resolved && pB > lastB && !(pB & 1u);
                if (__builtin_amdgcn_ballot_w64(ovf) != 0) {
                    if (ovf) {
                        const int at = consumed() - (int)(e_last & 0xFFu);            // where that symbol began
                        seek((uint32_t)at);
                        const int r = canon_code(huff + acg, (uint32_t)(bb >> 48), 1);
                        seek((uint32_t)(at + (r < 0 ? 0 : r >> 8)));
                        pf = voff & ~15u;                                             // (the window again, from there)
                        top_up(kRing - 16);
                    }
                }
            }
            // a segment that consumed more bits than it has is corrupt (it has been reading its neighbour's bytes)
            err = (act && err == 0 && consumed() > nbits) ? MJ_ST_OVERRUN : err;
#ifdef MJ_X_STAMP
            const uint64_t dbg_a1 = __builtin_amdgcn_s_memtime();
            dbg_ac += dbg_a1 - dbg_a0;
#endif
            // ---- round of blocks done: LDS -> HBM, eight blocks (8 x 128 bytes) per store instruction, and clear
            const uint64_t act_mask = __ballot(in_mcu);
            const uint32_t blk_byte = (uint32_t)(m * bpm + b) * 128u;    // same for every lane (same layout)
            unsigned char *dst0 = reinterpret_cast<unsigned char *>(coef) + blk_byte + fpart * 16;
#ifdef MJ_X_SPARSE_ST    // probe (holes in the store): rows 4..7 of every chroma block are not written — what a half-block store could save on the write side
            const int fslot_x = (comp != 0 && fpart >= 4) ? fslot + 4096 : fslot;
#else
            const int fslot_x = fslot;
#endif
            if (act_mask == full_mask) {
                // every lane of the wave has a block (all but a segment's last rounds): by hand — per eight blocks eight 16-bit
                // reads and four packs (a d16 load clears the other half of its register on this chip), the block's address from
                // LDS, one 16-byte store per lane
                u32x4 fd, fh;
                uint64_t fad;
#define MJ_FLUSH8(it, SC) \
    "s_sub_u32 s42, %[lpw], " #it "*8\n\t"                                   \
    "v_cmpx_gt_u32 s42, %[fslot]\n\t"                                      \
    "ds_read_u16 v10, %[fa0] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v16, %[fa1] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v11, %[fa2] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v17, %[fa3] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v12, %[fa4] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v18, %[fa5] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v13, %[fa6] offset:" #it "*1056\n\t"                      \
    "ds_read_u16 v19, %[fa7] offset:" #it "*1056\n\t"                      \
    "ds_read_b64 v[14:15], %[fb] offset:" #it "*64\n\t"                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                             \
    "v_lshl_or_b32 v10, v16, 16, v10\n\t"                                  \
    "v_lshl_or_b32 v11, v17, 16, v11\n\t"                                  \
    "v_lshl_or_b32 v12, v18, 16, v12\n\t"                                  \
    "v_lshl_or_b32 v13, v19, 16, v13\n\t"                                  \
    "v_lshl_add_u64 v[14:15], v[14:15], 0, %[dst]\n\t"                     \
    "global_store_dwordx4 v[14:15], v[10:13], off" SC "\n\t"
#define MJ_FLUSH_MORE(it, SC) "s_cmp_le_u32 %[lpw], " #it "*8\n\ts_cbranch_scc1 L_fend%=\n\t" MJ_FLUSH8(it, SC)
#define MJ_FLUSH_ALL(SC) \
                asm volatile(                                                                                                                   \
                    "s_mov_b64 s[40:41], exec\n\t"                                                                                               \
                    MJ_FLUSH8(0, SC) MJ_FLUSH_MORE(1, SC) MJ_FLUSH_MORE(2, SC) MJ_FLUSH_MORE(3, SC) MJ_FLUSH_MORE(4, SC) MJ_FLUSH_MORE(5, SC)     \
                    MJ_FLUSH_MORE(6, SC) MJ_FLUSH_MORE(7, SC)                                                                                   \
                    "L_fend%=:\n\t"                                                                                                              \
                    "s_mov_b64 exec, s[40:41]\n\t"                                                                                               \
                    : "=&{v[10:13]}"(fd), "=&{v[14:15]}"(fad), "=&{v[16:19]}"(fh)                                                               \
                    : [lpw] "s"(lpw), [fslot] "v"(fslot_x), [fa0] "v"(fa[0]), [fa1] "v"(fa[1]), [fa2] "v"(fa[2]), [fa3] "v"(fa[3]), [fa4] "v"(fa[4]), \
                      [fa5] "v"(fa[5]), [fa6] "v"(fa[6]), [fa7] "v"(fa[7]), [fb] "v"(fb_addr), [dst] "v"((uint64_t)(uintptr_t)dst0)            \
                    : "memory", "vcc", "scc", "s40", "s41", "s42")
                if constexpr (MODE == 2) { MJ_FLUSH_ALL(" sc1"); } else { MJ_FLUSH_ALL(""); }
#undef MJ_FLUSH_ALL
#undef MJ_FLUSH8
#undef MJ_FLUSH_MORE
            } else
            for (int it = 0; it * 8 < lpw; ++it) {
                if (((act_mask >> (8 * it)) & 0xFF) == 0) continue;      // uniform
                const uint32_t ro = (uint32_t)it * (8u * kRow * 4u);
                auto rd = [&](int u) { return (uint32_t)*(const uint16_t __attribute__((address_space(3))) *)(uintptr_t)(fa[u] + ro); };
                uint4 v;
                v.x = rd(0) | (rd(1) << 16);
                v.y = rd(2) | (rd(3) << 16);
                v.z = rd(4) | (rd(5) << 16);
                v.w = rd(6) | (rd(7) << 16);
                const int o = it * 8 + fslot;
                if (((act_mask >> o) & 1) && fslot_x == fslot) {
                    if constexpr (MODE == 2) {
                        const u32x4 vv = {v.x, v.y, v.z, v.w};
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst0 + s_base[o]), "v"(vv) : "memory");
                    } else {
                        *reinterpret_cast<uint4 *>(dst0 + s_base[o]) = v;
                    }
                }
            }
            for (int i = lane; i < wstride / 4; i += 64) reinterpret_cast<uint4 *>(s_blk)[i] = make_uint4(0, 0, 0, 0);
#ifdef MJ_X_STAMP
            dbg_fl += __builtin_amdgcn_s_memtime() - dbg_a1;
#endif
        }
    }
#ifdef MJ_X_STAMP
    if (lane == 0) {
        atomicAdd(&g_dbg13[6], (unsigned long long)dbg_in);
        atomicAdd(&g_dbg13[1], (unsigned long long)dbg_iter);
        for (int i = 0; i < 7; ++i) atomicAdd(&g_dbg13[9 + i], (unsigned long long)dbg_d[i]);
        atomicAdd(&g_dbg13[8], (unsigned long long)dbg_d[7]);
        atomicAdd(&g_dbg13[0], (unsigned long long)dbg_w[0]); atomicAdd(&g_dbg13[7], (unsigned long long)dbg_w[1]);
    }
    if (lane == 0) {
        atomicAdd(&g_dbg13[2], (unsigned long long)dbg_ac);
        atomicAdd(&g_dbg13[3], (unsigned long long)dbg_fl);
        atomicAdd(&g_dbg13[4], (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_t0));
        atomicAdd(&g_dbg13[5], 1ull);
    }
#endif

#ifdef MJ_DIAGNOSTIC
    if (lane == 0 && wg < 256 && wave < 16) {
        unsigned long long *o = g_dbg13_waves + ((size_t)wg * 16 + wave) * 3;
        o[0] = dbg_r0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = 0;
    }
#endif
    if (have) {
        const int left = nbits - consumed();
        if (!err && vsegs && vs.last == 0 && left != 0) err = MJ_ST_DESYNC;     // a virtual segment ends exactly where the next starts
        if (!err && (vsegs ? vs.last == 2 : !sg.last) && left >= 8) err = MJ_ST_DESYNC;
        if (err) atomicMax(status + sg.image, err);
    }
    if constexpr (FUSED) {      // the last MCU's blocks: in memory before the wave says "everything"
        uint32_t t_prog;
        if constexpr (MODE == 1)
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %2\n\tds_write_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t_prog)
                         : "v"(progress_addr), "s"(progress_base + (1u << kFusedPassShift)) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, 0x7fffffff\n\tglobal_store_dword %1, %0, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t_prog)
                         : "v"(A.progress_global + (wg * nw + wave)) : "memory");
    }
}

}  // namespace lanes13
}  // namespace mj
