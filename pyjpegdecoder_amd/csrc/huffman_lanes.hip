// Stage 1, lane-parallel form — baseline Huffman entropy decode with one restart segment PER LANE.
//
// Same contract and the same output as huffman.hip (one segment per wavefront); the two are A/B-tested
// against each other and against the reference's coefficients.  Why a second form: the wave-per-segment
// walker is wave-uniform code, so it lives on the scalar unit (one instruction per cycle per CU) and
// rocprof shows it issue-bound there (SALU:VALU 4:1, 55 % of wave cycles waiting to issue).  Batches of
// DRI-coded images offer tens of thousands of independent segments, enough to give every lane its own.
//
// Shape of the kernel (256 threads = 4 waves x 64 segments):
//   * lock-step at block granularity: all lanes decode block b of MCU m together (the component, hence
//     which table, is wave-uniform; all images of a plan share one sampling layout).  One DC symbol, then
//     AC symbols until every lane has reached its end of block; lanes that finish early idle.
//   * per-lane bit reader: the input is the stream stage 0 (destuff.hip) prepared — the segment's bytes with the
//     ones the reference's get_bits drops after a 0xFF (jpeg_decoder.py:676-677) already removed, as aligned
//     big-endian dwords — so a refill is one dword load (issued an iteration before it can be needed), a shift
//     and an or; reading past the segment's end is detected by bit count.
//   * Huffman tables: an 11-bit primary LUT per table, ALL tables of the batch resident in LDS (this form
//     is chosen when the batch has at most kMaxLaneTables distinct tables, e.g. everybody uses the Annex-K
//     tables); AC tables are stored as (length, zero run, size) with the end-of-block symbol as a run of 64;
//     codes longer than 11 bits take a per-lane canonical search (rare).
//   * coefficients: each lane owns a 128-byte block in LDS (row stride 132 B -> conflict-free scatter of
//     "coefficient kk of every lane"); a finished round of 64 blocks leaves as 32 store instructions that each
//     write two full 128-byte lines, in the natural [v][u] order stage 2 reads.
#include <stdio.h>
#include <stdlib.h>

#include "mijpeg_internal.h"

namespace mj {

#ifdef MJ_DIAGNOSTIC
__device__ unsigned long long g_dbg_lanes[4];        // MJ_DEBUG_STAGE1=3: cycles per loop segment, summed over waves
void dbg_lanes_report() {
    unsigned long long h[4] = {0, 0, 0, 0}, z[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_lanes), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_lanes), z, sizeof(z));
    const double tot = (double)h[0] + (double)h[1] + (double)h[2] + (double)h[3];
    if (tot > 0)
        fprintf(stderr, "[mijpeg diag] lanes kernel, share of wave time: loop test + refill %.1f %%, first symbol %.1f %%, second symbol %.1f %%, "
                        "DC + flush + clear %.1f %%\n", 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot);
}
#endif

namespace {

constexpr int kLBits = 11;
constexpr int kLSize = 1 << kLBits;
constexpr int kBlkStride = 33;   // dwords per lane block in LDS (32 + 1 pad)

// natural position (v*8+u) -> zig-zag index (the inverse of the table below)
__constant__ uint8_t c_zz_of_nat_l[64] = {
    0,  1,  5,  6, 14, 15, 27, 28,  2,  4,  7, 13, 16, 26, 29, 42,
    3,  8, 12, 17, 25, 30, 41, 43,  9, 11, 18, 24, 31, 40, 44, 53,
   10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
   21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
// Per-lane bit reader state (plain scalars: the decode loop is straight-line, select-based code):
//   bb/bc   bit buffer (next bit = bit 63) and its fill
//   voff    byte offset (from the stream buffer's start) of the next dword; nxtw = that dword, loaded one
//           iteration before it can be needed
struct LaneBits {
    uint64_t bb;
    uint32_t voff, nxtw;
    int bc;
};

// Codes longer than the 11-bit LUT (rare): canonical search (jpeg_decoder.py:366-377 semantics).
// Returns (len << 8) | symbol, or -1.
// The code book of the lengths 12..16 sits in LDS, kLongInts ints per table slot: first_code, count, first_sym per length, then
// the 256 symbol values (round 4: it used to be read from global memory, three dependent loads per length, by every lane of
// the wave that met such a code — rare per lane, but with 64 lanes some lane meets one in a good part of the turns).
constexpr int kLongInts = 16 + 64;
__device__ __forceinline__ int long_code(const int32_t *book, uint32_t p16) {
    int l = 0, at = 0;
#pragma unroll
    for (int i = 4; i >= 0; --i) {                                  // (the shortest length that matches wins: walked downwards)
        const int len = kLBits + 1 + i;
        const int d = (int)(p16 >> (16 - len)) - book[i];
        const bool ok = d >= 0 && d < book[5 + i];
        l = ok ? len : l;
        at = ok ? book[10 + i] + d : at;
    }
    if (l == 0) return -1;
    return (l << 8) | reinterpret_cast<const uint8_t *>(book + 16)[at & 255];
}

// Branch-free: when the buffer is at most half full the next dword goes in.  The load of the dword after it is
// issued right away and not needed before the next call.
__device__ __forceinline__ void refill(LaneBits &s, const unsigned char *streamb) {
    const bool want = s.bc <= 32;       // idle lanes top up as well: harmless, their stream is theirs
    const uint32_t t = want ? s.nxtw : 0u;
    s.bb |= (uint64_t)t << ((32 - s.bc) & 63);
    const uint32_t inc = want ? 4u : 0u;
    s.voff += inc;
    s.bc += (int)(inc * 8u);
    if (want) s.nxtw = *reinterpret_cast<const uint32_t *>(streamb + s.voff);     // every lane reads its own cache line: only who needs it
}

__device__ __forceinline__ int opaque(int x) {      // the value, with its provenance hidden from the optimiser (no instruction)
    asm("" : "+v"(x));
    return x;
}

__device__ __forceinline__ uint32_t bfm0(uint32_t n) {      // (1 << n) - 1 in one instruction
    uint32_t r;
    asm("v_bfm_b32 %0, %1, 0" : "=v"(r) : "v"(n));
    return r;
}

__device__ __forceinline__ int extend(uint32_t raw, int n) {   // bin_twos_complement (:1636-1646); n = 0 -> 0
    const int half = (1 << n) >> 1;
    return (int)raw - (((int)raw < half) ? ((1 << n) - 1) : 0);
}

}  // namespace

// WGT: the batch has more tables than LDS holds (files with their own optimised tables): every workgroup gets the up
// to 8 or 16 (kMaxWgTables) tables its segments' images use (`wg_tabs`, built by api.hip for exactly this launch shape); a
// lane's table numbers are then positions in that list.  Without WGT the LDS slot of a table is its index in the batch.
template <bool WGT>
__global__ __launch_bounds__(256) void k_huffman_lanes(const uint32_t *__restrict__ stream,   // stage 0's output
                                                       const int32_t *__restrict__ seg_bits,  // bits per segment
                                                       const DevSegment *__restrict__ segs, int64_t n_segs,
                                                       const DevImage *__restrict__ images,
                                                       const DevHuff *__restrict__ huff,
                                                       const uint16_t *__restrict__ lut11,   // [n_huff][kLSize]
                                                       int n_huff, int16_t *__restrict__ coef,
                                                       int32_t *__restrict__ status, int lpw, int tr,
                                                       const DevVSeg *__restrict__ vsegs /* or null */,
                                                       const int32_t *__restrict__ wg_tabs /* WGT: [gridDim.x][kMaxWgTables], -1 = unused */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef MJ_DIAGNOSTIC
    const int dbg = tr >> 8;          // MJ_DEBUG_STAGE1 of the diagnostic build: 1 = no coefficient stores, 2 = no stream refill loads
    tr &= 1;
#endif
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);                          // [n_huff][kLSize]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block buffers: lpw lanes per wave are active (rounded up to even: the flush moves blocks in pairs)
    const int lpw2 = (lpw + 1) & ~1;
    const int wstride = (lpw2 * kBlkStride + 3) & ~3;                             // dwords per wave, 16-byte multiple
    uint32_t *s_blk = reinterpret_cast<uint32_t *>(smem + (size_t)n_huff * kLSize * 2) + wave * wstride;
    // output byte offset of every lane's segment, for the flush (lane (o, dw) needs block o's, not its own)
    uint64_t *s_base = reinterpret_cast<uint64_t *>(smem + (size_t)n_huff * kLSize * 2 + (size_t)4 * wstride * 4) + wave * lpw2;
    // what a lane that sits a symbol out reads instead of its LUT: length 0, run 64, size 0 (bit 15 keeps it apart from
    // the all-zero "longer than 11 bits" entries; real lengths are 1..11 and use bits 11..14)
    uint16_t *s_null = reinterpret_cast<uint16_t *>(smem + (size_t)n_huff * kLSize * 2 + (size_t)4 * wstride * 4 + (size_t)4 * lpw2 * 8);
    if (tid == 0) *s_null = (uint16_t)(0x8000u | (64u << 4));
    int32_t *s_long = reinterpret_cast<int32_t *>(s_null + 8);                      // [n_huff][kLongInts], see long_code

    const int32_t *my_tabs = WGT ? wg_tabs + (size_t)blockIdx.x * kMaxWgTables : nullptr;
    for (int i = tid; i < n_huff * kLongInts; i += 256) {
        const int j = i / kLongInts, q = i - j * kLongInts;
        const int t = WGT ? my_tabs[j] : j;
        int32_t v = 0;
        if (t >= 0) {
            const DevHuff *h = huff + t;
            if (q < 5) v = h->first_code[kLBits + 1 + q];
            else if (q < 10) v = h->count[kLBits + 1 + q - 5];
            else if (q < 15) v = h->first_sym[kLBits + 1 + q - 10];
            else if (q >= 16) v = reinterpret_cast<const int32_t *>(h->vals)[q - 16];
        }
        s_long[i] = v;
    }
    if (WGT) {
        for (int j = 0; j < n_huff; ++j) {                  // n_huff = slots in LDS here
            const int t = my_tabs[j];
            if (t < 0) continue;
            for (int i = tid; i < kLSize / 8; i += 256)
                reinterpret_cast<uint4 *>(s_lut + j * kLSize)[i] = reinterpret_cast<const uint4 *>(lut11 + (size_t)t * kLSize)[i];
        }
    } else {
        for (int i = tid; i < n_huff * kLSize / 8; i += 256)
            reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(lut11)[i];
    }
    for (int i = tid; i < 4 * wstride; i += 256)
        (reinterpret_cast<uint32_t *>(smem + (size_t)n_huff * kLSize * 2))[i] = 0;
    __syncthreads();

    const int64_t seg_id = ((int64_t)blockIdx.x * 4 + wave) * lpw + lane;   // lpw segments per wave (tunable)
    const bool have = lane < lpw && seg_id < n_segs;
    // The unit of work is a restart segment, or — for long segments cut up by the synchronisation passes
    // (huffman_sync.hip) — a virtual segment: a run of whole MCUs inside one, with its start bit and DC predictors.
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
    // per-lane table numbers per component (LDS slot = index into the batch's table array)
    int dcT[3], acT[3];
    {
        int seen = 0;
        for (int b = 0; b < 8 && b < im->blocks_per_mcu; ++b) {
            const int c = im->blk_comp[b];
            if (!((seen >> c) & 1)) {
                seen |= 1 << c;
                const int dt = im->tab_index[im->blk_dc_slot[b]], at = im->tab_index[im->blk_ac_slot[b]];
                if (c == 0) { dcT[0] = dt; acT[0] = at; } else if (c == 1) { dcT[1] = dt; acT[1] = at; } else { dcT[2] = dt; acT[2] = at; }
            }
        }
        if (!(seen & 2)) { dcT[1] = dcT[0]; acT[1] = acT[0]; }
        if (!(seen & 4)) { dcT[2] = dcT[0]; acT[2] = acT[0]; }
    }
    // the tables' indices in the batch (for the canonical search of long codes) and their LDS slots (for everything else)
    int dcG[3] = {dcT[0], dcT[1], dcT[2]}, acG[3] = {acT[0], acT[1], acT[2]};
    if (WGT) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int d = 0, a = 0;
            for (int j = 0; j < kMaxWgTables; ++j) {
                const int t = my_tabs[j];
                d = t == dcG[c] ? j : d;
                a = t == acG[c] ? j : a;
            }
            dcT[c] = d; acT[c] = a;
        }
    }
    const int n_mcu = have ? sg.n_mcu : 0;
    int max_mcu = n_mcu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_mcu = max(max_mcu, __shfl_xor(max_mcu, o));
    max_mcu = __builtin_amdgcn_readfirstlane(max_mcu);

    // ---- per-lane bit reader (see LaneBits); lanes without a segment read dword 0 and never decode
    const unsigned char *streamb = reinterpret_cast<const unsigned char *>(stream);
    LaneBits br;
    const uint32_t bit_sh = vsegs ? (uint32_t)vs.bit0 & 31u : 0u;           // a virtual segment starts at any bit
    const uint32_t voff0 = !have ? 0u : (vsegs ? vs.voff0 + ((uint32_t)vs.bit0 >> 5) * 4u : (uint32_t)(((sg.begin >> 2) + seg_id) * 4));
    const int nbits = !have ? 0 : (vsegs ? vs.bit_end - vs.bit0 : seg_bits[seg_id]);
    {
        const uint32_t d0 = *reinterpret_cast<const uint32_t *>(streamb + voff0), d1 = *reinterpret_cast<const uint32_t *>(streamb + voff0 + 4);
        br.bb = (((uint64_t)d0 << 32) | d1) << bit_sh;
        br.bc = 64 - (int)bit_sh;
        br.voff = voff0 + 8;
        br.nxtw = *reinterpret_cast<const uint32_t *>(streamb + br.voff);
    }
    // bits consumed since the (virtual) segment's first bit: everything fetched minus what is still buffered
    auto consumed = [&]() { return (int)((br.voff - voff0) * 8u) - br.bc - (int)bit_sh; };
    // byte offset of this lane's first output block
    const int64_t out_off = (im->block_off + (int64_t)sg.mcu0 * bpm) * 128;
    if (lane < lpw2) s_base[lane] = (uint64_t)out_off;
    int pred0 = vs.pred[0], pred1 = vs.pred[1], pred2 = vs.pred[2];        // zero for a restart segment (:900)
    int err = 0;
    // store positions 2dw, 2dw+1 of a block; tr: the plan keeps blocks transposed ([u][v]) for the row-major stage 2
    const int na = 2 * (lane & 31), nb = na + 1;
    const int zz_a = c_zz_of_nat_l[tr ? ((na & 7) << 3 | na >> 3) : na], zz_b = c_zz_of_nat_l[tr ? ((nb & 7) << 3 | nb >> 3) : nb];
    uint32_t *myblk = s_blk + (lane < lpw2 ? lane : 0) * kBlkStride;
    int16_t *myblk16 = reinterpret_cast<int16_t *>(myblk);

    const uint64_t full_mask = lpw >= 64 ? ~0ull : (1ull << lpw) - 1;
#ifdef MJ_DIAGNOSTIC
    uint64_t dacc[4] = {0, 0, 0, 0}, dlast = __builtin_amdgcn_s_memtime();
#endif
    for (int m = 0; m < max_mcu; ++m) {
        const bool in_mcu = m < n_mcu;
        for (int b = 0; b < bpm; ++b) {
            const int comp = (int)((comp_pk_u >> (8 * b)) & 0xFF);          // wave-uniform
            const int dct = comp == 0 ? dcT[0] : (comp == 1 ? dcT[1] : dcT[2]);
            const int act_ = comp == 0 ? acT[0] : (comp == 1 ? acT[1] : acT[2]);
            const bool act = in_mcu && err == 0;

            // ---- DC (:810-820): one symbol per lane, straight-line
            refill(br, streamb);
            int k;
            {
                const uint32_t p16 = (uint32_t)(br.bb >> 48);
                const int e = s_lut[dct * kLSize + (p16 >> (16 - kLBits))];
                int len = e >> 8, s = e & 0xFF;
                if (__builtin_amdgcn_ballot_w64(act && len == 0) != 0) {                              // code longer than 11 bits: rare
                    if (act && len == 0) {
                        const int r = long_code(s_long + dct * kLongInts, p16);
                        len = r < 0 ? 0 : r >> 8; s = r < 0 ? 255 : r & 0xFF;
                    }
                }
                const bool bad = act && s > 16;
                err = bad ? MJ_ST_BAD_CODE : err;
                const bool ok = act && !bad;
                const int ln = ok ? len : 0, sz = ok ? s : 0;
                const uint32_t hw = (uint32_t)(br.bb >> 32) << ln;             // ln + sz <= 32 <= bc
                const uint32_t rawv = (hw >> 1) >> (31 - sz);
                br.bb <<= ln + sz;
                br.bc -= ln + sz;
                const int pred = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
                const int dcv = (int)(int16_t)(extend(rawv, sz) + pred);
                pred0 = (ok && comp == 0) ? dcv : pred0;
                pred1 = (ok && comp == 1) ? dcv : pred1;
                pred2 = (ok && comp == 2) ? dcv : pred2;
                myblk16[ok ? 0 : 64] = (int16_t)dcv;                           // 64 = the row's pad slot, never flushed
                k = ok ? 1 : 64;
            }
            // ---- AC (:833-866): one symbol per lane and iteration until every lane is at its end of block.
            // LUT entry = len << 11 | run << 4 | size; the end-of-block symbol carries run = 64, so "kk >= 64" covers
            // both :849 and :855-856 (the value bits stay unread in both cases).
            const uint16_t *alut = s_lut + act_ * kLSize;
            // One symbol of this lane: look up, EXTEND, store, consume — straight-line select code, written for the fewest
            // instructions: what bounds this kernel is the number of instructions on each wave's serial path (a wave issues
            // one every ~5 cycles at best, tools/issue_rate_probe.hip, and all segments are in flight at once), not their
            // cost on the SIMD.  Lanes that sit a symbol out (`on` false: block finished, or — second symbol of an
            // iteration — waiting for more bits) read the null entry instead of their LUT — length 0, run 64, size 0 —
            // and fall through the same code without consuming or storing anything.
            // k: index of the next coefficient; >= 64 = this lane's block is finished (64 or 65).
            auto symbol = [&](bool on, bool keep_k) {
                const uint32_t hi = (uint32_t)(br.bb >> 32);
                const uint16_t *ep = alut + (hi >> (32 - kLBits));
                const int e = *(on ? ep : s_null);
                int ln = (e >> 11) & 15, run = (e >> 4) & 127, size = e & 15;
                if (e < 2048) {                                               // code longer than 11 bits: rare (the branch
                    const int r = long_code(s_long + act_ * kLongInts, hi >> 16);   // is skipped when no lane has one)
                    err = r < 0 ? MJ_ST_BAD_CODE : err;
                    const int hv = r & 0xFF;
                    ln = r < 0 ? 0 : r >> 8;
                    run = (r < 0 || hv == 0) ? 64 : hv >> 4;
                    size = r < 0 ? 0 : hv & 15;
                }
                const int kk = k + run;
                const int n = kk < 64 ? size : 0;                              // not end of block (:849), not past it (:855-856): else the value bits stay unread
                const int tot = ln + n;                                        // <= 31 <= bc
                // EXTEND (bin_twos_complement, :1636-1646) of the n bits behind the code: a leading 1 is the value itself, a
                // leading 0 is value - (2^n - 1); the leading bit is 0 exactly when 2*raw <= 2^n - 1.
                const uint32_t raw = __builtin_amdgcn_ubfe(hi, (uint32_t)(32 - tot), (uint32_t)n);
                const uint32_t ones = bfm0((uint32_t)n);
                const int slot = min(kk, 64);                                  // zig-zag order (the flush permutes); 64 = the row's pad slot
                myblk16[slot] = (int16_t)(raw - ((raw << 1) <= ones ? ones : 0u));
                br.bb <<= tot;
                br.bc -= tot;
                const int knew = slot + 1;                                      // kk + 1, or 65 = finished
                k = keep_k ? (on ? knew : k) : knew;
            };
            // Two symbols per iteration: the refill, the loop test and the register shuffling at the loop head are paid
            // once.  After a refill the buffer holds >= 33 bits; the second symbol goes ahead when >= 31 are left
            // (16 code bits + 15 value bits is the longest symbol), else it simply waits for the next iteration.
#ifdef MJ_DIAGNOSTIC
#define LSTAMP(i) do { if (dbg == 3) { uint64_t s_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s_) :: "memory"); dacc[i] += s_ - dlast; dlast = s_; } } while (0)
            LSTAMP(3);        // DC symbol, flush, clear, loop set-up
#else
#define LSTAMP(i) do { } while (0)
#endif
            while (__builtin_amdgcn_ballot_w64(k < 64) != 0) {
                refill(br, streamb);
                LSTAMP(0);    // loop test + refill (the wait for the stream word is here)
                symbol(k < 64, false);
                LSTAMP(1);    // first symbol
                symbol(k < 64 && br.bc >= 31, true);
                LSTAMP(2);    // second symbol
            }
            // a segment that consumed more bits than it has is corrupt (it has been reading its neighbour's bytes)
            err = (act && err == 0 && consumed() > nbits) ? MJ_ST_OVERRUN : err;
            // ---- round of 64 blocks done: LDS -> HBM, two full lines per instruction, and clear
            // lane (o, dw) moves the two coefficients of natural positions 2dw, 2dw+1 of block o: they are read from
            // their zig-zag slots, so the block lands in HBM in the natural [v][u] order stage 2 wants
#ifdef MJ_DIAGNOSTIC
            const uint64_t act_mask = dbg == 1 ? 0ull : __ballot(in_mcu);
#else
            const uint64_t act_mask = __ballot(in_mcu);
#endif
            const uint32_t blk_byte = (uint32_t)(m * bpm + b) * 128u;    // same for every lane (same layout)
            const int half = lane >> 5, dw = lane & 31;
            unsigned char *dst0 = reinterpret_cast<unsigned char *>(coef) + blk_byte + dw * 4;
            if (act_mask == full_mask) {
                // every lane of the wave has a block: no per-pair tests, offsets fold into the instructions
                const int16_t *hb = reinterpret_cast<const int16_t *>(s_blk + half * kBlkStride);
                const uint64_t *bp = s_base + half;
#pragma unroll 4
                for (int s2 = 0; s2 < (lpw >> 1); ++s2) {
                    const int16_t *ob16 = hb + s2 * (4 * kBlkStride);
                    const uint32_t v = (uint32_t)(uint16_t)ob16[zz_a] | ((uint32_t)(uint16_t)ob16[zz_b] << 16);
                    *reinterpret_cast<uint32_t *>(dst0 + bp[2 * s2]) = v;
                }
                if ((lpw & 1) && half == 0) {
                    const int16_t *ob16 = hb + (lpw >> 1) * (4 * kBlkStride);
                    const uint32_t v = (uint32_t)(uint16_t)ob16[zz_a] | ((uint32_t)(uint16_t)ob16[zz_b] << 16);
                    *reinterpret_cast<uint32_t *>(dst0 + bp[lpw - 1]) = v;
                }
            } else {
#pragma unroll 2
                for (int s2 = 0; s2 < 32; ++s2) {
                    if (((act_mask >> (2 * s2)) & 3) == 0) continue;         // uniform
                    const int o = 2 * s2 + half;
                    const int16_t *ob16 = reinterpret_cast<const int16_t *>(s_blk + o * kBlkStride);
                    const uint32_t v = (uint32_t)(uint16_t)ob16[zz_a] | ((uint32_t)(uint16_t)ob16[zz_b] << 16);
                    if ((act_mask >> o) & 1) *reinterpret_cast<uint32_t *>(dst0 + s_base[o]) = v;
                }
            }
            // clear the wave's block rows for the next round, 16 bytes per lane and instruction
            for (int i = lane; i < wstride / 4; i += 64) reinterpret_cast<uint4 *>(s_blk)[i] = make_uint4(0, 0, 0, 0);
        }
    }

#ifdef MJ_DIAGNOSTIC
    if (dbg == 3 && lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_dbg_lanes[i], (unsigned long long)dacc[i]);
    }
#endif
    if (have) {
        // bits left over: a whole unread byte before the next restart marker means the count-driven reference and the
        // marker-driven segmentation disagree (:898-900)
        const int left = nbits - consumed();
        if (!err && vsegs && vs.last == 0 && left != 0) err = MJ_ST_DESYNC;     // a virtual segment ends exactly where the next starts
        if (!err && (vsegs ? vs.last == 2 : !sg.last) && left >= 8) err = MJ_ST_DESYNC;
        if (err) atomicMax(status + sg.image, err);
    }
}

// Lanes per wavefront for `n_segs` units of work with `n_slots` tables in LDS (api.hip asks too: a batch with more
// tables than LDS holds gets per-workgroup table lists, which depend on how the launch groups the segments).
int lanes_per_wave(int64_t n_segs, int n_slots) {
    if (const char *e = opt("MJ_LANES_PER_WAVE")) { const int lpw = atoi(e); if (lpw >= 1 && lpw <= 64) return lpw < 2 ? 2 : lpw; }   // (this form's flush wants two lanes at least)
    // measured on MI355X: the kernel is instruction-issue bound, every instruction costing the same
    // whatever the number of active lanes, and latency bound below ~2 waves per SIMD.  All workgroups are resident
    // at once and run equally long, so what matters besides ~3-4 waves per SIMD is that every CU gets the SAME
    // number of workgroups: lanes per wave = segments / (4 workgroups x 4 waves x CUs), rounded up.
    const int cus = device_cus();
    // ... and when there are more segments than one such round holds, as many equal rounds as needed: lanes per wave
    // capped where four workgroups still fit a CU's LDS, then spread evenly over the rounds
    auto lds_of = [&](int l) { const int l2 = (l + 1) & ~1; return (size_t)n_slots * kLSize * 2 + (size_t)4 * ((l2 * kBlkStride + 3) & ~3) * 4 + (size_t)4 * l2 * 8 + 16 + (size_t)n_slots * 320; };
    int fit = 64;
    while (fit > 8 && 4 * lds_of(fit) > 160 * 1024) --fit;
    const int64_t per = (int64_t)16 * cus;                   // waves of one round: 4 workgroups x 4 waves x CUs
    const int64_t rounds = (n_segs + per * fit - 1) / (per * fit);
    const int64_t want = (n_segs + per * rounds - 1) / (per * rounds);
    if (want < 8) {
        // not enough segments for four workgroups of 8-lane waves per CU: as many workgroups per CU as 8 lanes per wave
        // allow, and again the same number on every CU (26 112 segments: 9 lanes = 3 per CU, 7.4 ms; 8 lanes = 3.2 per
        // CU, i.e. four on some, 8.3 ms)
        int64_t m = n_segs / ((int64_t)4 * cus * 8);
        m = m < 1 ? 1 : (m > 4 ? 4 : m);
        const int64_t l = (n_segs + 4 * m * cus - 1) / (4 * m * cus);
        return (int)(l < 8 ? 8 : (l > 64 ? 64 : l));
    }
    return (int)(want > 64 ? 64 : want);
}

hipError_t launch_huffman_lanes(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs, int64_t n_segs,
                                const DevImage *images, const DevHuff *huff, const uint16_t *lut11, int n_huff,
                                int16_t *coef, int32_t *status, int transposed, const DevVSeg *vsegs, const int32_t *wg_tabs, int wg_slots) {
    if (n_segs == 0) return hipSuccess;
    const int n_slots = wg_tabs ? wg_slots : n_huff;
    const int lpw_run = lanes_per_wave(n_segs, n_slots);
#ifdef MJ_DIAGNOSTIC
    if (const char *e = getenv("MJ_DEBUG_STAGE1")) transposed |= atoi(e) << 8;
#endif
    const int64_t blocks = (n_segs + 4 * lpw_run - 1) / (4 * lpw_run);
    const int lpw2_run = (lpw_run + 1) & ~1, wstride_run = (lpw2_run * kBlkStride + 3) & ~3;
    const size_t lds = (size_t)n_slots * kLSize * 2 + (size_t)4 * wstride_run * 4 + (size_t)4 * lpw2_run * 8 + 16 + (size_t)n_slots * kLongInts * 4;
    static OncePerDevice attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_huffman_lanes<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_huffman_lanes<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    });
    if (wg_tabs)
        hipLaunchKernelGGL(k_huffman_lanes<true>, dim3((unsigned)blocks), dim3(256), lds, stream, dstream, seg_bits, segs, n_segs, images,
                           huff, lut11, n_slots, coef, status, lpw_run, transposed, vsegs, wg_tabs);
    else
        hipLaunchKernelGGL(k_huffman_lanes<false>, dim3((unsigned)blocks), dim3(256), lds, stream, dstream, seg_bits, segs, n_segs, images,
                           huff, lut11, n_slots, coef, status, lpw_run, transposed, vsegs, wg_tabs);
    return hipGetLastError();
}

}  // namespace mj
