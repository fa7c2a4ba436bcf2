// Stage 1, sub-segment synchronisation passes: parallelism inside long restart segments (files without DRI).
//
// A restart segment is cut into chunks of `cbits` bits of its stage-0 stream, one per lane.  A count-only decode of a
// chunk from an entry state (bit position, block within the MCU, coefficient index) yields its exit state — the first
// symbol that starts at or behind the chunk's end — and, on the way, the number of blocks it completed, the first MCU
// boundary inside the chunk and the DC differences summed up to there and over the whole chunk.
// Round 0: a lane cannot know its chunk's entry state, so it starts half a chunk EARLIER from the guess "a block
// starts here" and decodes towards its chunk; Huffman streams re-synchronise (and a wrong MCU phase desynchronises
// again quickly, the luma and chroma tables being different), so the state in which it crosses into the chunk is
// almost always the true one.  Round 1..: every chunk whose assumed entry state is not its predecessor's exit state
// is redone from that state, until a round changes no exit state (the host reads one counter per round).
// k_build_vsegs then turns the records into "virtual segments" (runs of whole MCUs with their start bit and DC
// predictors) which the lane-parallel kernel decodes like restart segments.
//
// Tables: the 11-bit LUTs in the unified format  len << 11 | run << 4 | size  (DC tables: run 0, size = the symbol;
// AC tables: end of block = run 64); a lane looks up the DC table of its block's component when its coefficient
// index is 0 and the AC table otherwise, so DC and AC symbols take the same straight-line step.
#include <stdlib.h>

#include <type_traits>

#include "mijpeg_internal.h"

namespace mj {

namespace {
constexpr int kLBits = 11;
constexpr int kLSize = 1 << kLBits;

struct LaneBits {
    uint64_t bb;
    uint32_t voff, nxtw;
    int bc;
};

__device__ __forceinline__ void refill(LaneBits &s, const unsigned char *streamb) {
    const bool want = s.bc <= 32;
    const uint32_t t = want ? s.nxtw : 0u;
    s.bb |= (uint64_t)t << ((32 - s.bc) & 63);
    const uint32_t inc = want ? 4u : 0u;
    s.voff += inc;
    s.bc += (int)(inc * 8u);
    if (want) s.nxtw = *reinterpret_cast<const uint32_t *>(streamb + s.voff);     // 64 lanes, 64 cache lines: only who needs it
}

// Codes of 12..16 bits (0.4 % of the symbols — but with 128 symbols per wave and turn, 40 % of the turns meet one): the
// canonical search over the five lengths with the code book in LDS.  Per table slot kLongInts ints: first_code, count,
// first_sym for the lengths 12..16, then the 256 symbol values.  (Round 4: the book used to be read from global memory,
// three dependent loads per length, by every lane of the wave that had such a code.)
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
}  // namespace

// state word: bit position within the segment's stream (32) | block within the MCU (8) | coefficient index (8)
__device__ __forceinline__ uint64_t pack_state(uint32_t pos, int b, int k) { return (uint64_t)pos | ((uint64_t)(uint32_t)b << 32) | ((uint64_t)(uint32_t)k << 40); }

// WARM = true: the first round.  A chunk cannot know its entry state, so its lane starts `warm` bits EARLIER, assuming
// a block starts there, decodes towards its own first bit — by then it has almost always re-synchronised — takes the
// state in which it crosses into the chunk as its entry state and records the chunk from there.
// WARM = false: a repair round — every chunk starts from its predecessor's exit state; chunks whose record was computed
// from exactly that state (nearly all, after the first round) are skipped.
template <bool WARM>
__global__ __launch_bounds__(256) void k_sync_count(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                    const DevSegment *__restrict__ segs, const DevImage *__restrict__ images,
                                                    const DevHuff *__restrict__ huff, const uint16_t *__restrict__ lut11u,
                                                    int n_huff, const DevChunk *__restrict__ chunks, int64_t n_chunks, int cbits, int warm,
                                                    const uint64_t *__restrict__ entry, uint64_t *__restrict__ exit_out,
                                                    DevChunkOut *__restrict__ outs, int32_t *__restrict__ changed,
                                                    const int32_t *__restrict__ wg_tabs /* or null: [gridDim.x][kMaxWgTables] */,
                                                    const int32_t *__restrict__ prev_changed /* or null: the round before this one's count */) {
    // A repair round that changed no chunk's exit state leaves both state buffers equal and every record valid: the rounds queued
    // behind it have nothing to do (they are queued blind, a fixed number: no host round trip) and leave at once.
    if (!WARM && prev_changed && *prev_changed == 0) return;
    if constexpr (!WARM) {
        // ... and a workgroup none of whose chunks has a new entry state (nearly all of them, from the first repair round on)
        // passes its exit states on before it has loaded a table
        const int64_t c0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
        bool keep = true;
        if (c0 < n_chunks) {
            const uint64_t e0 = chunks[c0].j == 0 ? pack_state(0, 0, 0) : entry[c0 - 1];
            keep = outs[c0].entry == e0 && outs[c0].blocks >= 0;
        }
        if (__syncthreads_and(keep)) {
            if (c0 < n_chunks) exit_out[c0] = entry[c0];
            return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint16_t *s_null = s_lut + (size_t)n_huff * kLSize;                 // n_huff = table slots in LDS
    int32_t *s_glob = reinterpret_cast<int32_t *>(s_null + 8);           // slot -> index of the table in the batch
    int32_t *s_long = s_glob + kMaxWgTables;                            // per slot: the code book of the lengths 12..16 (long_code)
    const int tid = threadIdx.x;
    // a batch with more tables than LDS holds: this workgroup's chunks use the (at most n_huff = 8 or 16) tables listed in
    // wg_tabs, and "slot" below is a position in that list
    const int32_t *my_tabs = wg_tabs ? wg_tabs + (size_t)blockIdx.x * kMaxWgTables : nullptr;
    if (my_tabs) {
        for (int j = 0; j < n_huff; ++j) {
            const int t = my_tabs[j];
            if (t < 0) continue;
            for (int i = tid; i < kLSize / 8; i += 256)
                reinterpret_cast<uint4 *>(s_lut + j * kLSize)[i] = reinterpret_cast<const uint4 *>(lut11u + (size_t)t * kLSize)[i];
        }
    } else {
        for (int i = tid; i < n_huff * kLSize / 8; i += 256)
            reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(lut11u)[i];
    }
    if (tid < kMaxWgTables) s_glob[tid] = my_tabs ? my_tabs[tid] : tid;
    if (tid == 0) *s_null = (uint16_t)(0x8000u | (64u << 4));          // length 0, run 64, size 0: a lane that is done
    for (int i = tid; i < n_huff * kLongInts; i += 256) {
        const int j = i / kLongInts, q = i - j * kLongInts;
        const int t = my_tabs ? my_tabs[j] : j;
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
    __syncthreads();

    const int64_t c = (int64_t)blockIdx.x * 256 + tid;
    const bool have = c < n_chunks;
    const DevChunk ch = chunks[have ? c : 0];
    const DevSegment sg = segs[ch.seg];
    const DevImage *im = images + sg.image;
    const int bpm = __builtin_amdgcn_readfirstlane(im->blocks_per_mcu);
    // component of every block of the MCU, 4 bits each, and this image's table slots: dc of comp 0,1,2 then ac of 0,1,2
    uint32_t comp_pk = 0, slots_pk = 0;
    for (int b = 0; b < 8 && b < im->blocks_per_mcu; ++b) {
        const int cc = im->blk_comp[b];
        comp_pk |= (uint32_t)cc << (4 * b);
        int dslot = im->tab_index[im->blk_dc_slot[b]], aslot = im->tab_index[im->blk_ac_slot[b]];
        if (my_tabs) {
            int d = 0, a = 0;
            for (int j = 0; j < kMaxWgTables; ++j) { d = my_tabs[j] == dslot ? j : d; a = my_tabs[j] == aslot ? j : a; }
            dslot = d; aslot = a;
        }
        slots_pk |= (uint32_t)dslot << (4 * cc);
        slots_pk |= (uint32_t)aslot << (4 * (cc + 3));
    }
    const int nbits = have ? seg_bits[ch.seg] : 0;
    const int64_t lim64 = (int64_t)(ch.j + 1) * cbits;
    uint32_t limit = have ? (uint32_t)(lim64 < nbits ? lim64 : nbits) : 0u;
    uint32_t pos;
    int b, k;
    uint64_t my_entry;
    const uint32_t own_start = (uint32_t)((int64_t)ch.j * cbits);
    if (ch.j == 0) my_entry = pack_state(0, 0, 0);
    else if (!WARM) my_entry = entry[c - 1];
    else my_entry = pack_state(own_start > (uint32_t)warm ? own_start - (uint32_t)warm : 0u, 0, 0);     // the guess
    pos = (uint32_t)my_entry; b = (int)((my_entry >> 32) & 0xFF); k = (int)((my_entry >> 40) & 0xFF);
    bool skip = false;
    if constexpr (!WARM) {
        // a record computed from this very entry state is still right: keep it, keep its exit state
        skip = have && outs[c].entry == my_entry && outs[c].blocks >= 0;
        if (skip) limit = 0;
    }
    const uint32_t final_limit = limit;
    if (WARM) limit = ch.j > 0 ? (own_start < final_limit ? own_start : final_limit) : 0u;   // first: up to the chunk's own first bit

    const unsigned char *streamb = reinterpret_cast<const unsigned char *>(stream);
    const uint32_t voff0 = (uint32_t)(((sg.begin >> 2) + ch.seg) * 4);
    LaneBits br;
    {
        const uint32_t w0 = voff0 + (pos >> 5) * 4;
        const uint32_t d0 = *reinterpret_cast<const uint32_t *>(streamb + w0), d1 = *reinterpret_cast<const uint32_t *>(streamb + w0 + 4);
        br.bb = (((uint64_t)d0 << 32) | d1) << (pos & 31);
        br.bc = 64 - (int)(pos & 31);
        br.voff = w0 + 8;
        br.nxtw = *reinterpret_cast<const uint32_t *>(streamb + br.voff);
    }
    int blocks = 0;
    int dc0 = 0, dc1 = 0, dc2 = 0;                        // DC differences summed per component
    int bnd_pos = -1, bnd_blocks = 0, bd0 = 0, bd1 = 0, bd2 = 0;
    if (!WARM && b == 0 && k == 0 && pos < limit) bnd_pos = (int)pos;                     // entered at an MCU boundary
    const unsigned char *lutb = reinterpret_cast<const unsigned char *>(s_lut);

    auto symbol = [&](bool allow, auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;      // false during the run-up: nothing is recorded there
        const bool on = allow && pos < limit;
        const int comp = (int)((comp_pk >> (4 * b)) & 15u);
        const bool isdc = k == 0;
        const int slot = (int)((slots_pk >> (4 * (comp + (isdc ? 0 : 3)))) & 15u);
        const uint32_t hi = (uint32_t)(br.bb >> 32);
        const unsigned char *ep = lutb + (((uint32_t)slot << (kLBits + 1)) | ((hi >> (31 - kLBits)) & ((kLSize - 1) << 1)));
        const int e = *reinterpret_cast<const uint16_t *>(on ? ep : reinterpret_cast<const unsigned char *>(s_null));
        int ln = (e >> 11) & 15, run = (e >> 4) & 127, size = e & 15;
        if (e < 2048) {                                               // longer than 11 bits: rare
            const int r = long_code(s_long + slot * kLongInts, hi >> 16);
            const int hv = r & 0xFF;
            ln = r < 0 ? 1 : r >> 8;                                  // no code at all: skip a bit (the data is garbage anyway)
            run = r < 0 ? 0 : (isdc ? 0 : (hv == 0 ? 64 : hv >> 4));
            size = r < 0 ? 0 : hv & 15;
        }
        const int kk = k + run;
        const bool val = kk < 64;
        const int n = val ? size : 0;
        if constexpr (ACC) {                                           // the DC difference (EXTEND, :818-820)
            const uint32_t hw = hi << ln;
            const uint32_t lead = (uint32_t)((int32_t)hw >> 31);
            const uint32_t raw = __builtin_amdgcn_ubfe(hw, (uint32_t)(32 - n) & 31u, (uint32_t)n);
            const int d = (on && isdc) ? (int)(raw - (((1u << n) - 1u) & ~lead)) : 0;
            dc0 += comp == 0 ? d : 0;
            dc1 += comp == 1 ? d : 0;
            dc2 += comp == 2 ? d : 0;
        }
        const int tot = ln + n;
        br.bb <<= tot;
        br.bc -= tot;
        pos += (uint32_t)tot;
        const bool be = on && (!val || kk == 63);                     // this symbol ended its block
        k = on ? (be ? 0 : kk + 1) : k;
        const int b1 = b + 1;
        const bool wrap = be && b1 == bpm;                            // ... and its MCU
        b = be ? (wrap ? 0 : b1) : b;
        blocks += be ? 1 : 0;
        if constexpr (ACC) {
            const bool take = wrap && bnd_pos < 0 && pos < limit;
            bnd_pos = take ? (int)pos : bnd_pos;
            bnd_blocks = take ? blocks : bnd_blocks;
            bd0 = take ? dc0 : bd0; bd1 = take ? dc1 : bd1; bd2 = take ? dc2 : bd2;
        }
    };
    if constexpr (WARM) {
        while (__builtin_amdgcn_ballot_w64(pos < limit) != 0) {           // run-up: reach the chunk's own first bit
            refill(br, streamb);
            symbol(true, std::false_type{});
            symbol(br.bc >= 31, std::false_type{});
        }
        // the chunk starts here, as far as this lane can tell (the first chunk of a segment knows)
        my_entry = pack_state(pos, b, k);
        blocks = 0;
        limit = final_limit;
        if (b == 0 && k == 0 && pos < limit) bnd_pos = (int)pos;
    }
    while (__builtin_amdgcn_ballot_w64(pos < limit) != 0) {
        refill(br, streamb);
        symbol(true, std::true_type{});
        symbol(br.bc >= 31, std::true_type{});
    }
    if (have) {
        if constexpr (!WARM) {
            const uint64_t ex = skip ? entry[c] : pack_state(pos, b, k);      // entry[] holds last round's exit states
            if (!skip) {
                DevChunkOut o;
                o.entry = my_entry; o.blocks = blocks; o.bnd_pos = bnd_pos; o.bnd_blocks = bnd_blocks;
                o.dc_bnd[0] = (int16_t)bd0; o.dc_bnd[1] = (int16_t)bd1; o.dc_bnd[2] = (int16_t)bd2;
                o.dc_sum[0] = (int16_t)dc0; o.dc_sum[1] = (int16_t)dc1; o.dc_sum[2] = (int16_t)dc2;
                outs[c] = o;
            }
            if (ex != entry[c]) atomicAdd(changed, 1);
            exit_out[c] = ex;
        } else {
            DevChunkOut o;
            o.entry = my_entry; o.blocks = blocks; o.bnd_pos = bnd_pos; o.bnd_blocks = bnd_blocks;
            o.dc_bnd[0] = (int16_t)bd0; o.dc_bnd[1] = (int16_t)bd1; o.dc_bnd[2] = (int16_t)bd2;
            o.dc_sum[0] = (int16_t)dc0; o.dc_sum[1] = (int16_t)dc1; o.dc_sum[2] = (int16_t)dc2;
            outs[c] = o;
            exit_out[c] = pack_state(pos, b, k);
        }
    }
}

// ---- counting on resolved tables, repairs from a work list (round 5) ---------------------------------------------------------
// Same records, same states as k_sync_count — for batches of at most eight tables, one role each, MCUs of at most eight blocks
// (every batch of everyday files).  What is different:
//   * the symbol step.  A table entry (plan_create.hip: build_count_tables; all tables with the same index width, 12 bits where
//     LDS allows) is 32 bits:  bits consumed, code AND value (0..5) | how far the coefficient index moves: run + 1, 128 = end of
//     block (8..15) | for DC tables the EXTENDed difference (16..30, jpeg_decoder.py:818-820 evaluated when the table is built).
//     Counting does not need AC values, so every AC code that fits the index is a finished entry whatever its value bits are;
//     the step has no field arithmetic, no EXTEND and no search for long codes.  Bit 31 marks what is left — codes longer than
//     the index (second-level tables) and DC symbols whose value bits reach beyond it — for one arithmetic step under a
//     wave-uniform branch (taken by ~10 % of the steps, the old long-code path by 40 %).
//   * a symbol that lands behind its block (:855-856: its value bits stay unread) can only be a damaged stream; the step does not
//     look for it, it keeps the maximum index reached, and a record computed across one gets an entry state no chunk can have:
//     k_build_vsegs then marks the image MJ_ST_UNCONVERGED and the caller decodes it without synchronisation rounds, exactly.
//   * the DC sums of the three components are three 16-bit fields (their int16 wrap is what the records store anyway) moved by
//     packed adds; where the first MCU boundary lay is a minimum, the snapshot behind a wave-uniform branch.
//   * repairs.  The old rounds ran every workgroup that held one chunk with a new entry state (257 of 652 workgroups for ~1 000
//     of 166 900 chunks, round after round).  Now k_sync_scan lists the chunks whose recorded entry state is not their
//     predecessor's exit state, with that state, and k_count<true> walks each listed chunk again on one lane — four wavefronts
//     per workgroup, one per SIMD: a lone wavefront runs ~1.5 x faster than one of three on a SIMD.  A walk from the true entry
//     state nearly always leaves the chunk in the state the old one left it in (the old walk had found its way inside the
//     chunk); where not, the lane walks on into the next chunk (at most max_links times), whose record — the old one, or the
//     one its own lane is writing from the stale state at this moment, a chunk's walk before this lane gets there — is void.
//     One launch instead of a fixed number of rounds; chunks behind the end of the stream (the chunk list is cut from the
//     file's byte count, byte stuffing included) take no part.  Whether everything settled is still decided by
//     k_build_vsegs: a record that is not its predecessor's continuation marks the image.
struct SyncItem { int32_t c, pad; uint64_t entry; };
struct CountArgs {
    const uint32_t *stream; const int32_t *seg_bits; const DevSegment *segs; const DevImage *images;
    const uint32_t *lutc; int32_t tab_bytes, n_tabs, wbits;
    const DevChunk *chunks; int64_t n_chunks; int32_t cbits, warm;
    uint64_t *exit_state; DevChunkOut *outs;
    const SyncItem *items; const int32_t *n_items; int32_t max_links;
    int32_t *owner;            // repair launch: per chunk, the first chunk of the leftmost walk that has taken it
};

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

template <bool REPAIR>
__global__ __launch_bounds__(1024) void k_count(CountArgs A) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    int32_t n_items = 0;
    if constexpr (REPAIR) {
        n_items = *A.n_items;
        if ((int64_t)blockIdx.x * nthreads >= n_items) return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = tid; i < A.n_tabs * A.tab_bytes / 16; i += nthreads) reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(A.lutc)[i];
    __syncthreads();
    typedef const uint32_t __attribute__((address_space(3))) *lds_cu32;
    const uint32_t tab0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)smem;
    const uint32_t S = (uint32_t)A.tab_bytes, W = (uint32_t)A.wbits;

    int64_t c;
    bool have;
    uint64_t my_entry = 0;
    if constexpr (REPAIR) {
        const int64_t i = (int64_t)blockIdx.x * nthreads + tid;
        have = i < n_items;
        const SyncItem it = A.items[have ? i : 0];
        c = it.c; my_entry = it.entry;
    } else {
        c = (int64_t)blockIdx.x * nthreads + tid;
        have = c < A.n_chunks;
    }
    DevChunk ch = A.chunks[have ? c : 0];
    const DevSegment sg = A.segs[ch.seg];
    const DevImage *im = A.images + sg.image;
    const uint32_t bpm8 = 8u * (uint32_t)im->blocks_per_mcu;
    // per block of the MCU one byte: DC table (0..2) | AC table (3..5) | component (6..7)
    uint64_t pk = 0;
    for (int b = 0; b < 8 && b < im->blocks_per_mcu; ++b) {
        const uint32_t dt = (uint32_t)im->tab_index[im->blk_dc_slot[b]] & 7u, at = (uint32_t)im->tab_index[im->blk_ac_slot[b]] & 7u;
        pk |= (uint64_t)(dt | (at << 3) | ((uint32_t)im->blk_comp[b] << 6)) << (8 * b);
    }
    const uint32_t nbits = have ? (uint32_t)A.seg_bits[ch.seg] : 0u;
    const uint32_t cbits = (uint32_t)A.cbits;
    auto chunk_limit = [&](int j) { const uint64_t l = (uint64_t)(j + 1) * cbits; return (uint32_t)(l < nbits ? l : nbits); };
    const uint32_t own_start = (uint32_t)ch.j * cbits;
    if constexpr (!REPAIR) {
        if (ch.j == 0) my_entry = pack_state(0, 0, 0);
        else my_entry = pack_state(own_start > (uint32_t)A.warm ? own_start - (uint32_t)A.warm : 0u, 0, 0);     // the guess
    }
    uint32_t pos = (uint32_t)my_entry, b8 = 8u * (uint32_t)((my_entry >> 32) & 0xFF), k = (uint32_t)((my_entry >> 40) & 0xFF);
    uint32_t final_limit = have ? chunk_limit(ch.j) : 0u;

    // the bit reader: bb = the next bits (bit 63 first), valid up to bit position `top` of the segment's stream; nxtw = the dword behind
    const unsigned char *streamb = reinterpret_cast<const unsigned char *>(A.stream);
    uint64_t bb;
    uint32_t voff, nxtw, top;
    {
        const uint32_t w0 = (uint32_t)(((sg.begin >> 2) + ch.seg) * 4) + (pos >> 5) * 4;
        const uint32_t d0 = *reinterpret_cast<const uint32_t *>(streamb + w0), d1 = *reinterpret_cast<const uint32_t *>(streamb + w0 + 4);
        bb = (((uint64_t)d0 << 32) | d1) << (pos & 31);
        top = (pos & ~31u) + 64u;
        voff = w0 + 8;
        nxtw = *reinterpret_cast<const uint32_t *>(streamb + voff);
    }
    uint32_t blocks = 0, mx = 0, bnd_pos = ~0u, bnd_blocks = 0, acc01 = 0, acc2 = 0, bd01 = 0, bd2 = 0;

    auto refill = [&]() -> uint32_t {          // returns the position up to which a SECOND symbol of this turn may start
        const uint32_t bc = top - pos;
        const bool want = bc <= 32u;
        bb |= (uint64_t)(want ? nxtw : 0u) << ((32u - bc) & 63u);
        top += want ? 32u : 0u;
        voff += want ? 4u : 0u;
        if (want) nxtw = *reinterpret_cast<const uint32_t *>(streamb + voff);     // 64 lanes, 64 cache lines: only who needs it
        return top - 30u;                      // (a symbol takes at most 16 + 15 bits)
    };
    auto step = [&](uint32_t lim, auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;      // false during the run-up: nothing is recorded there
        const uint32_t d = (uint32_t)(pk >> b8);
        const bool isdc = k == 0;
        const uint32_t tbase = tab0 + __umul24(__builtin_amdgcn_ubfe(d, isdc ? 0u : 3u, 3u), S);
        const uint32_t hi = (uint32_t)(bb >> 32);
        uint32_t e = *(lds_cu32)(uintptr_t)(tbase + ((hi >> (32u - W)) << 2));
        e = pos < lim ? e : 0u;                             // a lane that is done: nothing moves
        int val = __builtin_amdgcn_sbfe((int)e, 16u, 15u);
        const uint32_t open = e >> 31;
        if (__builtin_amdgcn_ballot_w64(open != 0u) != 0) {
            if (open != 0u) {
                uint32_t o = e;
                if (o & 0x40000000u) o = *(lds_cu32)(uintptr_t)(tbase + (o & 0xFFFFu) + (__builtin_amdgcn_ubfe(hi, 16u, 16u - W) << 2));
                const uint32_t len = o & 31u, size = (o >> 16) & 15u;
                const uint32_t hw = hi << len;
                const uint32_t lead = (uint32_t)((int32_t)hw >> 31);
                const uint32_t raw = __builtin_amdgcn_ubfe(hw, (32u - size) & 31u, size);
                const int v = (int)(raw - (((1u << size) - 1u) & ~lead));
                val = (isdc && len) ? v : 0;
                e = len ? ((len + size) | (o & 0xFF00u)) : 0x101u;      // no code at all: skip a bit (the data is garbage anyway)
            }
        }
        bb <<= (e & 63u);
        pos += e & 63u;
        k += (e >> 8) & 0xFFu;
        mx = max(mx, k ^ 128u);
        const bool be = k >= 64u;                           // this symbol ended its block
        k = be ? 0u : k;
        const uint32_t b8n = b8 + 8u;
        const bool wrap = be && b8n == bpm8;                // ... and its MCU
        b8 = be ? (wrap ? 0u : b8n) : b8;
        blocks += be ? 1u : 0u;
        if constexpr (ACC) {
            const uint64_t t = (uint64_t)((uint32_t)val & 0xFFFFu) << ((d >> 2) & 0x30u);
            acc01 = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, acc01) + __builtin_bit_cast(u16x2, (uint32_t)t)));
            acc2 += (uint32_t)(t >> 32);
            if (__builtin_amdgcn_ballot_w64(wrap) != 0) {
                const uint32_t cand = wrap ? pos : ~0u;
                const bool take = cand < bnd_pos;
                bnd_pos = take ? cand : bnd_pos;
                bnd_blocks = take ? blocks : bnd_blocks;
                bd01 = take ? acc01 : bd01;
                bd2 = take ? acc2 : bd2;
            }
        }
    };
    if constexpr (!REPAIR) {
        // the run-up: from the guess to the chunk's own first bit — by then the walk has almost always re-synchronised
        const uint32_t lim = ch.j > 0 ? (own_start < final_limit ? own_start : final_limit) : 0u;
        if (__builtin_amdgcn_ballot_w64(pos < lim) != 0) {
            do {
                const uint32_t t2 = refill();
                step(lim, std::false_type{});
                step(lim < t2 ? lim : t2, std::false_type{});
            } while (__builtin_amdgcn_ballot_w64(pos < lim) != 0);
        }
        my_entry = pack_state(pos, (int)(b8 >> 3), (int)k);
    }
    int links = 0;
    const int32_t origin = (int32_t)c;                  // (repair) where this lane's walk started
    for (;;) {
        if constexpr (REPAIR) {
            // A chunk may be walked by several lanes of this launch — its own, from the state its predecessor USED to leave, and
            // lanes that walk on into it from further left because their chunk now leaves differently.  Truth travels from left
            // to right: the walk that started furthest left wins, whenever the others finish.  A lane takes a chunk before it
            // walks it (atomic minimum of the walks' first chunks) and writes its record only if nobody from further left has
            // taken it meanwhile.  (Without this the order of two such writes was left to the launch's timing — fine alone on the
            // chip, not under other streams' kernels: a sharded queue's plans, two in flight per rank, showed it as one image in
            // a few hundred executes coming back MJ_ST_UNCONVERGED.)
            if (have) __hip_atomic_fetch_min(A.owner + c, origin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        blocks = 0; mx = 0; bnd_blocks = 0; acc01 = 0; acc2 = 0; bd01 = 0; bd2 = 0;
        bnd_pos = (b8 == 0 && k == 0 && pos < final_limit) ? pos : ~0u;      // entered at an MCU boundary
        if (__builtin_amdgcn_ballot_w64(pos < final_limit) != 0) {
            do {
                const uint32_t t2 = refill();
                step(final_limit, std::true_type{});
                step(final_limit < t2 ? final_limit : t2, std::true_type{});
            } while (__builtin_amdgcn_ballot_w64(pos < final_limit) != 0);
        }
        const uint64_t ex = pack_state(pos, (int)(b8 >> 3), (int)k);
        uint64_t old_exit = ex;
        bool mine = true;
        DevChunkOut o{};
        if (have) {
            o.entry = mx > 192u ? ~0ull : my_entry;
            o.blocks = (int32_t)blocks;
            o.bnd_pos = bnd_pos < final_limit ? (int32_t)bnd_pos : -1;
            o.bnd_blocks = (int32_t)bnd_blocks;
            o.dc_bnd[0] = (int16_t)bd01; o.dc_bnd[1] = (int16_t)(bd01 >> 16); o.dc_bnd[2] = (int16_t)bd2;
            o.dc_sum[0] = (int16_t)acc01; o.dc_sum[1] = (int16_t)(acc01 >> 16); o.dc_sum[2] = (int16_t)acc2;
            if constexpr (!REPAIR) {
                A.outs[c] = o;
                A.exit_state[c] = ex;
            }
        }
        if constexpr (REPAIR) {
            // Writes of different lanes to one chunk have to land in memory in the order they are made: device-scope loads and
            // write-through stores (a plain store stays in its XCD's L2 until the launch ends, and the write-backs come in any order).
            // And "am I still the leftmost taker?" + the record's four words + the exit state are ONE step per chunk: a lane holds
            // the chunk's lock (A.owner[n_chunks + c]) while it looks and writes.  Without it a lane that had looked before a walk
            // from further left took the chunk, and was then held up (wavefronts are pre-empted when queues are oversubscribed),
            // could land its words behind the winner's — in part: a record with one walk's entry state and the other's counts
            // passes k_build_vsegs' entry == exit test.  (A lane never waits with the lock taken, and the loop is the SIMT form:
            // whoever gets the lock does its writing and lets go in the same turn, lanes of one wavefront included.)
            bool pending = have;
            while (__builtin_amdgcn_ballot_w64(pending) != 0) {
                if (pending) {
                    int32_t *lock = A.owner + A.n_chunks + c;
                    int32_t expect = 0;
                    if (__hip_atomic_compare_exchange_strong(lock, &expect, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        mine = __hip_atomic_load(A.owner + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= origin;
                        if (mine) {
                            old_exit = __hip_atomic_load(A.exit_state + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint64_t *w = reinterpret_cast<const uint64_t *>(&o);
                            uint64_t *dst = reinterpret_cast<uint64_t *>(A.outs + c);
#pragma unroll
                            for (int q = 0; q < 4; ++q) __hip_atomic_store(dst + q, w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(A.exit_state + c, ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __hip_atomic_store(lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        pending = false;
                    }
                }
            }
        }
        if constexpr (!REPAIR) break;
        // The chunk leaves in the state it left in before (nearly always: the old walk had found its way inside the chunk): the next
        // chunk's record was computed from that state, or is being computed from it by its own lane right now.  Else this lane
        // walks on: whatever the next chunk's record says, or its lane writes — a chunk's walk earlier than this one — is void.
        bool go = have && mine && ex != old_exit && links < A.max_links && c + 1 < A.n_chunks;
        if (go) go = A.chunks[c + 1].seg == ch.seg && (uint64_t)(ch.j + 1) * cbits < nbits;
        have = go;
        if (go) { ++c; ++ch.j; ++links; my_entry = ex; final_limit = chunk_limit(ch.j); }
        else final_limit = 0;
        if (__builtin_amdgcn_ballot_w64(go) == 0) break;
    }
}

// The chunks whose record was not computed from their predecessor's exit state: the work list of k_count<true>, each with the
// state to start from.
__global__ void k_sync_scan(const DevChunk *__restrict__ chunks, int64_t n_chunks, const DevChunkOut *__restrict__ outs,
                            const uint64_t *__restrict__ exit_state, const int32_t *__restrict__ seg_bits, int cbits,
                            SyncItem *__restrict__ items, int32_t *__restrict__ n_items, int32_t *__restrict__ owner) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool add = c < n_chunks;
    if (add) { owner[c] = 0x7FFFFFFF; owner[n_chunks + c] = 0; }      // nobody has taken the chunk yet, nobody holds its lock (k_count<true>)
    uint64_t e = 0;
    if (add) {
        const DevChunk ch = chunks[c];
        add = ch.j > 0 && (int64_t)ch.j * cbits < seg_bits[ch.seg];               // (behind the end of the stream: nothing there)
        if (add) {
            e = exit_state[c - 1];
            add = outs[c].entry != e;
        }
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(add);                          // one atomic per wavefront
    if (m == 0) return;
    const int lane = threadIdx.x & 63, first = __builtin_ctzll(m);
    int base = 0;
    if (lane == first) base = atomicAdd(n_items, (int32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readlane(base, first);
    if (add) {
        SyncItem it;
        it.c = (int32_t)c; it.pad = 0; it.entry = e;
        items[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = it;
    }
}

hipError_t launch_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs, const DevImage *images,
                        const uint32_t *lutc, int tab_bytes, int n_tabs, int wbits, const DevChunk *chunks, int64_t n_chunks, int cbits,
                        int warm_bits, uint64_t *exit_state, DevChunkOut *outs, void *items, int32_t *n_items, int max_links, int32_t *owner) {
    if (n_chunks == 0) return hipSuccess;
    CountArgs A{};
    A.owner = owner;
    A.stream = dstream; A.seg_bits = seg_bits; A.segs = segs; A.images = images;
    A.lutc = lutc; A.tab_bytes = tab_bytes; A.n_tabs = n_tabs; A.wbits = wbits;
    // the run-up in front of every chunk: half a chunk.  A wrong guess costs one lane of the repair launch, whose duration is one
    // lone wavefront's walk of one chunk however many lanes there are (up to a wave per SIMD) — unless some old walk had not found
    // its way by the END of its chunk either: then that lane walks a second chunk.  256 files of 1080p, 1 KiB chunks, first walk +
    // repair in us: run-up 256 B 647 + 622 (12 % of the guesses wrong, some second links), 512 B 790 + 330, 1 KiB 929 + 296.
    A.chunks = chunks; A.n_chunks = n_chunks; A.cbits = cbits; A.warm = warm_bits >= 0 ? warm_bits : cbits / 2;
    A.exit_state = exit_state; A.outs = outs; A.items = reinterpret_cast<const SyncItem *>(items); A.n_items = n_items; A.max_links = max_links;
    const size_t lds = (size_t)n_tabs * (size_t)tab_bytes;
    static OncePerDevice attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_count<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    // the first walk: every chunk, one workgroup per CU and as few waves of workgroups as that allows, all equally full
    const int64_t cus = device_cus();
    const int64_t waves_of_wgs = (n_chunks + cus * 1024 - 1) / (cus * 1024);
    int threads = (int)(((n_chunks + cus * waves_of_wgs - 1) / (cus * waves_of_wgs) + 63) / 64 * 64);
    threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
    hipLaunchKernelGGL(k_count<false>, dim3((unsigned)((n_chunks + threads - 1) / threads)), dim3(threads), lds, stream, A);
    if (max_links > 0) {
        if (hipError_t e = launch_fill_words(stream, n_items, 0u, 1); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_sync_scan, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, stream, chunks, n_chunks, outs, exit_state, seg_bits,
                           cbits, reinterpret_cast<SyncItem *>(items), n_items, owner);
        // (four wavefronts per workgroup = one per SIMD: each still runs alone, and the tables are in LDS four times sooner)
        hipLaunchKernelGGL(k_count<true>, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), lds, stream, A);
    }
    return hipGetLastError();
}

// One virtual segment per chunk that holds an MCU boundary: from that boundary to the next chunk's (or to the end of
// the restart segment).  Block offsets and DC predictors are the sums over the segment's earlier chunks: one workgroup per
// restart segment runs a prefix sum over its chunks' records, 256 at a time (round 5; every chunk used to add up all its
// predecessors by itself — 123 us per 256 files of 650 chunks each, and quadratic in the length of a file).
__global__ __launch_bounds__(256) void k_build_vsegs(const DevChunk *__restrict__ chunks, const int32_t *__restrict__ seg_chunk0,
                                                     const DevChunkOut *__restrict__ outs, const DevSegment *__restrict__ segs,
                                                     const int32_t *__restrict__ seg_bits, const DevImage *__restrict__ images,
                                                     DevVSeg *__restrict__ vsegs, const uint64_t *__restrict__ final_exit, int cbits,
                                                     int32_t *__restrict__ status) {
    __shared__ int64_t s_blocks[256];
    __shared__ int32_t s_dc[3][256];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int64_t c0 = seg_chunk0[seg], n = seg_chunk0[seg + 1] - c0;
    const DevSegment sg = segs[seg];
    const int32_t nbits = seg_bits[seg];
    const int bpm = images[sg.image].blocks_per_mcu;
    int64_t carry_blocks = 0;
    int32_t carry_dc[3] = {0, 0, 0};
    for (int64_t base = 0; base < n; base += 256) {
        const int64_t c = c0 + base + tid;
        const bool have = base + tid < n;
        DevChunkOut me{};
        if (have) me = outs[c];
        // The repairs are bounded, queued without looking.  They have settled iff every chunk's record was computed from the
        // state its predecessor finally left; a chunk for which that is not so (long chains of wrong guesses: a pathological
        // stream; a symbol that lands behind its block: a damaged one) marks its image, and the host decodes that image again
        // without synchronisation rounds.  Chunks behind the end of the stream hold nothing.
        const int64_t j = base + tid;
        if (have && j > 0 && j * cbits < nbits && me.entry != final_exit[c - 1]) atomicMax(status + sg.image, MJ_ST_UNCONVERGED);
        if (have && j == 0 && me.entry != 0) atomicMax(status + sg.image, MJ_ST_UNCONVERGED);      // (a damaged first chunk)
        s_blocks[tid] = have ? me.blocks : 0;
        for (int q = 0; q < 3; ++q) s_dc[q][tid] = have ? me.dc_sum[q] : 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                         // inclusive prefix sums
            int64_t vb = 0;
            int32_t v0 = 0, v1 = 0, v2 = 0;
            if (tid >= d) { vb = s_blocks[tid - d]; v0 = s_dc[0][tid - d]; v1 = s_dc[1][tid - d]; v2 = s_dc[2][tid - d]; }
            __syncthreads();
            s_blocks[tid] += vb; s_dc[0][tid] += v0; s_dc[1][tid] += v1; s_dc[2][tid] += v2;
            __syncthreads();
        }
        if (have) {
            DevVSeg v{};
            v.image = sg.image;
            v.voff0 = (uint32_t)(((sg.begin >> 2) + seg) * 4);
            if (me.bnd_pos >= 0) {
                const int64_t P = carry_blocks + s_blocks[tid] - me.blocks;            // blocks before this chunk
                const int d0 = carry_dc[0] + s_dc[0][tid] - me.dc_sum[0], d1 = carry_dc[1] + s_dc[1][tid] - me.dc_sum[1],
                          d2 = carry_dc[2] + s_dc[2][tid] - me.dc_sum[2];
                const int64_t start_block = P + me.bnd_blocks;
                // the next boundary, or the end of the restart segment
                int64_t Pn = P + me.blocks, end_block = (int64_t)sg.n_mcu * bpm;
                int32_t bit_end = nbits;
                int last = 1;
                for (int64_t q = c + 1; q < c0 + n; ++q) {
                    const DevChunkOut o = outs[q];
                    if (o.bnd_pos >= 0) { end_block = Pn + o.bnd_blocks; bit_end = o.bnd_pos; last = 0; break; }
                    Pn += o.blocks;
                }
                v.bit0 = me.bnd_pos;
                v.bit_end = bit_end;
                v.mcu0 = sg.mcu0 + (int32_t)(start_block / bpm);
                v.n_mcu = (int32_t)((end_block - start_block) / bpm);
                if (v.n_mcu < 0 || start_block % bpm != 0) v.n_mcu = 0;        // cannot happen for a consistent stream; decode nothing then
                // records that have not settled (the image is marked above and decoded again) may add up to anything: whatever they
                // say, a virtual segment stays inside its restart segment's MCUs — its lane must not write into the next image
                const int32_t seg_end_mcu = sg.mcu0 + sg.n_mcu;
                if (v.mcu0 < sg.mcu0 || v.mcu0 >= seg_end_mcu) v.n_mcu = 0;
                else if (v.n_mcu > seg_end_mcu - v.mcu0) v.n_mcu = seg_end_mcu - v.mcu0;
                v.pred[0] = (int16_t)(d0 + me.dc_bnd[0]);
                v.pred[1] = (int16_t)(d1 + me.dc_bnd[1]);
                v.pred[2] = (int16_t)(d2 + me.dc_bnd[2]);
                v.last = (int16_t)(last && sg.last ? 1 : (last ? 2 : 0));      // 2 = ends its restart segment, another follows
            }
            vsegs[c] = v;
        }
        carry_blocks += s_blocks[255];
        for (int q = 0; q < 3; ++q) carry_dc[q] += s_dc[q][255];
        __syncthreads();
    }
}

hipError_t launch_sync_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                             const DevImage *images, const DevHuff *huff, const uint16_t *lut11u, int n_huff,
                             const DevChunk *chunks, int64_t n_chunks, int cbits, const uint64_t *entry, uint64_t *exit_out,
                             DevChunkOut *outs, int32_t *changed, const int32_t *wg_tabs, int wg_slots, const int32_t *prev_changed, int warm_bits) {
    if (n_chunks == 0) return hipSuccess;
    if (wg_tabs) n_huff = wg_slots;                             // table slots in LDS
    const size_t lds = (size_t)n_huff * kLSize * 2 + 16 + kMaxWgTables * 4 + (size_t)n_huff * kLongInts * 4;
    const dim3 grid((unsigned)((n_chunks + 255) / 256));
    static OncePerDevice attr_once;
    attr_once.run([&] {                          // 16 table slots: just over the 64 KiB default
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync_count<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    });
    const int warm = warm_bits >= 0 ? warm_bits : cbits / 2;   // run-up in front of every chunk (swept: half a chunk is best; MJ_SYNC_WARM is read once, when the plan is created)
    if (entry)
        hipLaunchKernelGGL(k_sync_count<false>, grid, dim3(256), lds, stream, dstream, seg_bits, segs, images, huff, lut11u, n_huff,
                           chunks, n_chunks, cbits, warm, entry, exit_out, outs, changed, wg_tabs, prev_changed);
    else
        hipLaunchKernelGGL(k_sync_count<true>, grid, dim3(256), lds, stream, dstream, seg_bits, segs, images, huff, lut11u, n_huff,
                           chunks, n_chunks, cbits, warm, entry, exit_out, outs, changed, wg_tabs, nullptr);
    return hipGetLastError();
}

hipError_t launch_build_vsegs(hipStream_t stream, const DevChunk *chunks, const int32_t *seg_chunk0, int64_t n_segs, const DevChunkOut *outs,
                              const DevSegment *segs, const int32_t *seg_bits, const DevImage *images, DevVSeg *vsegs,
                              const uint64_t *final_exit, int cbits, int32_t *status) {
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_vsegs, dim3((unsigned)n_segs), dim3(256), 0, stream, chunks, seg_chunk0, outs, segs, seg_bits, images, vsegs,
                       final_exit, cbits, status);
    return hipGetLastError();
}

}  // namespace mj
