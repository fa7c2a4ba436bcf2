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

// One virtual segment per chunk that holds an MCU boundary: from that boundary to the next chunk's (or to the end of
// the restart segment).  Block offsets and DC predictors come from the sums over the segment's earlier chunks.
__global__ void k_build_vsegs(const DevChunk *__restrict__ chunks, int64_t n_chunks, const DevChunkOut *__restrict__ outs,
                              const DevSegment *__restrict__ segs, const int32_t *__restrict__ seg_bits,
                              const DevImage *__restrict__ images, DevVSeg *__restrict__ vsegs,
                              const uint64_t *__restrict__ final_exit, int32_t *__restrict__ status) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const DevChunk ch = chunks[c];
    const DevSegment sg = segs[ch.seg];
    // The repair rounds are a fixed number, queued without looking.  They have settled iff every chunk's record was computed
    // from the state its predecessor finally left; a chunk for which that is not so (long chains of wrong guesses: a
    // pathological stream) marks its image, and the host decodes that image again without synchronisation rounds.
    if (ch.j > 0 && outs[c].entry != final_exit[c - 1]) atomicMax(status + sg.image, MJ_ST_UNCONVERGED);
    const int bpm = images[sg.image].blocks_per_mcu;
    DevVSeg v{};
    v.image = sg.image;
    v.voff0 = (uint32_t)(((sg.begin >> 2) + ch.seg) * 4);
    const DevChunkOut me = outs[c];
    if (me.bnd_pos >= 0) {
        int64_t P = 0;                      // blocks before this chunk
        int d0 = 0, d1 = 0, d2 = 0;
        for (int64_t q = c - ch.j; q < c; ++q) {
            const DevChunkOut o = outs[q];
            P += o.blocks; d0 += o.dc_sum[0]; d1 += o.dc_sum[1]; d2 += o.dc_sum[2];
        }
        const int64_t start_block = P + me.bnd_blocks;
        // the next boundary, or the end of the restart segment
        int64_t Pn = P + me.blocks, end_block = (int64_t)sg.n_mcu * bpm;
        int32_t bit_end = seg_bits[ch.seg];
        int last = 1;
        for (int64_t q = c + 1; q < n_chunks && chunks[q].seg == ch.seg; ++q) {
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

hipError_t launch_sync_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                             const DevImage *images, const DevHuff *huff, const uint16_t *lut11u, int n_huff,
                             const DevChunk *chunks, int64_t n_chunks, int cbits, const uint64_t *entry, uint64_t *exit_out,
                             DevChunkOut *outs, int32_t *changed, const int32_t *wg_tabs, int wg_slots, const int32_t *prev_changed, int warm_bits) {
    if (n_chunks == 0) return hipSuccess;
    if (wg_tabs) n_huff = wg_slots;                             // table slots in LDS
    const size_t lds = (size_t)n_huff * kLSize * 2 + 16 + kMaxWgTables * 4 + (size_t)n_huff * kLongInts * 4;
    const dim3 grid((unsigned)((n_chunks + 255) / 256));
    static bool attr_set[kMaxDevices] = {false};
    if (!attr_set[current_device()]) {                          // 16 table slots: just over the 64 KiB default
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync_count<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set[current_device()] = true;
    }
    const int warm = warm_bits >= 0 ? warm_bits : cbits / 2;   // run-up in front of every chunk (swept: half a chunk is best; MJ_SYNC_WARM is read once, when the plan is created)
    if (entry)
        hipLaunchKernelGGL(k_sync_count<false>, grid, dim3(256), lds, stream, dstream, seg_bits, segs, images, huff, lut11u, n_huff,
                           chunks, n_chunks, cbits, warm, entry, exit_out, outs, changed, wg_tabs, prev_changed);
    else
        hipLaunchKernelGGL(k_sync_count<true>, grid, dim3(256), lds, stream, dstream, seg_bits, segs, images, huff, lut11u, n_huff,
                           chunks, n_chunks, cbits, warm, entry, exit_out, outs, changed, wg_tabs, nullptr);
    return hipGetLastError();
}

hipError_t launch_build_vsegs(hipStream_t stream, const DevChunk *chunks, int64_t n_chunks, const DevChunkOut *outs,
                              const DevSegment *segs, const int32_t *seg_bits, const DevImage *images, DevVSeg *vsegs,
                              const uint64_t *final_exit, int32_t *status) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_vsegs, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, stream, chunks, n_chunks, outs, segs,
                       seg_bits, images, vsegs, final_exit, status);
    return hipGetLastError();
}

}  // namespace mj
