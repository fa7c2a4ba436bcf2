// Stage 1, sub-segment synchronisation passes (EXPERIMENTAL — reached only through MJ_SYNC_PROBE, see api.hip).
//
// A restart segment is cut into chunks of `cbits` bits of its stage-0 stream, one per lane.  A count-only decode of a
// chunk from a given entry state (bit position, block within the MCU, coefficient index) yields its exit state — the
// first symbol that starts at or behind the chunk's end — and the number of blocks it completed.  Run once from the
// guess "a block starts at the chunk's first bit" and then from the predecessor's exit state until nothing changes,
// this finds the true state at every chunk boundary (Huffman streams re-synchronise), i.e. places inside a segment
// where an independent decoder can start: parallelism for files without restart markers.
//
// Tables: the 11-bit LUTs in the unified format  len << 11 | run << 4 | size  (DC tables: run 0, size = the symbol;
// AC tables: end of block = run 64); a lane looks up the DC table of its block's component when its coefficient
// index is 0 and the AC table otherwise, so DC and AC symbols take the same straight-line step.
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

__device__ __forceinline__ int long_code(const DevHuff *t, uint32_t p16) {
    int r = -1;
#pragma unroll 1
    for (int l = kLBits + 1; l <= 16; ++l) {
        const int d = (int)(p16 >> (16 - l)) - t->first_code[l];
        if (r < 0 && d >= 0 && d < t->count[l]) r = (l << 8) | t->vals[t->first_sym[l] + d];
    }
    return r;
}
}  // namespace

// state word: bit position within the segment's stream (32) | block within the MCU (8) | coefficient index (8)
__device__ __forceinline__ uint64_t pack_state(uint32_t pos, int b, int k) { return (uint64_t)pos | ((uint64_t)(uint32_t)b << 32) | ((uint64_t)(uint32_t)k << 40); }

__global__ __launch_bounds__(256) void k_sync_count(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                    const DevSegment *__restrict__ segs, const DevImage *__restrict__ images,
                                                    const DevHuff *__restrict__ huff, const uint16_t *__restrict__ lut11u,
                                                    int n_huff, const DevChunk *__restrict__ chunks, int64_t n_chunks, int cbits,
                                                    const uint64_t *__restrict__ entry /* null: speculative */,
                                                    uint64_t *__restrict__ exit_out, int32_t *__restrict__ blocks_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);
    uint16_t *s_null = s_lut + (size_t)n_huff * kLSize;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_huff * kLSize / 8; i += 256)
        reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(lut11u)[i];
    if (tid == 0) *s_null = (uint16_t)(0x8000u | (64u << 4));          // length 0, run 64, size 0: a lane that is done
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
        slots_pk |= (uint32_t)im->tab_index[im->blk_dc_slot[b]] << (4 * cc);
        slots_pk |= (uint32_t)im->tab_index[im->blk_ac_slot[b]] << (4 * (cc + 3));
    }
    const int nbits = have ? seg_bits[ch.seg] : 0;
    const int64_t lim64 = (int64_t)(ch.j + 1) * cbits;
    const uint32_t limit = have ? (uint32_t)(lim64 < nbits ? lim64 : nbits) : 0u;
    uint32_t pos;
    int b, k;
    if (ch.j == 0) { pos = 0; b = 0; k = 0; }
    else if (entry) {
        const uint64_t e = entry[c - 1];
        pos = (uint32_t)e; b = (int)((e >> 32) & 0xFF); k = (int)((e >> 40) & 0xFF);
    } else { pos = (uint32_t)((int64_t)ch.j * cbits); b = 0; k = 0; }

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
    const unsigned char *lutb = reinterpret_cast<const unsigned char *>(s_lut);

    auto symbol = [&](bool allow) {
        const bool on = allow && pos < limit;
        const int comp = (int)((comp_pk >> (4 * b)) & 15u);
        const int slot = (int)((slots_pk >> (4 * (comp + (k == 0 ? 0 : 3)))) & 15u);
        const uint32_t hi = (uint32_t)(br.bb >> 32);
        const unsigned char *ep = lutb + (((uint32_t)slot << (kLBits + 1)) | ((hi >> (31 - kLBits)) & ((kLSize - 1) << 1)));
        const int e = *reinterpret_cast<const uint16_t *>(on ? ep : reinterpret_cast<const unsigned char *>(s_null));
        int ln = (e >> 11) & 15, run = (e >> 4) & 127, size = e & 15;
        if (e < 2048) {                                               // longer than 11 bits: rare
            const int r = long_code(huff + slot, hi >> 16);
            const int hv = r & 0xFF;
            ln = r < 0 ? 1 : r >> 8;                                  // no code: skip a bit (the chunk is garbage anyway)
            run = r < 0 ? 0 : ((k != 0 && hv == 0) ? 64 : (k != 0 ? hv >> 4 : 0));
            size = r < 0 ? 0 : hv & 15;
        }
        const int kk = k + run;
        const bool val = kk < 64;
        const int n = val ? size : 0;
        const int tot = ln + n;
        br.bb <<= tot;
        br.bc -= tot;
        pos += (uint32_t)tot;
        const bool be = on && (!val || kk == 63);                     // this symbol ended its block
        k = on ? (be ? 0 : kk + 1) : k;
        const int b1 = b + 1;
        b = be ? (b1 == bpm ? 0 : b1) : b;
        blocks += be ? 1 : 0;
    };
    while (__builtin_amdgcn_ballot_w64(pos < limit) != 0) {
        refill(br, streamb);
        symbol(true);
        symbol(br.bc >= 31);
    }
    if (have) {
        exit_out[c] = pack_state(pos, b, k);
        blocks_out[c] = blocks;
    }
}

hipError_t launch_sync_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                             const DevImage *images, const DevHuff *huff, const uint16_t *lut11u, int n_huff,
                             const DevChunk *chunks, int64_t n_chunks, int cbits, const uint64_t *entry, uint64_t *exit_out,
                             int32_t *blocks_out) {
    if (n_chunks == 0) return hipSuccess;
    const size_t lds = (size_t)n_huff * kLSize * 2 + 16;
    hipLaunchKernelGGL(k_sync_count, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), lds, stream, dstream, seg_bits, segs,
                       images, huff, lut11u, n_huff, chunks, n_chunks, cbits, entry, exit_out, blocks_out);
    return hipGetLastError();
}

}  // namespace mj
