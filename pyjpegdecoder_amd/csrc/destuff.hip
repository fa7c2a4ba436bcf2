// Stage 0 (gfx950): the bit reader's byte rules, applied once per restart segment by a whole wavefront.
//
// The reference's bits_generator (jpeg_decoder.py:654-695) hands out the bytes of the entropy-coded segment
// MSB first and, after ANY byte 0xFF, skips the next byte unconditionally (:676-677).  Doing that inside the
// serial Huffman loop costs every decoded symbol a dozen instructions of "is there an 0xFF in the next dword"
// bookkeeping, so it is done here instead, data-parallel: one wavefront per restart segment reads the segment
// 256 bytes at a time, drops the bytes the reference would skip and writes the survivors as BIG-ENDIAN dwords
// (first stream byte in bits 31..24), zero-padded to a dword, at dword  (begin >> 2) + segment index  of the
// stream buffer (regions of different segments cannot overlap, see api.hip).  The lane-parallel stage-1 kernel
// then refills its bit buffer with plain aligned dword loads.
//
// "Dropped" is a sequential rule (a dropped byte that is itself 0xFF does not drop its successor), solved per
// chunk as a two-state transfer function per lane and a scalar fix-up over the few lanes whose output state
// depends on their input state.
#include "mijpeg_internal.h"

namespace mj {

namespace {
constexpr int kRing = 512;     // staging bytes per wave: < 256 left over + <= 256 new
}

__global__ __launch_bounds__(256) void k_destuff(const uint8_t *__restrict__ blob, const DevSegment *__restrict__ segs,
                                                 int64_t n_segs, uint32_t *__restrict__ stream,
                                                 int32_t *__restrict__ seg_bits) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][kRing];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
    if (seg >= n_segs) return;                                  // wave-uniform; no block barriers below
    uint8_t *stage = s_stage[wave];
    const int64_t begin = segs[seg].begin;
    const int len = __builtin_amdgcn_readfirstlane(segs[seg].len);
    const int64_t abase = begin & ~(int64_t)3;
    const int lead = (int)(begin - abase);
    const int span = lead + len;                                // bytes from abase to the segment's end
    const uint32_t *src = reinterpret_cast<const uint32_t *>(blob + abase);
    uint32_t *out = stream + (begin >> 2) + seg;

    int fill = 0, base = 0, total = 0;                          // wave-uniform: staged bytes, ring origin, kept bytes
    uint32_t carry = 0;                                         // 1 = the next valid byte is dropped
    for (int c0 = 0; c0 < span; c0 += 256) {
        const int p = c0 + 4 * lane;                            // this lane's first byte, relative to abase
        const uint32_t w = p < span ? src[p >> 2] : 0u;
        // valid bytes of the dword: lead <= p + i < span
        uint32_t k0 = 0, k1 = 0, s0 = 0, s1 = 1, vm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = p + i >= lead && p + i < span;
            const bool isff = ((w >> (8 * i)) & 0xFFu) == 0xFFu;
            vm |= valid ? 1u << i : 0u;
            // state 1: this byte is dropped and the state clears; state 0: kept, next state = (byte == 0xFF)
            k0 |= (valid && !s0) ? 1u << i : 0u;
            k1 |= (valid && !s1) ? 1u << i : 0u;
            s0 = valid ? (s0 ? 0u : (uint32_t)isff) : s0;
            s1 = valid ? (s1 ? 0u : (uint32_t)isff) : s1;
        }
        const uint64_t A = __ballot(s0 != 0), B = __ballot(s1 != 0);
        const uint64_t V = __ballot(vm != 0);
        // input state of lane L+1 = output of lane L; start from "independent of the input" (A) and repair, in lane
        // order, the lanes whose output does depend on it (rare: an 0xFF in the first byte, runs of 0xFF)
        uint64_t S = (A << 1) | carry;
        uint32_t carry_out = (uint32_t)(A >> 63);
        uint64_t dep = (A ^ B) & V;
        while (dep) {
            const int L = __builtin_ctzll(dep);
            dep &= dep - 1;
            const uint32_t o = (uint32_t)((((S >> L) & 1) ? B : A) >> L) & 1u;
            if (L == 63) carry_out = o;
            else S = (S & ~(2ull << L)) | ((uint64_t)o << (L + 1));
        }
        const uint32_t keep = ((S >> lane) & 1) ? k1 : k0;
        const int cnt = __builtin_popcount(keep);
        const uint64_t b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
        const uint64_t below = (1ull << lane) - 1;
        int pos = base + fill + __builtin_popcountll(b0 & below) + 2 * __builtin_popcountll(b1 & below) +
                  4 * __builtin_popcountll(b2 & below);
        const int tot = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((keep >> i) & 1) {
                stage[pos & (kRing - 1)] = (uint8_t)(w >> (8 * i));
                ++pos;
            }
        }
        fill += tot;
        total += tot;
        carry = carry_out;
        if (fill >= 256) {                                       // one full 256-byte line, coalesced
            const uint32_t d = *reinterpret_cast<const uint32_t *>(stage + ((base + 4 * lane) & (kRing - 1)));
            out[lane] = __builtin_bswap32(d);
            out += 64;
            base = (base + 256) & (kRing - 1);
            fill -= 256;
        }
    }
    // tail: zero-pad to a whole dword; nothing is written past it (the next dword belongs to the next segment)
    if (lane < 4) stage[(base + fill + lane) & (kRing - 1)] = 0;
    const int n_dw = (fill + 3) >> 2;
    if (lane < n_dw) {
        const uint32_t d = *reinterpret_cast<const uint32_t *>(stage + ((base + 4 * lane) & (kRing - 1)));
        out[lane] = __builtin_bswap32(d);
    }
    if (lane == 0) seg_bits[seg] = total * 8;
}

hipError_t launch_destuff(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                          uint32_t *out_stream, int32_t *seg_bits) {
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_destuff, dim3((unsigned)((n_segs + 3) / 4)), dim3(256), 0, stream, blob, segs, n_segs, out_stream,
                       seg_bits);
    return hipGetLastError();
}

}  // namespace mj
