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

// Restart-marker scan: with MJ_FLAG_GPU_SEGMENT the caller hands over ONE byte range per image —
// first entropy-coded byte up to any bound at or behind the end of the scan (normally the end of the file) — and
// this kernel does what pyjpegdecoder_amd/_parse.py does on the host otherwise (find_entropy_end,
// find_restart_segments): the entropy-coded data ends at the first 0xFF followed by anything but 0x00, 0xFF or
// RSTn; every "FF D0..D7" before that ends a restart segment, the next one starts two bytes later.  The reference
// itself never looks at those bytes (it skips two bytes after every restart_interval MCUs, jpeg_decoder.py:667-669,
// :898-900); a file whose marker count differs from ceil(mcu_count / restart_interval) - 1 is reported as
// MJ_ST_DESYNC, and one whose scan is not followed by EOI as MJ_ST_TAIL (the host parser has to look at it).
// One wavefront per image, 4 KiB per iteration; markers are rare, so they are emitted by scalar code in file order.
__global__ __launch_bounds__(64) void k_scan_markers(const uint8_t *__restrict__ blob, const DevScanJob *__restrict__ jobs,
                                                    int n_jobs, DevSegment *__restrict__ segs, int32_t *__restrict__ status) {
    const int lane = threadIdx.x;
    const DevScanJob *jb = jobs + blockIdx.x;
    const int64_t begin = jb->begin, end = jb->end, first = jb->first_seg;
    const int n_seg = jb->n_seg;
    const int64_t abase = begin & ~(int64_t)15;
    const int len32 = (int)(end - begin);
    int r = 0;                           // markers seen
    int64_t cur_begin = begin, T = end;
    bool stop = false;
    for (int64_t c0 = abase; c0 < end && !stop; c0 += 4096) {
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t pos = c0 + q * 1024 + lane * 16;
            w[q] = pos < end ? *reinterpret_cast<const uint4 *>(blob + pos) : make_uint4(0, 0, 0, 0);
        }
        const uint32_t tail = c0 + 4096 < end ? blob[c0 + 4096] : 0u;
#pragma unroll
        for (int q = 0; q < 4 && !stop; ++q) {
            const uint32_t d[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
            // the byte after this lane's sixteen: the next lane's first, the next row's, or the next iteration's
            uint32_t nb = __shfl_down(d[0], 1) & 0xFFu;
            const uint32_t wrap = q < 3 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)w[q < 3 ? q + 1 : 3].x) & 0xFFu : tail;
            nb = lane == 63 ? wrap : nb;
            uint32_t rstm = 0, termm = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t b = (d[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                const uint32_t n = j < 15 ? (d[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu : nb;
                const bool isr = (n & 0xF8u) == 0xD0u;
                const bool ist = n != 0u && n != 0xFFu && !isr;
                rstm |= (b == 0xFFu && isr) ? 1u << j : 0u;
                termm |= (b == 0xFFu && ist) ? 1u << j : 0u;
            }
            // bytes of the range whose successor is in the range too: begin <= p, p + 1 < end
            const int rel = (int)(c0 + q * 1024 + lane * 16 - begin);
            const int lo = rel < 0 ? -rel : 0, hi = len32 - 1 - rel;          // valid j: lo <= j < hi
            const uint32_t vmask = hi <= 0 || lo >= 16 ? 0u : ((hi >= 16 ? 0xFFFFu : (1u << hi) - 1u) & ~((1u << lo) - 1u));
            rstm &= vmask;
            termm &= vmask;
            const uint64_t tb = __ballot(termm != 0);
            int64_t tpos = end;
            if (tb) {
                const int L = __builtin_ctzll(tb);
                tpos = c0 + q * 1024 + L * 16 + __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)termm, L));
                T = tpos;
                stop = true;
            }
            uint64_t rb = __ballot(rstm != 0);
            while (rb) {
                const int L = __builtin_ctzll(rb);
                rb &= rb - 1;
                uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)rstm, L);
                while (bits) {
                    const int j = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int64_t p = c0 + q * 1024 + L * 16 + j;
                    if (p > tpos) { rb = 0; break; }
                    if (r + 1 < n_seg && lane == 0) {
                        segs[first + r].begin = cur_begin;
                        segs[first + r].len = (int32_t)(p - cur_begin);
                    }
                    cur_begin = p + 2;
                    ++r;
                }
            }
        }
    }
    if (lane == 0) {
        const int last = r < n_seg - 1 ? r : n_seg - 1;
        if (r <= n_seg - 1) {
            segs[first + last].begin = cur_begin;
            segs[first + last].len = (int32_t)(T > cur_begin ? T - cur_begin : 0);
        }
        for (int k = last + 1; k < n_seg; ++k) { segs[first + k].begin = T; segs[first + k].len = 0; }
        if (r != n_seg - 1) atomicMax(status + jb->image, MJ_ST_DESYNC);
        else if (T + 1 < end && blob[T + 1] != 0xD9u) atomicMax(status + jb->image, MJ_ST_TAIL);
    }
}

hipError_t launch_scan_markers(hipStream_t stream, const uint8_t *blob, const DevScanJob *jobs, int n_jobs, DevSegment *segs,
                               int32_t *status) {
    if (n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_markers, dim3((unsigned)n_jobs), dim3(64), 0, stream, blob, jobs, n_jobs, segs, status);
    return hipGetLastError();
}

hipError_t launch_destuff(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                          uint32_t *out_stream, int32_t *seg_bits) {
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_destuff, dim3((unsigned)((n_segs + 3) / 4)), dim3(256), 0, stream, blob, segs, n_segs, out_stream,
                       seg_bits);
    return hipGetLastError();
}

}  // namespace mj
