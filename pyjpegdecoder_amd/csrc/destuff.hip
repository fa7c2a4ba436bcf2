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
constexpr int kTile = 1024;    // k_destuff: source bytes per turn, sixteen per lane
constexpr int kStage = 2 * kTile + 64;   // its staging bytes per wave: < 1024 left over + <= 1024 new (+ the reach of a 16-byte write)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;

// bit 7 of every byte that is 0xFF: the low seven bits all set (their sum with 1 carries into bit 7, never beyond) and bit 7 set
__device__ __forceinline__ uint32_t ff_flags(uint32_t w) { return ((w & 0x7F7F7F7Fu) + 0x01010101u) & w & 0x80808080u; }
// flags (bit 7 of each byte) of the first n bytes of dword number i of a lane's sixteen
__device__ __forceinline__ uint32_t valid_flags(int nv, int i) {
    const int n = nv - 4 * i;
    return n >= 4 ? 0x80808080u : (n <= 0 ? 0u : (0x80808080u >> (8 * (4 - n))));
}

// The byte rules for the `len` source bytes at `src` the slow way — four bytes per lane and turn, a two-state transfer function per
// lane, a scalar fix-up over the lanes whose output state depends on their input state: any content, any alignment —, kept bytes
// appended to the staging buffer at `fill`.  Returns the number kept; `carry` (1 = the next byte is dropped) goes in and out.
template <bool STORE>
__device__ __forceinline__ int destuff_range_slow(const uint8_t *src_bytes, int len, uint8_t *stage, int fill, uint32_t &carry, int lane) {
    const uint64_t addr = reinterpret_cast<uint64_t>(src_bytes);
    const int lead = (int)(addr & 3u);
    const int span = lead + len;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(addr & ~(uint64_t)3);
    int total = 0;
    for (int c0 = 0; c0 < span; c0 += 256) {
        const int p = c0 + 4 * lane;                            // this lane's first byte, relative to the aligned base
        const uint32_t w = p < span ? src[p >> 2] : 0u;
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
        // (a lane that holds no byte of the range passes its input state on — A = 0, B = 1 there —, so it is one of these: a range
        // that starts off a dword boundary ends in a turn of one dword, and the state behind its last byte has to come through
        // the 63 empty lanes behind it.  Round 5, found by bench.py's 256-image parity check: the rule here used to skip such
        // lanes — right as long as empty lanes only followed the END of a segment, wrong for a range in the middle of one whose
        // last byte is an 0xFF.)
        uint64_t dep = A ^ B;
        (void)V;
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
        int pos = fill + total + __builtin_popcountll(b0 & below) + 2 * __builtin_popcountll(b1 & below) + 4 * __builtin_popcountll(b2 & below);
        const int tot = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
        if constexpr (STORE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((keep >> i) & 1) {
                    stage[pos] = (uint8_t)(w >> (8 * i));
                    ++pos;
                }
            }
        }
        total += tot;
        carry = carry_out;
    }
    return total;
}
// Round 5: sixteen bytes per lane and turn.  In entropy-coded data an 0xFF is followed by its stuffed 0x00, never by another 0xFF,
// so the sequential rule degenerates: dropped = the byte behind an 0xFF.  A turn takes 1 KiB of the segment: each lane flags its 0xFF
// bytes with three integer operations per dword, the flags moved up one byte (across dwords with v_alignbit, across lanes with a
// DPP shift, across turns with a carry) are the dropped bytes; a lane takes its (at most seven) dropped bytes out of its
// sixteen in registers, fills up from its successor's first bytes, and writes SIXTEEN bytes at its byte offset in the staging
// buffer — gfx950's LDS takes a 16-byte access at any byte address (tools/unaligned_lds_probe.hip), and where two lanes' writes
// overlap they carry the same bytes.  A turn in which some dropped byte is itself an 0xFF (fill bytes in front of a marker,
// damaged data), or a lane has more than seven, goes through the slow step above.  Instructions per source byte: 0.11 against
// 0.4; the kernel had been bound by instruction issue (0.52 ms per 1024 x 1080p, 1.3 TB/s each way).
// One turn: the `tile_len` (<= 1 KiB) source bytes at `src`, kept bytes to the staging buffer at `fill` (STORE) or only counted.
template <bool STORE>
__device__ __forceinline__ int destuff_tile(const uint8_t *src, int tile_len, uint8_t *stage, int fill, uint32_t &carry, int lane) {
    const int p = 16 * lane;
    const int nv = min(16, max(0, tile_len - p));           // this lane's bytes that belong to the range
    u32x4 w = {0u, 0u, 0u, 0u};
    if (nv == 16) {
        w = *reinterpret_cast<const u32x4_u *>(src + p);
    } else if (nv > 0) {                                    // the range's last lane: byte by byte (nothing is read behind it)
        for (int i = 0; i < nv; ++i) w[i >> 2] |= (uint32_t)src[p + i] << (8 * (i & 3));
    }
    uint32_t F[4], D[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) F[i] = ff_flags(w[i]) & valid_flags(nv, i);
    // the flags one byte up: the byte behind an 0xFF
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)(carry << 31), (int)F[3], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    D[0] = __builtin_amdgcn_alignbit(F[0], prev, 24);
#pragma unroll
    for (int i = 1; i < 4; ++i) D[i] = __builtin_amdgcn_alignbit(F[i], F[i - 1], 24);
    uint32_t both = 0;
    int drops = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        D[i] &= valid_flags(nv, i);
        both |= D[i] & F[i];
        drops += __builtin_popcount(D[i]);
    }
    if (__builtin_amdgcn_ballot_w64(both != 0u || drops > 7) != 0) return destuff_range_slow<STORE>(src, tile_len, stage, fill, carry, lane);
    // the state behind the range: its last byte an 0xFF (kept: a dropped one would have sent the turn the slow way)
    const int last_lane = (tile_len - 1) >> 4;
    const uint32_t lastF = (uint32_t)__builtin_amdgcn_readlane((int)F[((tile_len - 1) >> 2) & 3], last_lane);
    // (F[] is an array in registers: select by the wave-uniform index of the last byte's dword)
    carry = (lastF >> (8 * ((tile_len - 1) & 3) + 7)) & 1u;
    const uint64_t b0 = __builtin_amdgcn_ballot_w64(drops & 1), b1 = __builtin_amdgcn_ballot_w64(drops & 2), b2 = __builtin_amdgcn_ballot_w64(drops & 4);
    const int kept = tile_len - (__builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2));
    if constexpr (STORE) {
        const uint64_t below = (1ull << lane) - 1;
        const int drops_before = __builtin_popcountll(b0 & below) + 2 * __builtin_popcountll(b1 & below) + 4 * __builtin_popcountll(b2 & below);
        int nk = nv;
        while (__builtin_amdgcn_ballot_w64(drops > 0) != 0) {             // the highest dropped byte of each lane that has one: out
            if (drops > 0) {
                int q = 0;                                               // its index, 0..15
#pragma unroll
                for (int i = 0; i < 4; ++i) q = D[i] ? 4 * i + ((31 - __builtin_clz(D[i])) >> 3) : q;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t up = i < 3 ? __builtin_amdgcn_alignbit(w[i + 1], w[i], 8) : (w[3] >> 8);    // the sixteen bytes moved down by one
                    const int n = q - 4 * i;                                                                     // bytes of this dword in front of q
                    const uint32_t low = n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << (8 * n)) - 1u));
                    w[i] = (w[i] & low) | (up & ~low);
                    if (4 * i <= q && q < 4 * i + 4) D[i] &= ~(0x80u << (8 * (q & 3)));
                }
                --drops;
                --nk;
            }
        }
        // behind its own bytes (nine at least) a lane writes its successor's first ones: what the successor writes there itself
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
        const uint32_t next1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[1], 0x130, 0xF, 0xF, false);
        if (nk < 16 && nk >= 9) {
            const int sh = 8 * (nk - 8);
            const uint64_t mine = ((uint64_t)w[3] << 32) | w[2], fill_in = ((uint64_t)next1 << 32) | next0;
            const uint64_t both_halves = sh ? (mine & ((1ull << sh) - 1ull)) | (fill_in << sh) : fill_in;
            w[2] = (uint32_t)both_halves;
            w[3] = (uint32_t)(both_halves >> 32);
        }
        if (nv > 0) *reinterpret_cast<u32x4_u *>(stage + fill + min(p, tile_len) - drops_before) = w;
    }
    return kept;
}
}  // namespace

__global__ __launch_bounds__(256) void k_destuff(const uint8_t *__restrict__ blob, const DevSegment *__restrict__ segs,
                                                 int64_t n_segs, uint32_t *__restrict__ stream,
                                                 int32_t *__restrict__ seg_bits) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][kStage];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
    if (seg >= n_segs) return;                                  // wave-uniform; no block barriers below
    uint8_t *stage = s_stage[wave];
    const int64_t begin = segs[seg].begin;
    const int len = __builtin_amdgcn_readfirstlane(segs[seg].len);
    const uint8_t *src = blob + begin;
    uint32_t *out = stream + (begin >> 2) + seg;

    int fill = 0, total = 0;                                    // wave-uniform: staged bytes, kept bytes
    uint32_t carry = 0;                                         // 1 = the next byte is dropped
    for (int t0 = 0; t0 < len; t0 += kTile) {
        const int kept = destuff_tile<true>(src + t0, min(kTile, len - t0), stage, fill, carry, lane);
        fill += kept;
        total += kept;
        if (fill >= kTile) {                                     // 1 KiB of stream, coalesced: sixteen bytes per lane
            u32x4 d = *reinterpret_cast<const u32x4 *>(stage + 16 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = __builtin_bswap32(d[i]);
            *reinterpret_cast<u32x4_u *>(out + 4 * lane) = d;
            out += kTile / 4;
            fill -= kTile;
            if (16 * lane < fill) {                              // what is left moves to the front
                const u32x4 r = *reinterpret_cast<const u32x4 *>(stage + kTile + 16 * lane);
                *reinterpret_cast<u32x4 *>(stage + 16 * lane) = r;
            }
        }
    }
    // tail: zero-pad to a whole dword; nothing is written past it (the next dword belongs to the next segment)
    if (lane < 4) stage[fill + lane] = 0;
    const int n_dw = (fill + 3) >> 2;
    for (int i = lane; i < n_dw; i += 64) out[i] = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(stage + 4 * i));
    if (lane == 0) seg_bits[seg] = total * 8;
}

// ---- long segments: the same rules, piece by piece ------------------------------------------------------------------
// One wavefront per restart segment is fine for restart segments of ordinary size; a file without DRI is ONE segment
// of hundreds of kilobytes and a batch of them left the chip to ~1000 serial walks (2.3 ms per 1024 x 1080p).  So long
// segments are processed in pieces of source bytes, one wavefront each, in two launches: k_destuff_pieces<false>
// counts the bytes each piece keeps, k_destuff_pieces<true> writes them at the offset the counts of the earlier pieces
// add up to.  What a piece must know about its predecessors is little: whether its first byte is dropped — the parity
// of the run of 0xFF bytes right in front of it — and, when writing, that the first and last output dword may be shared
// with its neighbours (those bytes are stored one by one).
namespace {

// is the byte at `pos` dropped?  = is the run of 0xFF bytes ending at pos - 1 (inside [seg_begin, pos)) of odd length
__device__ __forceinline__ uint32_t carry_in_at(const uint8_t *blob, int64_t seg_begin, int64_t pos, int lane) {
    int run = 0;
    for (int64_t q = pos - 1;; q -= 64) {
        const int64_t mine = q - lane;
        const bool in = mine >= seg_begin;
        const bool ff = in && blob[mine] == 0xFFu;
        const uint64_t notff = ~__ballot(ff);
        if (notff) { run += __builtin_ctzll(notff); break; }
        run += 64;
    }
    return (uint32_t)run & 1u;
}

// stream dword `d` (big-endian in memory) from four stream bytes, of which only those in `mask` (bit i = stream byte i)
// belong to this piece
__device__ __forceinline__ void store_stream_dword(uint32_t *d, uint32_t ring_dword, uint32_t mask) {
    if (mask == 0xFu) { *d = __builtin_bswap32(ring_dword); return; }
    uint8_t *bytes = reinterpret_cast<uint8_t *>(d);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((mask >> i) & 1) bytes[3 - i] = (uint8_t)(ring_dword >> (8 * i));
}

}  // namespace

template <bool WRITE>
__global__ __launch_bounds__(256) void k_destuff_pieces(const uint8_t *__restrict__ blob, const DevSegment *__restrict__ segs,
                                                        const DevPiece *__restrict__ pieces, int64_t n_pieces,
                                                        int32_t *__restrict__ kept, uint32_t *__restrict__ stream,
                                                        int32_t *__restrict__ seg_bits) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][kStage];      // (the counting launch stages only what its slow turns write)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t pi = (int64_t)blockIdx.x * 4 + wave;
    if (pi >= n_pieces) return;                                  // wave-uniform; no block barriers below
    uint8_t *stage = s_stage[wave];
    const DevPiece pc = pieces[pi];
    const int64_t seg_begin = segs[pc.seg].begin;
    const int seg_len = __builtin_amdgcn_readfirstlane(segs[pc.seg].len);    // may be shorter than the host's bound (GPU scan)
    const int p_off = __builtin_amdgcn_readfirstlane(pc.off);
    const int p_end = min(p_off + pc.len, seg_len);
    const bool empty = p_off >= p_end;
    const bool last = p_off + pc.len >= seg_len && (p_off < seg_len || p_off == 0);    // this piece holds the segment's end
    if (empty && !(WRITE && last)) { if (!WRITE && lane == 0) kept[pi] = 0; return; }

    const int64_t begin = seg_begin + p_off;
    const int len = empty ? 0 : p_end - p_off;
    const uint8_t *src = blob + begin;

    // where this piece's bytes go: behind everything the segment's earlier pieces keep
    int o = 0;
    if constexpr (WRITE) {
        int part = 0;
        for (int64_t q = pc.first + lane; q < pi; q += 64) part += kept[q];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft);
        o = __builtin_amdgcn_readfirstlane(part);
    }
    uint32_t *out = stream + (seg_begin >> 2) + pc.seg + (o >> 2);
    int fill = o & 3, total = 0;                                 // the first o & 3 staged bytes stand for the predecessor's
    uint32_t head_mask = (0xFu << (o & 3)) & 0xFu;               // ... bytes of the first dword, which are not ours to write
    uint32_t carry = (p_off > 0 && !empty) ? carry_in_at(blob, seg_begin, begin, lane) : 0u;

    for (int t0 = 0; t0 < len; t0 += kTile) {
        const int got = destuff_tile<WRITE>(src + t0, min(kTile, len - t0), stage, fill, carry, lane);
        total += got;
        if constexpr (WRITE) {
            fill += got;
            if (fill >= kTile) {                                 // 1 KiB of stream, coalesced
                if (head_mask != 0xFu) {                         // (the piece's first: its first dword is shared)
                    for (int i = lane; i < kTile / 4; i += 64)
                        store_stream_dword(out + i, *reinterpret_cast<const uint32_t *>(stage + 4 * i), i == 0 ? head_mask : 0xFu);
                    head_mask = 0xFu;
                } else {
                    u32x4 d = *reinterpret_cast<const u32x4 *>(stage + 16 * lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = __builtin_bswap32(d[i]);
                    *reinterpret_cast<u32x4_u *>(out + 4 * lane) = d;
                }
                out += kTile / 4;
                fill -= kTile;
                if (16 * lane < fill) {                          // what is left moves to the front
                    const u32x4 r = *reinterpret_cast<const u32x4 *>(stage + kTile + 16 * lane);
                    *reinterpret_cast<u32x4 *>(stage + 16 * lane) = r;
                }
            }
        }
    }
    if constexpr (!WRITE) {
        if (lane == 0) kept[pi] = total;
    } else {
        // tail: the segment's last piece zero-pads to a whole dword; any other piece leaves the rest of its last dword to
        // its successor
        const int rem = fill & 3;
        if (last && lane < 4) stage[fill + lane] = 0;
        const int n_dw = (fill + 3) >> 2;
        for (int i = lane; i < n_dw; i += 64) {
            uint32_t mask = i == 0 ? head_mask : 0xFu;
            if (i == n_dw - 1 && rem != 0 && !last) mask &= (1u << rem) - 1u;
            store_stream_dword(out + i, *reinterpret_cast<const uint32_t *>(stage + 4 * i), mask);
        }
        if (last && lane == 0) seg_bits[pc.seg] = (o + total) * 8;
    }
}

hipError_t launch_destuff_pieces(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, const DevPiece *pieces,
                                 int64_t n_pieces, int32_t *kept, uint32_t *out_stream, int32_t *seg_bits) {
    if (n_pieces == 0) return hipSuccess;
    const dim3 grid((unsigned)((n_pieces + 3) / 4));
    hipLaunchKernelGGL(k_destuff_pieces<false>, grid, dim3(256), 0, stream, blob, segs, pieces, n_pieces, kept, out_stream, seg_bits);
    hipLaunchKernelGGL(k_destuff_pieces<true>, grid, dim3(256), 0, stream, blob, segs, pieces, n_pieces, kept, out_stream, seg_bits);
    return hipGetLastError();
}

// Restart-marker scan: with MJ_FLAG_GPU_SEGMENT the caller hands over ONE byte range per image —
// first entropy-coded byte up to any bound at or behind the end of the scan (normally the end of the file) — and
// this kernel does what pyjpegdecoder_amd/_parse.py does on the host otherwise (find_entropy_end,
// find_restart_segments): the entropy-coded data ends at the first 0xFF followed by anything but 0x00, 0xFF or
// RSTn; every "FF D0..D7" before that ends a restart segment, the next one starts two bytes later.  The reference
// itself never looks at those bytes (it skips two bytes after every restart_interval MCUs, jpeg_decoder.py:667-669,
// :898-900); a file whose marker count differs from ceil(mcu_count / restart_interval) - 1 is reported as
// MJ_ST_DESYNC, and one whose scan is not followed by EOI as MJ_ST_TAIL (the host parser has to look at it).
// One workgroup per image: its four waves take 4 KiB pieces of the range in turn, append the (rare) marker positions
// to a list in LDS in whatever order they find them and keep the lowest terminator position; the list is then
// rank-sorted and thread r writes segment r.  More markers than the list holds: MJ_ST_TAIL, the host segments it.
namespace {
constexpr int kScanCap = 2048;
constexpr int kScanThreads = 512;        // eight wavefronts per image (round 6; four before: 16 wavefronts per CU left the loads' latency exposed)
}
__global__ __launch_bounds__(kScanThreads) void k_scan_markers(const uint8_t *__restrict__ blob, const DevScanJob *__restrict__ jobs,
                                                     int n_jobs, DevSegment *__restrict__ segs, int32_t *__restrict__ status) {
    __shared__ uint32_t s_pos[kScanCap], s_sorted[kScanCap];
    __shared__ uint32_t s_count, s_term;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DevScanJob *jb = jobs + blockIdx.x;
    const int64_t begin = jb->begin, end = jb->end, first = jb->first_seg;
    const int n_seg = jb->n_seg;
    const int64_t abase = begin & ~(int64_t)15;
    const uint32_t len32 = (uint32_t)(end - begin);
    if (tid == 0) { s_count = 0; s_term = len32; }
    __syncthreads();
    for (int64_t c0 = abase + (int64_t)wave * 4096; c0 < end; c0 += (kScanThreads / 64) * 4096) {
        // everything behind the terminator is somebody else's data (wave-uniform test; the value only ever drops)
        if (c0 > begin && (uint32_t)(c0 - begin) > *(volatile uint32_t *)&s_term) break;
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t pos = c0 + q * 1024 + lane * 16;
            w[q] = pos < end ? *reinterpret_cast<const uint4 *>(blob + pos) : make_uint4(0, 0, 0, 0);
        }
        const uint32_t tail = c0 + 4096 < end ? blob[c0 + 4096] : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t d[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
            // the byte after this lane's sixteen: the next lane's first, the next row's, or the next piece's
            uint32_t nb = __shfl_down(d[0], 1) & 0xFFu;
            const uint32_t wrap = q < 3 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)w[q < 3 ? q + 1 : 3].x) & 0xFFu : tail;
            nb = lane == 63 ? wrap : nb;
            // Nearly every 0xFF of entropy-coded data is followed by its stuffed 0x00: the byte-by-byte look below is only
            // needed where some lane holds an 0xFF in front of a NON-zero byte (a marker, fill bytes, damage) — about one
            // wavefront row in ten of a file with a restart marker per MCU row.  Four dwords per lane, word-parallel: bytes
            // equal to 0xFF, bytes whose successor is not zero (round 6: 0.26 -> 0.1x ms per 1024 x 1080p; the loop below
            // costs ~130 instructions per 16 bytes, this filter ~40)
            {
                uint32_t hit = 0;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const uint32_t wd = d[k2], nxt = k2 < 3 ? d[k2 + 1] : nb;
                    const uint32_t nx = __builtin_amdgcn_alignbit(nxt, wd, 8);                       // each byte's successor
                    const uint32_t ff = ((wd & 0x7F7F7F7Fu) + 0x01010101u) & wd & 0x80808080u;       // 0x80 where the byte is 0xFF
                    const uint32_t nz = (((nx & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nx) & 0x80808080u;     // 0x80 where the successor is not 0
                    hit |= ff & nz;
                }
                if (__builtin_amdgcn_ballot_w64(hit != 0u) == 0) continue;
            }
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
            const int lo = rel < 0 ? -rel : 0, hi = (int)len32 - 1 - rel;           // valid j: lo <= j < hi
            const uint32_t vmask = hi <= 0 || lo >= 16 ? 0u : ((hi >= 16 ? 0xFFFFu : (1u << hi) - 1u) & ~((1u << lo) - 1u));
            rstm &= vmask;
            termm &= vmask;
            if (termm) atomicMin(&s_term, (uint32_t)(rel + __builtin_ctz(termm)));
            while (rstm) {
                const int j = __builtin_ctz(rstm);
                rstm &= rstm - 1;
                const uint32_t slot = atomicAdd(&s_count, 1u);
                if (slot < (uint32_t)kScanCap) s_pos[slot] = (uint32_t)(rel + j);
            }
        }
    }
    __syncthreads();
    const uint32_t T = s_term;                       // range-relative end of the entropy-coded data
    const uint32_t n_found = s_count;
    if (n_found > (uint32_t)kScanCap) {              // pathological restart interval: let the host do it
        if (tid == 0) atomicMax(status + jb->image, MJ_ST_TAIL);
        return;
    }
    // rank sort of the markers in front of the terminator (positions are distinct)
    for (uint32_t i = tid; i < n_found; i += kScanThreads) {
        const uint32_t p = s_pos[i];
        if (p >= T) continue;
        uint32_t r = 0;
        for (uint32_t k2 = 0; k2 < n_found; ++k2) r += s_pos[k2] < p ? 1u : 0u;
        s_sorted[r] = p;
    }
    __shared__ uint32_t s_m;
    if (tid == 0) s_m = 0;
    __syncthreads();
    {
        uint32_t mine = 0;
        for (uint32_t i = tid; i < n_found; i += kScanThreads) mine += s_pos[i] < T ? 1u : 0u;
        if (mine) atomicAdd(&s_m, mine);
    }
    __syncthreads();
    const int m = (int)s_m;                          // markers = segments that end at one
    for (int r = tid; r < n_seg; r += kScanThreads) {
        DevSegment *g = segs + first + r;
        if (r < m && r < n_seg - 1) {
            const uint32_t b0 = r == 0 ? 0u : s_sorted[r - 1] + 2u;
            g->begin = begin + b0;
            g->len = (int32_t)(s_sorted[r] - b0);
        } else if (r == (m < n_seg - 1 ? m : n_seg - 1) && m <= n_seg - 1) {
            const uint32_t b0 = m == 0 ? 0u : s_sorted[m - 1] + 2u;              // the segment the terminator ends
            g->begin = begin + b0;
            g->len = (int32_t)(T > b0 ? T - b0 : 0u);
        } else {
            g->begin = begin + T;
            g->len = 0;
        }
    }
    if (tid == 0) {
        if (m != n_seg - 1) atomicMax(status + jb->image, MJ_ST_DESYNC);
        else if ((int64_t)T + 1 < (int64_t)len32 && blob[begin + T + 1] != 0xD9u) atomicMax(status + jb->image, MJ_ST_TAIL);
    }
}

hipError_t launch_scan_markers(hipStream_t stream, const uint8_t *blob, const DevScanJob *jobs, int n_jobs, DevSegment *segs,
                               int32_t *status) {
    if (n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_markers, dim3((unsigned)n_jobs), dim3(kScanThreads), 0, stream, blob, jobs, n_jobs, segs, status);
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
