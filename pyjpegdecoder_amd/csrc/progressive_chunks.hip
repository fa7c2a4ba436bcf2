// Stage 1 for progressive files, large batches: the AC FIRST scans cut into self-synchronising chunks (round 5).
//
// A first AC scan (jpeg_decoder.py:1122-1179, :1236-1250) is a plain Huffman stream: (run, size) symbols with their value bits,
// ZRL, and end-of-band runs EOBn that close this block and the next 2^n + extra - 1.  Nothing in it depends on what earlier scans
// left in the blocks — unlike a refining scan, whose correction bits are counted by the block's history — so it can be walked the
// way huffman_sync.hip walks files without restart markers: a chunk of the segment's stage-0 stream per LANE from a guessed state
// (bit position, "a block starts here"), counting blocks; the chunks whose guess was wrong walked again from their predecessor's
// exit state; a prefix sum over the chunks' block counts; and then every chunk's stretch of whole blocks decoded and placed by a
// lane of its own.  progressive_fast.hip gives each such scan a wavefront that walks alone; on lanes the same symbols cost a
// sixtieth of the issue slots.  What that buys depends on the batch: up to ~1 500 files a batch lasts as long as ONE image's chain
// through its last refinement (the band launches of 1 024 files take 0.93 ms each with the first AC scans in them and 0.89 ms
// without), and this pass in front of the pipeline only adds its 4.5 ms per 1 024 files; beyond, where the wavefront walks run
// out of instruction issue, it takes a third of their work away: 2 048 files 104.2 -> 93.5 ms, 4 096: 204.9 -> 184.1
// (profiles/r05_progressive_chunks.txt).  Taken from 2 048 images on (MJ_PROG_CHUNKS).  The refining scans stay wavefront walks
// and find the first scans' coefficients complete when the band pipeline starts (their dependency levels drop accordingly).
//
// Tables: Pillow / libjpeg write optimised tables per progressive file, so a batch has thousands.  A wavefront's lanes are chunks
// of ONE scan segment (the chunk list is padded to whole wavefronts per segment), so a wavefront keeps one table in LDS — the 11-bit
// LUT of the wavefront walks (len << 8 | symbol) and, for longer codes, the canonical code book (per length the left-aligned upper limit of its codes,
// first symbol minus first code, the symbol values): sixteen compares give the length.  The repair launch's lanes belong to
// different segments and keep a table each (a 9-bit LUT + the code book: 1.3 KiB per lane).
#include <type_traits>

#include "mijpeg_internal.h"

namespace mj {

namespace {
constexpr int kL9 = 512;                                  // entries of the 9-bit LUT
constexpr int kTabBytes = kL9 * 2 + kProgCanonBytes;      // one table in LDS: LUT, then the code book
constexpr int kTabStride = kTabBytes + 4;                 // ... per lane in the repair launch: an odd number of dwords apart

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef const uint16_t __attribute__((address_space(3))) *lds_cu16;
typedef const uint8_t __attribute__((address_space(3))) *lds_cu8;

__device__ __forceinline__ uint64_t pc_state(uint32_t pos, int k) { return (uint64_t)pos | ((uint64_t)(uint32_t)k << 40); }

// One lane's view of its segment's stage-0 stream: bb = the bits from `pos` on (bit 63 first), valid up to `top`; nxt = the dword
// behind, asked for a refill ahead (many wavefronts per SIMD hide the latency).  Behind the end the stream reads as zeros.
struct PcReader {
    const uint32_t *sw;
    uint32_t n_dw, pos, top, nxt;
    uint64_t bb;
    __device__ __forceinline__ uint32_t dword(uint32_t d) const { return d < n_dw ? sw[d] : 0u; }
    __device__ __forceinline__ void init(const uint32_t *sw_, uint32_t n_dw_, uint32_t p) {
        sw = sw_; n_dw = n_dw_; pos = p;
        const uint32_t d = p >> 5;
        bb = (((uint64_t)dword(d) << 32) | dword(d + 1)) << (p & 31u);
        top = (p & ~31u) + 64u;
        nxt = dword(top >> 5);
    }
    __device__ __forceinline__ void ensure() {               // at least 32 bits behind pos
        const uint32_t bc = top - pos;
        if (bc <= 32u) {
            bb |= (uint64_t)nxt << (32u - bc);
            top += 32u;
            nxt = dword(top >> 5);
        }
    }
    __device__ __forceinline__ uint32_t take(int n) {        // n <= 32, ensure() first
        if (n == 0) return 0u;
        const uint32_t v = (uint32_t)(bb >> (64 - n));
        bb <<= n;
        pos += (uint32_t)n;
        return v;
    }
};

// next_huffval (:951-961): the symbol whose code starts at the reader's position, -1 = none.  tab = LDS address of the table.
template <int LB>
__device__ __forceinline__ int pc_decode(PcReader &r, uint32_t tab) {
    r.ensure();
    const uint32_t w16 = (uint32_t)(r.bb >> 48);
    const uint32_t e = *(lds_cu16)(uintptr_t)(tab + ((w16 >> (16 - LB)) << 1));
    int len = (int)(e >> 8), hv = (int)(e & 255u);
    if (len == 0) {                                          // longer than the index: the length from the sixteen upper limits
        const uint32_t canon = tab + (2u << LB);
        int n = 0;
#pragma unroll
        for (int l = 0; l < 16; ++l) n += w16 >= (uint32_t)*(lds_cu16)(uintptr_t)(canon + 2 * l) ? 1 : 0;
        if (n >= 16) return -1;
        len = n + 1;
        const int base = (int)(int16_t)*(lds_cu16)(uintptr_t)(canon + 32 + 2 * n);
        hv = (int)*(lds_cu8)(uintptr_t)(canon + 64 + ((base + (int)(w16 >> (16 - len))) & 255));
    }
    r.bb <<= len;
    r.pos += (uint32_t)len;
    return hv;
}

// table t to LDS address dst.  SHARED (a wavefront's one table): the 11-bit LUT the wavefront walks use (lut11p, 4 KiB) + the code
// book; else (a lane's own, repair launch): the 9-bit LUT + the code book (tabs + t * kTabBytes)
template <bool SHARED>
__device__ __forceinline__ void pc_load_table(const uint8_t *tabs, const uint16_t *lut11p, int t, uint32_t dst, int lane) {
    const u32x4 *cb = reinterpret_cast<const u32x4 *>(tabs + (size_t)t * kTabBytes + kL9 * 2);
    if constexpr (SHARED) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(lut11p + (size_t)t * (1 << kProgLutBits));
        for (int i = lane; i < (2 << kProgLutBits) / 16; i += 64) *(u32x4_u __attribute__((address_space(3))) *)(uintptr_t)(dst + 16 * i) = src[i];
        if (lane < kProgCanonBytes / 16) *(u32x4_u __attribute__((address_space(3))) *)(uintptr_t)(dst + (2 << kProgLutBits) + 16 * lane) = cb[lane];
    } else {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(tabs + (size_t)t * kTabBytes);
        for (int i = 0; i < kTabBytes / 16; ++i) *(u32x4_u __attribute__((address_space(3))) *)(uintptr_t)(dst + 16 * i) = src[i];
    }
}
constexpr int kSharedTab = (2 << kProgLutBits) + kProgCanonBytes + 16;      // bytes of a wavefront's table in LDS
}  // namespace

struct PcItem { int32_t c, pad; uint64_t entry; };
struct PcArgs {
    const uint32_t *stream; const int32_t *seg_bits; const DevAcSeg *segs; const uint8_t *tabs; const uint16_t *lut11p;
    const DevChunk *chunks; int64_t n_chunks; int32_t cbits, warm;
    uint64_t *exit_state; DevChunkOut *outs;
    const PcItem *items; const int32_t *n_items; int32_t max_links; int32_t *owner;
};

// The counting walks.  REPAIR = false: every chunk from the guess "a block starts `warm` bits in front of the chunk"; true: the
// listed chunks from their predecessor's exit state, walking on where a chunk now leaves in another state than before
// (huffman_sync.hip: k_count — the same protocol, the same records; here a record counts blocks, not MCUs, and carries no DC sums).
template <bool REPAIR>
__global__ __launch_bounds__(256) void k_pc_count(PcArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)smem;
    int32_t n_items = 0;
    if constexpr (REPAIR) {
        n_items = *A.n_items;
        if ((int64_t)blockIdx.x * 64 >= n_items) return;
    }
    int64_t c;
    bool have;
    uint64_t my_entry = 0;
    if constexpr (REPAIR) {
        const int64_t i = (int64_t)blockIdx.x * 64 + tid;
        have = i < n_items;
        const PcItem it = A.items[have ? i : 0];
        c = it.c; my_entry = it.entry;
    } else {
        c = (int64_t)blockIdx.x * 256 + tid;
        have = c < A.n_chunks;
    }
    DevChunk ch = A.chunks[have ? c : 0];
    have = have && ch.seg >= 0;                                  // (the list is padded to whole wavefronts per segment)
    // the wavefront's segment (first walk: every lane's; repair: each lane its own)
    const int64_t cw = (int64_t)blockIdx.x * 256 + wave * 64;
    const int seg_w = REPAIR ? ch.seg : __builtin_amdgcn_readfirstlane(cw < A.n_chunks ? A.chunks[cw].seg : -1);
    const DevAcSeg sg = A.segs[have ? ch.seg : (seg_w >= 0 ? seg_w : 0)];
    const uint32_t tab = REPAIR ? lds0 + (uint32_t)lane * kTabStride : lds0 + (uint32_t)wave * kSharedTab;
    constexpr int LB = REPAIR ? 9 : kProgLutBits;
    if constexpr (REPAIR) { if (have) pc_load_table<false>(A.tabs, A.lut11p, sg.table, tab, lane); }
    else pc_load_table<true>(A.tabs, A.lut11p, sg.table, tab, lane);
    __syncthreads();

    const int ss = sg.ss, se = sg.se;
    const uint32_t nbits = have ? (uint32_t)A.seg_bits[sg.stream_slot] : 0u;
    const uint32_t cbits = (uint32_t)A.cbits;
    // (no symbol is started in the stream's last seven bits: they may be the padding of its last byte, which is no code — and what
    // ends there cannot start a stretch of its own anyway: the last stretch runs to the segment's last block whatever was counted)
    const uint32_t nbits_w = nbits > 7u ? nbits - 7u : 0u;
    auto chunk_limit = [&](int j) { const uint64_t l = (uint64_t)(j + 1) * cbits; return (uint32_t)(l < nbits_w ? l : nbits_w); };
    const uint32_t own_start = (uint32_t)ch.j * cbits;
    if constexpr (!REPAIR) my_entry = ch.j == 0 ? pc_state(0, ss) : pc_state(own_start > (uint32_t)A.warm ? own_start - (uint32_t)A.warm : 0u, ss);
    uint32_t limit = have ? chunk_limit(ch.j) : 0u;
    PcReader rd;
    rd.init(A.stream + sg.stream_dw, (nbits + 31u) >> 5, (uint32_t)my_entry);
    int k = (int)((my_entry >> 40) & 0xFF);
    int blocks = 0, bnd_blocks = 0;
    uint32_t bnd_pos = ~0u;
    bool damaged = false, dead = false;                  // dead: the walk met bits that are no code (it stops; the placing walk will say so)

    // one symbol: the position moves on, blocks are counted where they end (an end-of-band run ends all its blocks at once)
    auto step = [&](uint32_t lim, auto rec_tag) {
        constexpr bool REC = decltype(rec_tag)::value;
        if (dead || rd.pos >= lim) return;
        const int hv = pc_decode<LB>(rd, tab);
        bool ended = false;
        if (hv < 0) { dead = true; return; }                 // bits that are no code: a wrong guess's walk (the repair comes), or a damaged stream
                                                             // (its records never fit together: the image goes to the wavefront walks, which say so)
        const int r = hv >> 4, s = hv & 15;
        if (hv == 0) { blocks += 1; ended = true; }
        else if (s == 0 && r != 15) { rd.ensure(); blocks += (1 << r) + (int)rd.take(r); ended = true; }
        else {
            k += hv == 0xF0 ? 16 : r;
            if (s > 0) {
                if (k > 63) damaged = true;
                rd.ensure();
                (void)rd.take(s);
                ++k;
            }
            if (k > se) { blocks += 1; ended = true; }
        }
        if (ended) {
            k = ss;
            if constexpr (REC) {
                if (bnd_pos == ~0u && rd.pos < lim) { bnd_pos = rd.pos; bnd_blocks = blocks; }
            }
        }
    };
    if constexpr (!REPAIR) {
        const uint32_t lim = ch.j > 0 ? (own_start < limit ? own_start : limit) : 0u;
        while (__builtin_amdgcn_ballot_w64(have && !dead && rd.pos < lim) != 0) step(lim, std::false_type{});
        if (dead) { rd.init(A.stream + sg.stream_dw, (nbits + 31u) >> 5, lim); k = ss; dead = false; }      // (a guess that met no code: any state will do, the repair comes)
        my_entry = pc_state(rd.pos, k);
        damaged = false;
    }
    int links = 0;
    const int32_t origin = (int32_t)c;
    for (;;) {
        if constexpr (REPAIR) {
            if (have) __hip_atomic_fetch_min(A.owner + c, origin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        blocks = 0; bnd_blocks = 0; damaged = false;
        bnd_pos = (k == ss && rd.pos < limit) ? rd.pos : ~0u;               // entered at the start of a block
        while (__builtin_amdgcn_ballot_w64(have && !dead && rd.pos < limit) != 0) step(limit, std::true_type{});
        const uint64_t ex = pc_state(rd.pos, k);
        uint64_t old_exit = ex;
        bool mine = true;
        DevChunkOut o{};
        if (have) {
            o.entry = damaged ? ~0ull : my_entry;            // (a record no chunk's state matches: the image goes to the wavefront walks)
            o.blocks = blocks;
            o.bnd_pos = bnd_pos < limit ? (int32_t)bnd_pos : -1;
            o.bnd_blocks = bnd_blocks;
            if constexpr (!REPAIR) {
                A.outs[c] = o;
                A.exit_state[c] = ex;
            }
        }
        if constexpr (REPAIR) {
            // look ("still the leftmost taker?") and write under the chunk's lock, A.owner[n_chunks + 16 + c]: huffman_sync.hip, k_count
            bool pending = have;
            while (__builtin_amdgcn_ballot_w64(pending) != 0) {
                if (pending) {
                    int32_t *lock = A.owner + A.n_chunks + 16 + c;
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
        bool go = have && mine && !damaged && ex != old_exit && links < A.max_links && c + 1 < A.n_chunks;
        if (go) go = A.chunks[c + 1].seg == ch.seg && (uint64_t)(ch.j + 1) * cbits < nbits;
        have = go;
        if (go) { ++c; ++ch.j; ++links; my_entry = ex; limit = chunk_limit(ch.j); }
        else limit = 0;
        if (__builtin_amdgcn_ballot_w64(go) == 0) break;
    }
}

// the chunks whose record was not computed from their predecessor's exit state: the repair launch's work list
__global__ void k_pc_scan(const DevChunk *__restrict__ chunks, int64_t n_chunks, const DevAcSeg *__restrict__ segs, const DevChunkOut *__restrict__ outs,
                          const uint64_t *__restrict__ exit_state, const int32_t *__restrict__ seg_bits, int cbits,
                          PcItem *__restrict__ items, int32_t *__restrict__ n_items, int32_t *__restrict__ owner) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool add = c < n_chunks;
    uint64_t e = 0;
    if (add) {
        owner[c] = 0x7FFFFFFF;
        owner[n_chunks + 16 + c] = 0;                 // (the chunk's lock: behind the owners and the work list's counter)
        const DevChunk ch = chunks[c];
        add = ch.seg >= 0 && ch.j > 0 && (int64_t)ch.j * cbits < seg_bits[segs[ch.seg >= 0 ? ch.seg : 0].stream_slot];
        if (add) {
            e = exit_state[c - 1];
            add = outs[c].entry != e;
        }
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(add);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, first = __builtin_ctzll(m);
    int base = 0;
    if (lane == first) base = atomicAdd(n_items, (int32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readlane(base, first);
    if (add) {
        PcItem it;
        it.c = (int32_t)c; it.pad = 0; it.entry = e;
        items[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = it;
    }
}

// One stretch of whole blocks per chunk that holds a block start: from there to the next chunk's (or the segment's end); block
// numbers from a prefix sum over the segment's chunks (one workgroup per segment).  A record that is not its predecessor's
// continuation marks the image MJ_ST_UNCONVERGED: the caller decodes it again with MJ_FLAG_NO_SYNC (wavefront walks).
__global__ __launch_bounds__(256) void k_pc_vsegs(const DevAcSeg *__restrict__ segs, const DevChunkOut *__restrict__ outs,
                                                  const int32_t *__restrict__ seg_bits, DevVSeg *__restrict__ vsegs,
                                                  const uint64_t *__restrict__ final_exit, int cbits, int32_t *__restrict__ status) {
    __shared__ int64_t s_blocks[256];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const DevAcSeg sg = segs[seg];
    const int64_t c0 = sg.chunk0, n = sg.n_chunks;
    const int32_t nbits = seg_bits[sg.stream_slot];
    int64_t carry = 0;
    for (int64_t base = 0; base < n; base += 256) {
        const int64_t c = c0 + base + tid, j = base + tid;
        const bool have = j < n;
        DevChunkOut me{};
        if (have) me = outs[c];
        if (have && j > 0 && j * cbits < nbits && me.entry != final_exit[c - 1]) atomicMax(status + sg.image, MJ_ST_UNCONVERGED);
        if (have && j == 0 && me.entry != pc_state(0, sg.ss)) atomicMax(status + sg.image, MJ_ST_UNCONVERGED);
        s_blocks[tid] = have ? me.blocks : 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            int64_t vb = 0;
            if (tid >= d) vb = s_blocks[tid - d];
            __syncthreads();
            s_blocks[tid] += vb;
            __syncthreads();
        }
        if (have) {
            DevVSeg v{};
            v.image = seg;                                   // (the stretch's segment; the placing walk looks the image up)
            if (me.bnd_pos >= 0) {
                const int64_t P = carry + s_blocks[tid] - me.blocks;
                const int64_t start = P + me.bnd_blocks;
                int64_t Pn = P + me.blocks, end = sg.n_blk;
                int32_t bit_end = nbits;
                int last = 1;
                for (int64_t q = c + 1; q < c0 + n; ++q) {
                    const DevChunkOut o = outs[q];
                    if (o.bnd_pos >= 0) { end = Pn + o.bnd_blocks; bit_end = o.bnd_pos; last = 0; break; }
                    Pn += o.blocks;
                }
                v.bit0 = me.bnd_pos;
                v.bit_end = bit_end;
                v.mcu0 = (int32_t)start;                      // blocks of the scan segment, from its first
                v.n_mcu = (int32_t)(end - start);
                if (start < 0 || start >= sg.n_blk || v.n_mcu < 0) v.n_mcu = 0;      // (records that have not settled: the image is marked above)
                else if (v.n_mcu > sg.n_blk - start) v.n_mcu = (int32_t)(sg.n_blk - start);
                v.last = (int16_t)last;
            }
            vsegs[c] = v;
        }
        carry += s_blocks[255];
        __syncthreads();
    }
}

// The placing walk: a lane decodes its stretch's blocks and writes the coefficients (:1177-1179, :1225, :1248-1250: first scans only
// write).  Lanes of a wavefront are consecutive stretches of one segment.
__global__ __launch_bounds__(256) void k_pc_place(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                  const DevAcSeg *__restrict__ segs, const uint8_t *__restrict__ tabs, const uint16_t *__restrict__ lut11p,
                                                  const DevChunk *__restrict__ chunks, int64_t n_chunks, const DevVSeg *__restrict__ vsegs,
                                                  const DevImage *__restrict__ images, int16_t *__restrict__ coef,
                                                  int32_t *__restrict__ status, int tr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)smem;
    unsigned char *s_nat = smem + 4 * kSharedTab;
    if (tid < 64) {   // zig-zag position -> place in the stored block ([v][u], or [u][v] for the row-major stage 2)
        constexpr uint8_t nat[64] = {0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        int n = 0;
#pragma unroll
        for (int z = 0; z < 64; ++z) n = tid == z ? nat[z] : n;
        s_nat[tid] = (unsigned char)(tr ? ((n & 7) << 3 | n >> 3) : n);
    }
    const int64_t c = (int64_t)blockIdx.x * 256 + tid;
    const int64_t cw = (int64_t)blockIdx.x * 256 + wave * 64;
    const int seg_w = __builtin_amdgcn_readfirstlane(cw < n_chunks ? chunks[cw].seg : -1);
    const DevAcSeg sg = segs[seg_w >= 0 ? seg_w : 0];
    const uint32_t tab = lds0 + (uint32_t)wave * kSharedTab;
    pc_load_table<true>(tabs, lut11p, sg.table, tab, lane);
    __syncthreads();
    if (c >= n_chunks || seg_w < 0 || chunks[c].seg < 0) return;
    const DevVSeg v = vsegs[c];
    if (v.n_mcu <= 0) return;

    const DevImage *im = images + sg.image;
    const int ss = sg.ss, se = sg.se, al = sg.al, cc = sg.comp;
    const int bpm = im->blocks_per_mcu, fmx = im->mcu_count_h, smh = sg.mcu_count_h;
    const int h = im->comp_h[cc], vv = im->comp_v[cc], first = im->comp_first[cc];
    int16_t *cbase = coef + im->block_off * 64;
    const int total_bits = seg_bits[sg.stream_slot];
    PcReader rd;
    rd.init(stream + sg.stream_dw, (uint32_t)((total_bits + 31) >> 5), (uint32_t)v.bit0);
    int err = 0;
    int left = v.n_mcu;
    // block coordinates are stepped, not divided out of the block number for every block; sampling factors 1, 2, 4 by shifts
    const int hs = h == 1 ? 0 : (h == 2 ? 1 : (h == 4 ? 2 : -1)), vs = vv == 1 ? 0 : (vv == 2 ? 1 : (vv == 4 ? 2 : -1));
    int by = (sg.first_blk + v.mcu0) / smh, bx = (sg.first_blk + v.mcu0) - by * smh;
    while (left > 0 && !err) {
        const int mx = hs >= 0 ? bx >> hs : bx / h, my = vs >= 0 ? by >> vs : by / vv;
        int16_t *p = cbase + ((int64_t)(my * fmx + mx) * bpm + first + (by - my * vv) * h + (bx - mx * h)) * 64;
        int run = 1, k = ss;
        while (k <= se) {
            const int hv = pc_decode<kProgLutBits>(rd, tab);
            if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
            const int r = hv >> 4, s = hv & 15;
            if (hv == 0) break;
            if (s == 0 && r != 15) { rd.ensure(); run = (1 << r) + (int)rd.take(r); break; }
            k += hv == 0xF0 ? 16 : r;
            if (s > 0) {
                if (k > 63) { err = MJ_ST_OVERRUN; break; }
                rd.ensure();
                const uint32_t raw = rd.take(s);
                const int val = (raw >> (s - 1)) ? (int)raw : (int)raw - ((1 << s) - 1);       // bin_twos_complement (:1636-1646)
                p[s_nat[k]] = (int16_t)(val << al);
                ++k;
            }
        }
        bx += run;
        if (bx >= smh) { const int rows = bx / smh; by += rows; bx -= rows * smh; }
        left -= run;
    }
    if (!err && v.last) {                                    // the stretch that ends its segment: what the other walks check there
        if ((int)rd.pos > total_bits) err = MJ_ST_OVERRUN;
        else if (!sg.last && total_bits - (int)rd.pos >= 8) err = MJ_ST_DESYNC;
    }
    if (!err && left < 0) err = MJ_ST_OVERRUN;               // an end-of-band run that reaches past the stretch: records and stream disagree
    if (err) atomicMax(status + sg.image, err);
}

hipError_t launch_progressive_chunks(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevAcSeg *segs, int n_segs,
                                     const uint8_t *tabs, const uint16_t *lut11p, const DevChunk *chunks, int64_t n_chunks, int cbits, uint64_t *exit_state,
                                     DevChunkOut *outs, void *items, int32_t *n_items, int32_t *owner, DevVSeg *vsegs, const DevImage *images,
                                     int16_t *coef, int32_t *status, int transposed, int max_links) {
    if (n_chunks == 0 || n_segs == 0) return hipSuccess;
    static OncePerDevice attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pc_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    PcArgs A{};
    A.stream = dstream; A.seg_bits = seg_bits; A.segs = segs; A.tabs = tabs; A.lut11p = lut11p; A.chunks = chunks; A.n_chunks = n_chunks;
    A.cbits = cbits; A.warm = cbits / 2; A.exit_state = exit_state; A.outs = outs;
    A.items = reinterpret_cast<const PcItem *>(items); A.n_items = n_items; A.max_links = max_links; A.owner = owner;
    const unsigned wgs = (unsigned)((n_chunks + 255) / 256);
    hipLaunchKernelGGL(k_pc_count<false>, dim3(wgs), dim3(256), 4 * kSharedTab, stream, A);
    if (hipError_t e = launch_fill_words(stream, n_items, 0u, 1); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pc_scan, dim3(wgs), dim3(256), 0, stream, chunks, n_chunks, segs, outs, exit_state, seg_bits, cbits,
                       reinterpret_cast<PcItem *>(items), n_items, owner);
    if (max_links > 0)
        hipLaunchKernelGGL(k_pc_count<true>, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 64 * kTabStride, stream, A);
    hipLaunchKernelGGL(k_pc_vsegs, dim3((unsigned)n_segs), dim3(256), 0, stream, segs, outs, seg_bits, vsegs, exit_state, cbits, status);
    hipLaunchKernelGGL(k_pc_place, dim3(wgs), dim3(256), 4 * kSharedTab + 64, stream, dstream, seg_bits, segs, tabs, lut11p, chunks, n_chunks, vsegs,
                       images, coef, status, transposed);
    return hipGetLastError();
}

}  // namespace mj
