// Stage 1 for progressive files, large batches: one AC scan segment per LANE (round 5).
//
// progressive_fast.hip gives every (image, scan, restart segment) a wavefront: one lane walks, 63 look symbols up for it.  That is
// the right shape for a few files — the time is one image's chain through its last refinement — and the wrong one for a thousand:
// from ~1 000 files on the chip runs out of instruction issue (eight walks per SIMD, SALU 796 M against VALU 428 M per step,
// r04d) and the batch time grows with the batch (1 024 files 67 ms, 2 048 105 ms, 8 192 510 ms).  Here the 64 lanes of a wavefront
// are the same scan of 64 DIFFERENT images, each lane a complete serial decoder: its own bit position in its segment's stage-0
// stream, its own Huffman table, its own block.  What makes that possible for the refining scans (jpeg_decoder.py:1122-1298),
// whose walk needs the block's history — which coefficients earlier scans left non-zero — is a 64-bit mask per block in HBM
// (`nzmask`, zig-zag positions), kept up to date by every AC scan as it places coefficients: a lane reads 8 bytes per block
// instead of the block, does the reference's run / queue bookkeeping (:1184-1215) as mask arithmetic in two registers, and the
// coefficient store is only ever WRITTEN — new coefficients with plain 2-byte stores, corrections with fire-and-forget atomic
// ORs: the reference's refinement is `value |= bit << Al` on the int16 (:1114), which needs no old value.
// (MJ_FLAG_SPEC_REFINE — T.81's decrement for negative values — does need it and keeps the wavefront walks.)
// DC scans (a twentieth of the work, and their refinement has no chain at all) stay with progressive_fast.hip; the band
// pipelining (launch `step` lets a scan of dependency level L do band step - L) and what a scan parks between its bands
// (DevProgState: bit position, end-of-band run, error, next MCU) are the same.
//
// Huffman symbols: a 9-bit LUT per lane in LDS (1 KiB: len << 8 | symbol), and for longer codes the canonical code book —
// left-aligned upper limits per length, symbol offsets, symbol values, 320 bytes per lane, also in LDS: sixteen compares give
// the length.  A global-memory fallback would be met by some lane in most turns and cost every lane its latency.
#include "mijpeg_internal.h"

namespace mj {

namespace {
constexpr int kL9Bits = 9;
constexpr int kLutStride = (1 << kL9Bits) * 2 + 4;      // bytes per lane: an odd number of dwords apart
constexpr int kCanonStride = kProgCanonBytes + 4;
constexpr int kLaneLds = 64 * kLutStride + 64 * kCanonStride + 64;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;

__device__ __forceinline__ uint64_t from_bit(int k) { return k >= 64 ? 0 : ~(uint64_t)0 << k; }          // bits k..63
__device__ __forceinline__ uint64_t bit_range(int a, int b) { return from_bit(a) & ~from_bit(b); }        // bits a..b-1

// One lane's view of its segment's stage-0 stream: bb = the bits from `pos` on (bit 63 first), valid up to `top`; nxt = the
// dword behind, asked for a refill ahead.  Behind the segment's end the stream reads as zeros (jpeg_decoder.py:689-693).
struct LaneReader {
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

__device__ __forceinline__ int extend(uint32_t v, int s) {   // bin_twos_complement (:1636-1646)
    return (v >> (s - 1)) ? (int)v : (int)v - ((1 << s) - 1);
}

typedef const uint16_t __attribute__((address_space(3))) *lds_cu16;
typedef const uint8_t __attribute__((address_space(3))) *lds_cu8;

// next_huffval (:951-961): the symbol whose code starts at the reader's position, -1 = none
__device__ __forceinline__ int decode(LaneReader &r, uint32_t lut, uint32_t canon) {
    r.ensure();
    const uint32_t w16 = (uint32_t)(r.bb >> 48);
    const uint32_t e = *(lds_cu16)(uintptr_t)(lut + ((w16 >> (16 - kL9Bits)) << 1));
    int len = (int)(e >> 8), hv = (int)(e & 255u);
    if (len == 0) {                                          // longer than nine bits: the length from the sixteen upper limits
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
}  // namespace

__global__ __launch_bounds__(64) void k_progressive_lanes(const uint32_t *__restrict__ stream, const int32_t *__restrict__ seg_bits,
                                                          const DevProgSeg *__restrict__ segs, int n_segs,
                                                          const DevProgScan *__restrict__ scans, const DevImage *__restrict__ images,
                                                          const uint16_t *__restrict__ lut9, const uint8_t *__restrict__ canon_tabs,
                                                          int16_t *__restrict__ coef, uint64_t *__restrict__ nzmask,
                                                          int32_t *__restrict__ status, int tr, DevProgState *__restrict__ states,
                                                          int step, int rows_per_band) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int seg_id = blockIdx.x * 64 + lane;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)smem;
    const uint32_t my_lut = lds0 + (uint32_t)lane * kLutStride, my_canon = lds0 + 64u * kLutStride + (uint32_t)lane * kCanonStride;
    unsigned char *s_nat = smem + 64 * kLutStride + 64 * kCanonStride;
    {   // zig-zag position -> place in the stored block ([v][u], or [u][v] for the row-major stage 2)
        constexpr uint8_t nat[64] = {0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        int n = 0;
#pragma unroll
        for (int z = 0; z < 64; ++z) n = lane == z ? nat[z] : n;
        s_nat[lane] = (unsigned char)(tr ? ((n & 7) << 3 | n >> 3) : n);
    }
    const bool have = seg_id < n_segs;
    const DevProgSeg *sg = segs + (have ? seg_id : 0);
    const DevProgScan *sc = scans + sg->scan;
    const DevImage *im = images + sc->image;
    const int ss = sc->ss, se = sc->se, al = sc->al;
    const bool refining = sc->ah != 0;
    const int c = sc->comp[0];

    // which part of the scan this launch does (progressive.hip: the same bands)
    const int band = step - sc->level;
    const int smh = sc->mcu_count_h;
    int b_lo = sg->mcu0, b_hi = sg->mcu0 + sg->n_mcu;
    {
        const int64_t mpb = (int64_t)smh * rows_per_band * im->comp_v[c];
        const int64_t lo = (int64_t)band * mpb, hi = lo + mpb;
        b_lo = (int)max((int64_t)b_lo, lo);
        b_hi = (int)min((int64_t)b_hi, hi);
    }
    bool active = have && band >= 0 && b_lo < b_hi;
    const bool resume = b_lo != sg->mcu0, finish = b_hi == sg->mcu0 + sg->n_mcu;
    DevProgState *st = states + (have ? seg_id : 0);
    uint32_t pos0 = 0;
    int eobrun = 0;
    if (active && resume) {
        if (st->err != 0 || st->mcu_next != b_lo) active = false;        // the scan failed earlier (its status is set)
        else { pos0 = (uint32_t)st->pos; eobrun = st->eobrun; }
    }
    // this lane's table: 9-bit LUT and canonical code book into LDS (every lane its own: files with optimised tables)
    if (active) {
        const int t = sc->ac_tab[0];
        const u32x4 *src = reinterpret_cast<const u32x4 *>(lut9 + (size_t)t * (1 << kL9Bits));
        for (int i = 0; i < (1 << kL9Bits) * 2 / 16; ++i)
            *(u32x4_u __attribute__((address_space(3))) *)(uintptr_t)(my_lut + 16 * i) = src[i];
        const u32x4 *cs = reinterpret_cast<const u32x4 *>(canon_tabs + (size_t)t * kProgCanonBytes);
        for (int i = 0; i < kProgCanonBytes / 16; ++i)
            *(u32x4_u __attribute__((address_space(3))) *)(uintptr_t)(my_canon + 16 * i) = cs[i];
    }
    __syncthreads();
    if (!active) return;

    LaneReader rd;
    const int total_bits = seg_bits[sg->stream_slot];
    rd.init(stream + (sg->begin >> 2) + sg->stream_slot, (uint32_t)((total_bits + 31) >> 5), pos0);

    // frame geometry (interleaved block order of the coefficient store)
    const int bpm = im->blocks_per_mcu, fmx = im->mcu_count_h;
    const int h = im->comp_h[c], v = im->comp_v[c], first = im->comp_first[c];
    int16_t *cbase = coef + im->block_off * 64;
    uint64_t *mbase = nzmask + im->block_off;
    const uint64_t band_mask = bit_range(ss, se + 1);
    const uint32_t orv = (uint32_t)(uint16_t)(int16_t)(1 << al);
    int err = 0;

    int by = b_lo / smh, bx = b_lo - by * smh;
    for (int m = b_lo; m < b_hi && !err; ++m) {
        const int mx = bx / h, my = by / v;
        const int64_t blk = (int64_t)(my * fmx + mx) * bpm + first + (by - my * v) * h + (bx - mx * h);
        int16_t *p = cbase + blk * 64;
        bx = bx + 1 == smh ? 0 : bx + 1;
        by += bx == 0;
        if (!refining) {
            // -------- first scan of the band: only writes (:1177-1179, :1225, :1248-1250)
            if (eobrun > 0) { --eobrun; continue; }
            uint64_t placed = 0;
            int k = ss;
            while (k <= se) {
                const int hv = decode(rd, my_lut, my_canon);
                if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                const int r = hv >> 4, s = hv & 15;
                if (hv == 0) { eobrun = 1; break; }
                if (s == 0 && r != 15) { rd.ensure(); eobrun = (1 << r) + (int)rd.take(r); break; }
                k += (hv == 0xF0) ? 16 : r;
                if (s > 0) {
                    if (k > 63) { err = MJ_ST_OVERRUN; break; }
                    rd.ensure();
                    const int16_t val = (int16_t)(extend(rd.take(s), s) << al);
                    p[s_nat[k]] = val;
                    placed |= (uint64_t)(val != 0) << k;
                    ++k;
                }
            }
            if (eobrun > 0) --eobrun;          // the band that raised the run counts as its first one
            if (placed) {
                atomicOr(reinterpret_cast<unsigned int *>(mbase + blk), (unsigned int)placed);
                atomicOr(reinterpret_cast<unsigned int *>(mbase + blk) + 1, (unsigned int)(placed >> 32));
            }
        } else {
            // -------- refining scan (:1122-1298): the block's history is its mask; `ones` collects the coefficients whose
            // correction bit is 1
            uint64_t nz = mbase[blk], ones = 0, placed = 0;
            auto refine_queue = [&](uint64_t queue) {       // one correction bit per queued coefficient, in ascending order (:1107-1115)
                while (queue) {
                    rd.ensure();
                    const int n = min(32, __builtin_popcountll(queue));
                    const uint32_t bits = rd.take(n);
                    for (int i = n - 1; i >= 0; --i) {
                        const int j = __builtin_ctzll(queue);
                        queue &= queue - 1;
                        ones |= (uint64_t)((bits >> i) & 1u) << j;
                    }
                }
            };
            if (eobrun > 0) {                   // inside an EOB run: every non-zero coefficient of the band gets a bit
                refine_queue(nz & band_mask);
                --eobrun;
            } else {
                int k = ss;
                while (k <= se) {
                    const int hv = decode(rd, my_lut, my_canon);
                    if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                    const int r = hv >> 4, s = hv & 15;
                    if (hv == 0) { eobrun = 1; break; }
                    if (s == 0 && r != 15) { rd.ensure(); eobrun = (1 << r) + (int)rd.take(r); break; }
                    const int zr = (hv == 0xF0) ? 16 : r;
                    // pass zr zero coefficients from k on (:1184-1193)
                    uint64_t zeros = ~nz & from_bit(k);
                    int pnext = k;
                    if (zr > 0) {
                        for (int i = 1; i < zr; ++i) zeros &= zeros - 1;
                        if (zeros == 0) { err = MJ_ST_OVERRUN; break; }
                        pnext = __builtin_ctzll(zeros) + 1;
                    }
                    uint64_t queue = nz & bit_range(k, pnext);
                    k = pnext;
                    if (s > 0) {
                        rd.ensure();
                        const int16_t val = (int16_t)(extend(rd.take(s), s) << al);     // value bits precede the corrections (:1202)
                        const uint64_t z2 = ~nz & from_bit(k);                        // next zero position (:1212-1215)
                        if (z2 == 0) { err = MJ_ST_OVERRUN; break; }
                        const int pz = __builtin_ctzll(z2);
                        queue |= nz & bit_range(k, pz);
                        k = pz;
                        p[s_nat[k]] = val;                                            // (:1225)
                        const uint64_t bit = (uint64_t)(val != 0) << k;
                        nz |= bit;
                        placed |= bit;
                        ++k;
                    }
                    refine_queue(queue);                                              // (:1231-1232)
                }
                if (!err && eobrun > 0) {       // rest of this band, then the run continues in the next blocks
                    refine_queue(nz & bit_range(k, se + 1));
                    --eobrun;
                }
            }
            // value |= 1 << Al (:1114) wherever the correction bit was 1: no old value needed
            while (ones) {
                const int j = __builtin_ctzll(ones);
                ones &= ones - 1;
                const int at = s_nat[j];
                atomicOr(reinterpret_cast<unsigned int *>(p + (at & ~1)), orv << (16 * (at & 1)));
            }
            if (placed) {
                atomicOr(reinterpret_cast<unsigned int *>(mbase + blk), (unsigned int)placed);
                atomicOr(reinterpret_cast<unsigned int *>(mbase + blk) + 1, (unsigned int)(placed >> 32));
            }
        }
    }
    if (!finish) {
        st->pos = (int32_t)rd.pos; st->eobrun = eobrun; st->err = err; st->mcu_next = b_hi;
    } else if (!err) {
        if ((int)rd.pos > total_bits) err = MJ_ST_OVERRUN;                          // bits were read from behind the end
        else if (!sg->last && total_bits - (int)rd.pos >= 8) err = MJ_ST_DESYNC;   // a whole unread byte before the next restart marker
    }
    if (err) atomicMax(status + sc->image, err);
}

hipError_t launch_progressive_lanes(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevProgSeg *segs, int n_segs,
                                    const DevProgScan *scans, const DevImage *images, const uint16_t *lut9, const uint8_t *canon,
                                    int16_t *coef, uint64_t *nzmask, int32_t *status, int transposed, DevProgState *states, int step,
                                    int rows_per_band) {
    if (n_segs == 0) return hipSuccess;
    static bool attr_set[kMaxDevices] = {false};
    if (!attr_set[current_device()]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_progressive_lanes), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[current_device()] = true;
    }
    hipLaunchKernelGGL(k_progressive_lanes, dim3((unsigned)((n_segs + 63) / 64)), dim3(64), kLaneLds, stream, dstream, seg_bits, segs, n_segs,
                       scans, images, lut9, canon, coef, nzmask, status, transposed, states, step, rows_per_band);
    return hipGetLastError();
}

}  // namespace mj
