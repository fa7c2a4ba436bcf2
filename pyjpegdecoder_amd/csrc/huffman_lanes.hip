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
//   * per-lane bit reader: bytes are fetched as aligned dwords one dword ahead of need (the load is not
//     consumed until a later iteration), unstuffed exactly like the reference's get_bits — whatever
//     follows a 0xFF is dropped (jpeg_decoder.py:676-677) — four at a time when none is 0xFF, else byte-wise.
//   * Huffman tables: an 11-bit primary LUT per table, ALL tables of the batch resident in LDS (this form
//     is chosen when the batch has at most kMaxLaneTables distinct tables, e.g. everybody uses the Annex-K
//     tables); codes longer than 11 bits take a per-lane canonical search (rare).
//   * coefficients: each lane owns a 128-byte block in LDS (row stride 132 B -> conflict-free scatter of
//     "coefficient kk of every lane"); a finished round of 64 blocks leaves as 32 store instructions that each
//     write two full 128-byte lines, in the natural [v][u] order stage 2 reads.
#include <stdlib.h>

#include "mijpeg_internal.h"

namespace mj {

namespace {

constexpr int kLBits = 11;
constexpr int kLSize = 1 << kLBits;
constexpr int kBlkStride = 33;   // dwords per lane block in LDS (32 + 1 pad)

__constant__ uint8_t c_nat_of_zz_l[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct LaneBits {
    uint64_t bb;          // bit buffer, next bit = bit 63
    uint64_t raw;         // fetched, not yet unstuffed bytes (little endian, next byte = low byte)
    const uint32_t *wp;   // next dword to append (the value is already in nxtw)
    uint32_t nxtw;        // prefetched *wp
    int bc;               // valid bits in bb
    int rawn;             // bytes in raw
    int left;             // segment bytes not yet appended to raw
    int pad;              // zero bits fed past the end of the segment
    bool skipnext;        // the next raw byte follows an 0xFF and is dropped

    __device__ __forceinline__ void init(const uint8_t *blob, int64_t begin, int len) {
        const int64_t abase = begin & ~(int64_t)3;
        const int lead = (int)(begin - abase);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(blob + abase);
        raw = (uint64_t)(p[0] >> (8 * lead));
        rawn = min(4 - lead, len);
        left = len - rawn;
        wp = p + 1;
        nxtw = *wp;
        bb = 0; bc = 0; pad = 0; skipnext = false;
    }
    __device__ __forceinline__ void append() {           // caller checked rawn <= 4 && left > 0
        const int take = min(4, left);
        raw |= (uint64_t)nxtw << (8 * rawn);
        rawn += take;
        left -= take;
        ++wp;
        nxtw = *wp;                                       // consumed at the next append, iterations from now
    }
    // one byte raw -> bb with the reference's stuffing rule, or a zero byte past the end
    __device__ __forceinline__ void move_byte() {
        if (rawn <= 4 && left > 0) append();
        if (rawn > 0) {
            const uint32_t b = (uint32_t)raw & 0xFFu;
            raw >>= 8;
            --rawn;
            if (skipnext) skipnext = false;
            else {
                bb |= (uint64_t)b << (56 - bc);
                bc += 8;
                skipnext = b == 0xFFu;
            }
        } else {
            bc += 8;
            pad += 8;
        }
    }
    // make >= 32 bits available in every lane of `need`
    __device__ __forceinline__ void refill(bool need) {
        if (need && rawn <= 4 && left > 0) append();
        const uint32_t w = (uint32_t)raw, nw = ~w;
        if (need && bc <= 32 && rawn >= 4 && !skipnext && (((nw - 0x01010101u) & ~nw & 0x80808080u) == 0)) {
            bb |= (uint64_t)__builtin_bswap32(w) << (32 - bc);
            bc += 32;
            raw >>= 32;
            rawn -= 4;
        }
        while (__any(need && bc < 32)) {
            if (need && bc < 32) move_byte();
        }
    }
    __device__ __forceinline__ uint32_t take(int n) {    // 0 <= n <= 16, branch-free for n == 0
        const uint32_t v = (uint32_t)((bb >> 1) >> (63 - n));
        bb <<= n;
        bc -= n;
        return v;
    }
};

__device__ __forceinline__ int extend(uint32_t raw, int n) {   // bin_twos_complement (:1636-1646); n = 0 -> 0
    const int half = (1 << n) >> 1;
    return (int)raw - (((int)raw < half) ? ((1 << n) - 1) : 0);
}

}  // namespace

__global__ __launch_bounds__(256) void k_huffman_lanes(const uint8_t *__restrict__ blob,
                                                       const DevSegment *__restrict__ segs, int64_t n_segs,
                                                       const DevImage *__restrict__ images,
                                                       const DevHuff *__restrict__ huff,
                                                       const uint16_t *__restrict__ lut11,   // [n_huff][kLSize]
                                                       int n_huff, int16_t *__restrict__ coef,
                                                       int32_t *__restrict__ status, int lpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *s_lut = reinterpret_cast<uint16_t *>(smem);                          // [n_huff][kLSize]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *s_blk = reinterpret_cast<uint32_t *>(smem + (size_t)n_huff * kLSize * 2) + wave * (64 * kBlkStride);
    uint8_t *s_nat = smem + (size_t)n_huff * kLSize * 2 + 4 * 64 * kBlkStride * 4;

    for (int i = tid; i < n_huff * kLSize / 8; i += 256)
        reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(lut11)[i];
    for (int i = tid; i < 4 * 64 * kBlkStride; i += 256)
        (reinterpret_cast<uint32_t *>(smem + (size_t)n_huff * kLSize * 2))[i] = 0;
    if (tid < 64) s_nat[tid] = c_nat_of_zz_l[tid];
    __syncthreads();

    const int64_t seg_id = ((int64_t)blockIdx.x * 4 + wave) * lpw + lane;   // lpw segments per wave (tunable)
    const bool have = lane < lpw && seg_id < n_segs;
    const DevSegment sg = segs[have ? seg_id : 0];
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
    const int n_mcu = have ? sg.n_mcu : 0;
    int max_mcu = n_mcu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_mcu = max(max_mcu, __shfl_xor(max_mcu, o));
    max_mcu = __builtin_amdgcn_readfirstlane(max_mcu);

    LaneBits br;
    br.init(blob, sg.begin, have ? sg.len : 0);
    // byte offset of this lane's first output block
    const int64_t out_off = (im->block_off + (int64_t)sg.mcu0 * bpm) * 128;
    const uint32_t out_lo = (uint32_t)out_off, out_hi = (uint32_t)(out_off >> 32);
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int err = 0;
    uint32_t *myblk = s_blk + lane * kBlkStride;
    int16_t *myblk16 = reinterpret_cast<int16_t *>(myblk);

    for (int m = 0; m < max_mcu; ++m) {
        const bool in_mcu = m < n_mcu;
        for (int b = 0; b < bpm; ++b) {
            const int comp = (int)((comp_pk_u >> (8 * b)) & 0xFF);          // wave-uniform
            const int dct = comp == 0 ? dcT[0] : (comp == 1 ? dcT[1] : dcT[2]);
            const int act_ = comp == 0 ? acT[0] : (comp == 1 ? acT[1] : acT[2]);
            bool act = in_mcu && err == 0;

            // ---- DC (:810-820)
            br.refill(act);
            {
                const uint32_t p16 = (uint32_t)(br.bb >> 48);
                int e = s_lut[dct * kLSize + (p16 >> (16 - kLBits))];
                int len = e >> 8, s = e & 0xFF;
                if (__any(act && len == 0)) {
                    if (act && len == 0) {
                        const DevHuff *t = huff + dct;
                        s = -1;
                        for (int l = kLBits + 1; l <= 16; ++l) {
                            const int d = (int)(p16 >> (16 - l)) - t->first_code[l];
                            if (d >= 0 && d < t->count[l]) { s = t->vals[t->first_sym[l] + d]; len = l; break; }
                        }
                    }
                }
                if (act && (s < 0 || s > 16)) { err = MJ_ST_BAD_CODE; act = false; }
                if (act) {
                    br.bb <<= len; br.bc -= len;
                    const int diff = extend(br.take(s), s);
                    const int pred = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
                    const int dcv = (int)(int16_t)(diff + pred);
                    if (comp == 0) pred0 = dcv; else if (comp == 1) pred1 = dcv; else pred2 = dcv;
                    myblk16[0] = (int16_t)dcv;
                }
            }
            // ---- AC (:833-866)
            int k = act ? 1 : 64;
            while (__any(k < 64)) {
                const bool on = k < 64;
                br.refill(on);
                const uint32_t p16 = (uint32_t)(br.bb >> 48);
                int e = s_lut[act_ * kLSize + (p16 >> (16 - kLBits))];
                int len = e >> 8, hv = e & 0xFF;
                if (__any(on && len == 0)) {
                    if (on && len == 0) {
                        const DevHuff *t = huff + act_;
                        hv = -1;
                        for (int l = kLBits + 1; l <= 16; ++l) {
                            const int d = (int)(p16 >> (16 - l)) - t->first_code[l];
                            if (d >= 0 && d < t->count[l]) { hv = t->vals[t->first_sym[l] + d]; len = l; break; }
                        }
                    }
                }
                if (on) {
                    if (hv < 0) { err = MJ_ST_BAD_CODE; k = 64; }
                    else {
                        br.bb <<= len; br.bc -= len;
                        if (hv == 0) k = 64;                             // EOB
                        else {
                            const int kk = k + (hv >> 4);
                            const int n = hv & 15;
                            if (kk >= 64) k = 64;                        // (:855-856) ends the block, bits of the value stay unread
                            else {
                                if (n > 0) {
                                    const int val = extend(br.take(n), n);
                                    myblk16[s_nat[kk]] = (int16_t)val;
                                }
                                k = kk + 1;
                            }
                        }
                    }
                }
            }
            // ---- round of 64 blocks done: LDS -> HBM, two full lines per instruction, and clear
            const uint64_t act_mask = __ballot(in_mcu);
            const uint32_t blk_byte = (uint32_t)(m * bpm + b) * 128u;    // same for every lane (same layout)
#pragma unroll 4
            for (int s2 = 0; s2 < 32; ++s2) {
                if (((act_mask >> (2 * s2)) & 3) == 0) continue;         // uniform
                const int o = 2 * s2 + (lane >> 5), dw = lane & 31;
                const uint32_t v = s_blk[o * kBlkStride + dw];
                s_blk[o * kBlkStride + dw] = 0;
                const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readlane((int)out_lo, 2 * s2), hi0 = (uint32_t)__builtin_amdgcn_readlane((int)out_hi, 2 * s2);
                const uint32_t lo1 = (uint32_t)__builtin_amdgcn_readlane((int)out_lo, 2 * s2 + 1), hi1 = (uint32_t)__builtin_amdgcn_readlane((int)out_hi, 2 * s2 + 1);
                const uint64_t base = (lane >> 5) ? (((uint64_t)hi1 << 32) | lo1) : (((uint64_t)hi0 << 32) | lo0);
                if ((act_mask >> o) & 1)
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(coef) + base + blk_byte + dw * 4) = v;
            }
        }
    }

    if (have) {
        if (!err) {
            const int unconsumed = br.rawn + br.left - (br.skipnext ? 1 : 0);
            if (br.pad > 0 && br.bc < br.pad) err = MJ_ST_OVERRUN;
            else if (!sg.last && (((br.bc - br.pad) >> 3) > 0 || unconsumed > 0)) err = MJ_ST_DESYNC;
        }
        if (err) atomicMax(status + sg.image, err);
    }
}

hipError_t launch_huffman_lanes(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                                const DevImage *images, const DevHuff *huff, const uint16_t *lut11, int n_huff,
                                int16_t *coef, int32_t *status) {
    if (n_segs == 0) return hipSuccess;
    static int lpw = 0;
    if (lpw == 0) { const char *e = getenv("MJ_LANES_PER_WAVE"); lpw = e ? atoi(e) : 64; if (lpw < 2 || lpw > 64 || (lpw & 1)) lpw = 64; }
    const int64_t blocks = (n_segs + 4 * lpw - 1) / (4 * lpw);
    const size_t lds = (size_t)n_huff * kLSize * 2 + (size_t)4 * 64 * kBlkStride * 4 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_huffman_lanes), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_huffman_lanes, dim3((unsigned)blocks), dim3(256), lds, stream, blob, segs, n_segs, images, huff,
                       lut11, n_huff, coef, status, lpw);
    return hipGetLastError();
}

}  // namespace mj
