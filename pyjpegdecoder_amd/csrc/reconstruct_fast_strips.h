// Stage 2, fast form: the strip worker — dequantise + 8x8 inverse DCT + chroma upsample + YCbCr->RGB of one wavefront's
// strips — shared by the stage-2 kernel (reconstruct_fast.hip, whose header describes the levels and the phases) and by
// the fused kernel (fused.hip), whose consumer wavefronts run exactly this code on the blocks their workgroup's producer
// wavefronts have just decoded.  What differs between the two is where a wavefront's JOBS come from (a `Src`, below).
// jpeg_decoder.py:869-891 (dequantise, InverseDCT, ResizeGrid, MCU store), :1373-1386, :1683-1700 (crop, YCbCr_to_RGB).
#pragma once
#include "mijpeg_internal.h"
#include "upsample_taps.h"

#pragma clang fp contract(off)

namespace mj {
namespace rfast {

// experiment switch (make XFLAGS=-DMJ_X_NTLOAD): the coefficient rows are read once — as streaming loads they would leave L2 to the pixel stores
#ifdef MJ_X_NTLOAD
typedef uint32_t mj_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 coef_load_nt(const uint4 *p) {
    const mj_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const mj_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
#define MJ_COEF_LOAD(p) coef_load_nt(p)
#else
#define MJ_COEF_LOAD(p) (*(p))
#endif

// K[x][u] = 0.5*c(u)*cos((2x+1)*u*pi/16), x = 0..3: even columns u = 0,2,4,6 and odd columns u = 1,3,5,7
constexpr double kA = 0.35355339059327373;    // 0.5/sqrt(2)
constexpr double kC2 = 0.46193976625564337, kC6 = 0.19134171618254492;
constexpr double kC1 = 0.4903926402016152, kC3 = 0.4157348061512726, kC5 = 0.27778511650980114, kC7 = 0.09754516100806417;

// 8-point IDCT: even part by butterflies, odd part as a 4x4 product whose 16 entries are +-{c1,c3,c5,c7}
// (7 fp64 constants in SGPRs instead of 20: no SGPR spilling)
__device__ __forceinline__ void idct8(const double f[8], double t[8]) {
    const double p = kA * (f[0] + f[4]), q = kA * (f[0] - f[4]);
    const double r = __builtin_fma(kC6, f[6], kC2 * f[2]);
    const double s = __builtin_fma(-kC2, f[6], kC6 * f[2]);
    const double e0 = p + r, e3 = p - r, e1 = q + s, e2 = q - s;
    const double o0 = __builtin_fma(kC7, f[7], __builtin_fma(kC5, f[5], __builtin_fma(kC3, f[3], kC1 * f[1])));
    const double o1 = __builtin_fma(-kC5, f[7], __builtin_fma(-kC1, f[5], __builtin_fma(-kC7, f[3], kC3 * f[1])));
    const double o2 = __builtin_fma(kC3, f[7], __builtin_fma(kC7, f[5], __builtin_fma(-kC1, f[3], kC5 * f[1])));
    const double o3 = __builtin_fma(-kC1, f[7], __builtin_fma(kC3, f[5], __builtin_fma(-kC5, f[3], kC7 * f[1])));
    t[0] = e0 + o0; t[7] = e0 - o0;
    t[1] = e1 + o1; t[6] = e1 - o1;
    t[2] = e2 + o2; t[5] = e2 - o2;
    t[3] = e3 + o3; t[4] = e3 - o3;
}

// the same transform in fp32 (level 1).  Constants are literals of the instructions (a constant held in an SGPR
// would halve the issue rate of the instruction that reads it).
constexpr float fA = 0.35355339059327373f;
constexpr float fC2 = 0.46193976625564337f, fC6 = 0.19134171618254492f;
constexpr float fC1 = 0.4903926402016152f, fC3 = 0.4157348061512726f, fC5 = 0.27778511650980114f, fC7 = 0.09754516100806417f;
__device__ __forceinline__ void idct8f(const float f[8], float t[8]) {
    const float p = fA * (f[0] + f[4]), q = fA * (f[0] - f[4]);
    const float r = __builtin_fmaf(fC6, f[6], fC2 * f[2]);
    const float s = __builtin_fmaf(-fC2, f[6], fC6 * f[2]);
    const float e0 = p + r, e3 = p - r, e1 = q + s, e2 = q - s;
    const float o0 = __builtin_fmaf(fC7, f[7], __builtin_fmaf(fC5, f[5], __builtin_fmaf(fC3, f[3], fC1 * f[1])));
    const float o1 = __builtin_fmaf(-fC5, f[7], __builtin_fmaf(-fC1, f[5], __builtin_fmaf(-fC7, f[3], fC3 * f[1])));
    const float o2 = __builtin_fmaf(fC3, f[7], __builtin_fmaf(fC7, f[5], __builtin_fmaf(-fC1, f[3], fC5 * f[1])));
    const float o3 = __builtin_fmaf(-fC1, f[7], __builtin_fmaf(fC3, f[5], __builtin_fmaf(-fC5, f[3], fC7 * f[1])));
    t[0] = e0 + o0; t[7] = e0 - o0;
    t[1] = e1 + o1; t[6] = e1 - o1;
    t[2] = e2 + o2; t[5] = e2 - o2;
    t[3] = e3 + o3; t[4] = e3 - o3;
}
// level-1 acceptance: distance to a half-integer must exceed kTieA * A + kTie0.  3 * 2^-24 = 1.79e-7 is the proven
// bound; the margin covers its second-order terms and the reference's own float64 noise (< 1e-9).
constexpr float kTieA = 2.0e-7f, kTie0 = 1.0e-6f;
// sum over the 8 lanes of a group (lanes 8g..8g+7), result in every lane: quad xor 1, quad xor 2, half-row mirror
__device__ __forceinline__ float group_sum8(float a) {
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xF, 0xF, true));
    return a;
}

__device__ __forceinline__ int lo16(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int hi16(uint32_t w) { return (int)w >> 16; }
// int16 half of a packed pair -> float in one instruction (the sub-dword select and the sign extension ride on the convert)
__device__ __forceinline__ float cvt_lo16(uint32_t w) {
    float r;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float cvt_hi16(uint32_t w) {
    float r;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int deq(int c, uint32_t q) { return (int)(int16_t)__mul24(c, (int)q); }   // int16 wrap (:869)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Image descriptors are read-only for the whole launch and every wave reads them at wave-uniform addresses: through the
// constant address space they are scalar loads, which (unlike vector loads) do not queue behind the wave's outstanding
// pixel stores — a vector load at the top of the strip loop made every strip wait for the previous strip's stores.
typedef const DevImage __attribute__((address_space(4))) *ConstImage;
__device__ __forceinline__ ConstImage cimg(const DevImage *p) { return (ConstImage)(uintptr_t)p; }
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte store at a dword-aligned address
// two coefficients at once: the low 16 bits of a 16x16 product are exactly numpy's int16 * int16 wrap (:869)
__device__ __forceinline__ uint32_t deq2(uint32_t c2, uint32_t q2) {
    const u16x2 p = __builtin_bit_cast(u16x2, c2) * __builtin_bit_cast(u16x2, q2);   // v_pk_mul_lo_u16
    return __builtin_bit_cast(uint32_t, p);
}

// YCbCr_to_RGB (jpeg_decoder.py:1693-1700) exactly as written: float64, no contraction.
__device__ __forceinline__ uint32_t ycc_to_rgb_f64(int Y, int Cb, int Cr) {
    double y = (double)Y, cb = (double)Cb - 128.0, cr = (double)Cr - 128.0;
    double r = y + 1.402 * cr;
    double g = (y - 0.34414 * cb) - 0.71414 * cr;
    double b = y + 1.772 * cb;
    r = fmin(fmax(r, 0.0), 255.0);
    g = fmin(fmax(g, 0.0), 255.0);
    b = fmin(fmax(b, 0.0), 255.0);
    return (uint32_t)(int)__builtin_rint(r) | ((uint32_t)(int)__builtin_rint(g) << 8) | ((uint32_t)(int)__builtin_rint(b) << 16);
}

template <int HS, int VS, int NC>
struct FGeo {
    static constexpr int NBY = HS * VS;
    static constexpr int NB = NC == 1 ? 1 : NBY + 2;
    static constexpr int MW = NC == 1 ? 8 : 8 * HS;
    static constexpr int MH = NC == 1 ? 8 : 8 * VS;
    static constexpr int TML = 64 / MW;                  // MCUs the 64 lanes cover in the pixel phase (lane = pixel column x MCU)
    // MCUs of 8 pixel rows (greyscale; VS = 1 in the kernel's geometry: 4:2:2 and 4:1:1 in x-major output, 4:4:0 in row-major)
    // make short column runs — TML * 8 * NC bytes: 96 for 4:2:2, 48 for 4:1:1, 64 for greyscale, against 192 for 4:2:0 and 4:4:4 —,
    // whose ends share their 64-byte sectors with the strips above and below.  Such strips are SV times as tall (strip_sv,
    // mijpeg_internal.h): a lane does SV MCUs in the pixel phase, and the runs are 192 / 96 / 256 bytes (round 6)
    static constexpr int SV = strip_sv(MW, MH, NC);
    static constexpr int TMW = TML * SV;                 // MCUs per wave strip
    static constexpr int NBT = TMW * NB;                 // blocks of a strip
    static constexpr int ROUNDS = (NBT + 7) / 8;         // 8 blocks per round; 4:1:1's 12 blocks leave half of the second round idle
    static constexpr int MCU_STRIDE = NB * 64 + 32;      // int16 elements; +64 B makes phase B's 16-byte reads conflict-free
    static constexpr bool SUB = NC == 3 && NBY > 1;
    static constexpr int STRIP_BYTES = TMW * MCU_STRIDE * 2;
    // per-wave scratch: the transposes of phase A (8 groups x 576 B), then the wave's 64 pixel runs on their way out
    static constexpr int SCR_BYTES = 64 * MH * NC * SV > 8 * 576 ? 64 * MH * NC * SV : 8 * 576;
    static constexpr int QT_BYTES = 3 * 128;              // this wave's image's quantisation tables
    static constexpr int WAVE_BYTES = STRIP_BYTES + SCR_BYTES + QT_BYTES;
    static constexpr int WTS_ROW = MH + 1;                // float4 per row, padded against bank conflicts
    static constexpr int WTS_BYTES = SUB ? MW * WTS_ROW * 16 : 0;
    static constexpr int LDS_BYTES = 4 * WAVE_BYTES + WTS_BYTES;
    static_assert(WAVE_BYTES % 16 == 0, "16-byte LDS accesses");
};

// The captured ResizeGrid operators in one form for every geometry: an output sample (x, y) of the kernel's (possibly
// transposed) MCU interpolates between the corners (sx0 + dx, sy0 + dy) of one source cell with integer weights over
// kDen — 15 for the factor-2 layouts (three taps of a triangle), 31 for 4:1:1 (two taps along its one subsampled axis).
template <int HS, int VS> constexpr int kDen = (HS == 4 || VS == 4) ? 31 : 15;
template <int N> __device__ __forceinline__ constexpr int src_pos(int p) { return N > 1 ? (7 * p) / (8 * N - 1) : p; }
// weights w[dx][dy] packed one byte each: bits [8*(2*dx+dy), +8).  T = the kernel runs on the transposed image
// (row-major output): its (x, y) is the original's (y, x).
template <int HS, int VS, bool T>
__device__ __forceinline__ uint32_t corner_weights(int x, int y) {
    if constexpr (HS == 4 || VS == 4) {
        const int ox = T ? y : x, oy = T ? x : y;                     // sample of the original 32x8 MCU
        const uint32_t w = UP_TAPS2_32x8[ox * 8 + oy];
        const int i0 = w & 63, n0 = (w >> 6) & 31, i1 = (w >> 11) & 63, n1 = (w >> 17) & 31;
        const int sx0 = (7 * ox) / 31;                                // the taps lie on the source row oy, columns sx0 and sx0 + 1
        uint32_t out = 0;
        // corner (dx, dy) in kernel coordinates: the original's column step is dx (dy when transposed)
        out |= (uint32_t)n0 << (8 * ((i0 >> 3) == sx0 ? 0 : (T ? 1 : 2)));
        if (n1) out |= (uint32_t)n1 << (8 * ((i1 >> 3) == sx0 ? 0 : (T ? 1 : 2)));
        return out;
    } else {
        const uint16_t *w4 = T ? ((HS == 2 && VS == 2) ? UP_W4T_16x16 : (HS == 2 ? UP_W4T_16x8 : UP_W4T_8x16))
                               : ((HS == 2 && VS == 2) ? UP_W4_16x16 : (HS == 2 ? UP_W4_16x8 : UP_W4_8x16));
        const uint32_t w = w4[x * (8 * VS) + y];
        return (w & 15) | (((w >> 4) & 15) << 8) | (((w >> 8) & 15) << 16) | ((w >> 12) << 24);
    }
}

// Upsampled chroma of pixel (column px, row y) from the two source rows, integer form (exact):
// round(sum(n_i*v_i)/kDen) — jpeg_decoder.py:1624-1626 through the captured operator.
template <int HS, int VS, typename P>
__device__ __forceinline__ int upsample_int(P cp, int sx0, int sx1, int y, uint32_t w) {
    constexpr int D = kDen<HS, VS>;
    const int sy0 = src_pos<VS>(y);
    const int sy1 = sy0 < 7 ? sy0 + 1 : 7;
    const int w00 = w & 255, w01 = (w >> 8) & 255, w10 = (w >> 16) & 255, w11 = w >> 24;
    auto v = [&](int i) { return (int)(int16_t)(cp[i] + 128); };        // strip holds chroma without the level shift
    const int s = w00 * v(sx0 * 8 + sy0) + w01 * v(sx0 * 8 + sy1) + w10 * v(sx1 * 8 + sy0) + w11 * v(sx1 * 8 + sy1);
    return (int)(int16_t)((int)((unsigned)(2 * s + D + 2 * D * 65536) / (unsigned)(2 * D)) - 65536);
}

// Slow, always-exact version of one lane's pixel run (rare): integer upsample + float64 colour, straight from
// the LDS strip to global memory.  Also serves the seam outputs (planes) of the parity tests.
template <int HS, int VS, int NC, bool T>
__device__ __noinline__ void pixel_run_exact(const int16_t *mt, int px, unsigned char *dst, int nrows,
                                             int16_t *planes /* or null */, int planes_step /* int16 elements per row */) {
    using G = FGeo<HS, VS, NC>;
    const int sx0 = src_pos<HS>(px);
    const int sx1 = sx0 < 7 ? sx0 + 1 : 7;
#pragma unroll 1
    for (int y = 0; y < nrows; ++y) {
        // block order inside an MCU is the original image's (block_count = by*h + bx, jpeg_decoder.py:875)
        const int yb = NC == 1 ? 0 : (T ? (px >> 3) * VS + (y >> 3) : (y >> 3) * HS + (px >> 3));
        const int Yv = mt[yb * 64 + (px & 7) * 8 + (y & 7)];
        if constexpr (NC == 3) {
            int Cbv, Crv;
            if constexpr (G::SUB) {
                const uint32_t w = corner_weights<HS, VS, T>(px, y);
                Cbv = upsample_int<HS, VS>(mt + G::NBY * 64, sx0, sx1, y, w);
                Crv = upsample_int<HS, VS>(mt + (G::NBY + 1) * 64, sx0, sx1, y, w);
            } else {
                Cbv = (int)(int16_t)(mt[G::NBY * 64 + px * 8 + y] + 128);
                Crv = (int)(int16_t)(mt[(G::NBY + 1) * 64 + px * 8 + y] + 128);
            }
            if (planes) { int16_t *pl = planes + (int64_t)y * planes_step; pl[0] = (int16_t)Yv; pl[1] = (int16_t)Cbv; pl[2] = (int16_t)Crv; }
            const uint32_t p = ycc_to_rgb_f64(Yv, Cbv, Crv);
            dst[3 * y] = (unsigned char)p; dst[3 * y + 1] = (unsigned char)(p >> 8); dst[3 * y + 2] = (unsigned char)(p >> 16);
        } else {
            if (planes) planes[(int64_t)y * planes_step] = (int16_t)Yv;
            dst[y] = (unsigned char)clamp255(Yv);
        }
    }
}

// Green of some pixels of one lane's run again, where the fp32 quotient of the fast path may be one off or sits on a tie
// (|17207 cb + 35707 cr  mod 50000| within 1 of 25000): the reference's float64 expression for exactly those pixels, patched
// into the staged bytes before they leave LDS.  Everything it touches is LDS, addressed as such (a generic pointer would make
// these flat accesses, which count on both memory counters), so the patch never waits for the pixel stores in flight.
// Integer upsample, so nothing here depends on fp32.  `halves`: bit b = rows 8b..8b+7 need the check.
typedef const int16_t __attribute__((address_space(3))) *lds_ci16;
typedef const float __attribute__((address_space(3))) *lds_cf32;
typedef unsigned char __attribute__((address_space(3))) *lds_u8;
__device__ __forceinline__ uint32_t lds_off(const void *p) {
    return (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3))) *)p;
}
template <int HS, int VS, bool T>
__device__ __noinline__ void green_fix_lds(uint32_t mt_off, uint32_t w_off, uint32_t stag_off, int px, int halves) {
    using G = FGeo<HS, VS, 3>;
    lds_ci16 mt = (lds_ci16)(uintptr_t)mt_off;
    lds_cf32 wrow = (lds_cf32)(uintptr_t)w_off;
    lds_u8 stag = (lds_u8)(uintptr_t)stag_off;
    const int sx0 = src_pos<HS>(px);
    const int sx1 = sx0 < 7 ? sx0 + 1 : 7;
    constexpr float D = (float)kDen<HS, VS>;
#pragma unroll 1
    for (int y = 0; y < G::MH; ++y) {
        if (!((halves >> (y >> 3)) & 1)) continue;
        int Cbv, Crv;
        if constexpr (G::SUB) {
            const uint32_t w = (uint32_t)(int)(wrow[4 * y] * D + 0.5f) | ((uint32_t)(int)(wrow[4 * y + 1] * D + 0.5f) << 8) |
                               ((uint32_t)(int)(wrow[4 * y + 2] * D + 0.5f) << 16) | ((uint32_t)(int)(wrow[4 * y + 3] * D + 0.5f) << 24);
            Cbv = upsample_int<HS, VS>(mt + G::NBY * 64, sx0, sx1, y, w);
            Crv = upsample_int<HS, VS>(mt + (G::NBY + 1) * 64, sx0, sx1, y, w);
        } else {
            Cbv = (int)(int16_t)(mt[G::NBY * 64 + px * 8 + y] + 128);
            Crv = (int)(int16_t)(mt[(G::NBY + 1) * 64 + px * 8 + y] + 128);
        }
        const int n = 17207 * (Cbv - 128) + 35707 * (Crv - 128);          // |n| < 2^24 on this path (|c| < 250)
        const int q = (int)__builtin_rintf((float)n * 2e-5f);
        int r = n - 50000 * q;
        r = r < 0 ? -r : r;
        if (r >= 24999) {
            const int yb = T ? (px >> 3) * VS + (y >> 3) : (y >> 3) * HS + (px >> 3);
            const uint32_t p = ycc_to_rgb_f64(mt[yb * 64 + (px & 7) * 8 + (y & 7)], Cbv, Crv);
            stag[3 * y + 1] = (unsigned char)(p >> 8);
        }
    }
}

// Exact-order IDCT of one block by a whole wave (lane = x*8+y), result into the LDS strip.
static __device__ __noinline__ void block_exact(const int16_t *cblk, const uint16_t *qblk, const double *tt, int16_t *out_lds,
                                         int16_t *idct_out /* or null */, bool transposed, bool chroma) {
    const int lane = threadIdx.x & 63;
    const int u = lane >> 3, v = lane & 7;
    const int src = transposed ? u * 8 + v : v * 8 + u;       // blocks (and tables) are stored [u][v] for row-major plans
    const int dn = deq(cblk[src], qblk[src]);
    const uint64_t mask = __ballot(dn != 0);
    // r[v] accumulates u = 0..7 in order (NumPy pairwise sum); zero terms are skipped (x + 0.0 == x)
    double rsum[8];
#pragma unroll
    for (int vv = 0; vv < 8; ++vv) rsum[vv] = 0.0;
#pragma unroll 1
    for (int uu = 0; uu < 8; ++uu) {
        const uint32_t rowbits = (uint32_t)(mask >> (uu * 8)) & 0xFFu;
        if (rowbits == 0) continue;
#pragma unroll
        for (int vv = 0; vv < 8; ++vv) {
            if ((rowbits >> vv) & 1) {
                const int cc = __builtin_amdgcn_readlane(dn, uu * 8 + vv);
                const double p = (double)cc * tt[(uu * 8 + vv) * 64 + lane];
                rsum[vv] = rsum[vv] + p;
            }
        }
    }
    const double s = ((rsum[0] + rsum[1]) + (rsum[2] + rsum[3])) + ((rsum[4] + rsum[5]) + (rsum[6] + rsum[7]));
    const int raw = (int)(int16_t)(int)__builtin_rint(s), val = (int)(int16_t)(raw + 128);
    out_lds[transposed ? (lane & 7) * 8 + (lane >> 3) : lane] = (int16_t)(chroma ? raw : val);   // [x'][y'] = [y][x] when transposed
    if (idct_out) idct_out[lane] = (int16_t)val;
}

// ---- where a wavefront's jobs come from ----------------------------------------------------------------------------------
// A Src hands out TICKETS (draw() may be called a job ahead: its latency is never waited for; take() makes the drawn value
// wave-uniform), says which jobs a ticket stands for, and — gated sources — whether a job's coefficient blocks are in memory
// yet.  The stage-2 kernel's: one counter word in global memory for the whole launch, `jpt` consecutive jobs per ticket.
struct TicketSource {
    static constexpr bool kSingleJobs = false;     // a ticket = `jpt` consecutive jobs
    uint32_t *counter;
    uint32_t jpt, n_tickets, last_ticket;
    int lane;
    __device__ __forceinline__ uint32_t draw() const {      // lane 0's ticket (other lanes 0); wave-uniform only after take()
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    }
    __device__ __forceinline__ uint32_t take(uint32_t t) const {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        // every wave draws exactly one ticket past the last job; the launch's very last ticket finds every other drawn
        if (c == last_ticket && lane == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return c;
    }
    __device__ __forceinline__ uint32_t first_job(uint32_t ticket) const { return ticket * jpt; }
    __device__ __forceinline__ uint32_t end_job(uint32_t ticket, uint32_t n_jobs) const { return min(n_jobs, (ticket + 1) * jpt); }
    __device__ __forceinline__ bool ready(uint32_t) const { return true; }
    __device__ __forceinline__ bool wait_ready(uint32_t) const { return true; }      // (false: the job was given up and the wave ends)
};

// One wavefront's share of a launch.  `wave_lds`: FGeo::WAVE_BYTES of LDS of its own (strip, scratch, quantisation tables);
// `s_wts`: the workgroup's corner-weight table (FGeo::WTS_BYTES, filled by fill_weights() before); `dump_slot`: which dump
// line the wave's workgroup uses; `wave_in_wg`: diagnostics only.
template <int HS, int VS, int NC, bool T>
__device__ __forceinline__ void fill_weights(float4 *wt, int tid, int nthreads) {
    using G = FGeo<HS, VS, NC>;
    if constexpr (G::SUB) {             // four corner weights / kDen as floats, [x][y]
        constexpr float D = (float)kDen<HS, VS>;
        for (int i = tid; i < G::MW * G::MH; i += nthreads) {
            const uint32_t w = corner_weights<HS, VS, T>(i / G::MH, i % G::MH);
            wt[(i / G::MH) * G::WTS_ROW + (i % G::MH)] = make_float4((float)(w & 255) / D, (float)((w >> 8) & 255) / D, (float)((w >> 16) & 255) / D,
                                                                   (float)(w >> 24) / D);
        }
    }
}

template <int HS, int VS, int NC, bool SEAMS, bool T, class Src>
__device__ __forceinline__ void strips_worker(const ReconArgs &a, const int64_t *__restrict__ job_prefix, const int64_t total_jobs, const int jobs_per_image,
                                              unsigned char *wave_lds, const float4 *s_wts, const int lane, const int dump_slot, const int wave_in_wg, Src &src) {
    using G = FGeo<HS, VS, NC>;
    int16_t *s_strip = reinterpret_cast<int16_t *>(wave_lds);
    double *scr = reinterpret_cast<double *>(wave_lds + G::STRIP_BYTES) + (lane >> 3) * 72;      // level 2
    // level 1's 8x8 transpose: element (x, v) of group g at float g*124 + x*16 + v — the ds_read_b128 of lane x is
    // conflict-free over the instruction's 16-lane sets, the eight ds_write_b32 are 2-way (free)
    float *scrf = reinterpret_cast<float *>(wave_lds + G::STRIP_BYTES) + (lane >> 3) * 124;
    uint16_t *s_qt = reinterpret_cast<uint16_t *>(wave_lds + G::STRIP_BYTES + G::SCR_BYTES);
    const int grp = lane >> 3, j = lane & 7;

    // phase-B identity of this lane: column px of MCU pk0 (+ TML per turn) of the strip
    const int px = lane / G::TML, pk0 = lane % G::TML;

    // A strip = TMW vertically adjacent MCUs of one MCU column (strips never wrap to the next column, so a
    // lane's MCU row is strip*TMW + k and every index below is either wave-uniform or a 24-bit multiply).
    // A JOB = up to a.chunk_strips vertically consecutive strips of one MCU column of one image: the unit the launch hands
    // out (one ticket, one wavefront).  Jobs are numbered image by image, column by column, piece by piece.
    struct Job {                         // all wave-uniform
        const DevImage *im;
        const int16_t *cfirst;           // coefficients of the job's first MCU row, this MCU column
        int row_elems;                   // int16 elements from one MCU row to the next
        int mcu_x, y_first, n_strips, mcv;
    };
    auto job_of = [&](uint32_t jb) -> Job {
        uint32_t img, r;
        if (a.uniform_geometry) {
            img = jb / (uint32_t)jobs_per_image;
            r = jb - img * (uint32_t)jobs_per_image;
        } else {
            int lo = 0, hi = a.n_images;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (job_prefix[mid] <= (int64_t)jb) lo = mid; else hi = mid;
            }
            img = (uint32_t)lo;
            r = jb - (uint32_t)job_prefix[lo];
        }
        img = (uint32_t)__builtin_amdgcn_readfirstlane((int)img);
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        Job jo;
        jo.im = a.images + img;
        // MCU grid of the (possibly transposed) image: mch columns, mcv rows
        const int mch = T ? cimg(jo.im)->mcu_count_v : cimg(jo.im)->mcu_count_h, mcv = T ? cimg(jo.im)->mcu_count_h : cimg(jo.im)->mcu_count_v;
        const uint32_t spc = (uint32_t)(mcv + G::TMW - 1) / G::TMW;     // strips per MCU column
        const uint32_t S = (uint32_t)a.chunk_strips, pieces = (spc + S - 1) / S;
        jo.mcv = mcv;
        jo.mcu_x = __builtin_amdgcn_readfirstlane((int)(r / pieces));
        const uint32_t s0 = (r - (uint32_t)jo.mcu_x * pieces) * S;      // first strip of the piece within its column
        jo.n_strips = (int)min(S, spc - s0);
        jo.y_first = (int)s0 * G::TMW;
        // coefficient blocks are in the ORIGINAL image's MCU raster: stepping down the strip moves one MCU row of the
        // original (or, transposed, one MCU to the right)
        if constexpr (T) {
            jo.row_elems = G::NB * 64;
            jo.cfirst = a.coef + (cimg(jo.im)->block_off + (int64_t)jo.mcu_x * mcv * G::NB) * 64 + (int64_t)jo.y_first * jo.row_elems;
        } else {
            jo.row_elems = mch * G::NB * 64;
            jo.cfirst = a.coef + (cimg(jo.im)->block_off + (int64_t)jo.mcu_x * G::NB) * 64 + (int64_t)jo.y_first * jo.row_elems;
        }
        return jo;
    };
    // all rounds' coefficient rows of a job's first strip: ROUNDS x 16 B per lane
    auto fetch_first = [&](const Job &jo, uint4 (&cw)[G::ROUNDS]) {
        const int nv = min(G::TMW, jo.mcv - jo.y_first);
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) {
            const int bt = min(r * 8 + grp, G::NBT - 1);        // (groups past the strip's last block repeat it: never stored)
            const int k = bt / G::NB, b = bt - k * G::NB;
#ifdef MJ_X_SPARSE_LD   // probe (wrong pixels): rows 4..7 of every chroma block are not fetched — what a half-block store could save on the read side
            if (NC == 3 && b >= G::NBY && j >= 4) cw[r] = MJ_COEF_LOAD(reinterpret_cast<const uint4 *>(a.dump + kDumpZeroLine));
            else
#endif
            cw[r] = MJ_COEF_LOAD(reinterpret_cast<const uint4 *>(jo.cfirst + __mul24(k < nv ? k : 0, jo.row_elems) + b * 64 + j * 8));
            asm volatile("" ::: "memory");     // keep the loads in round order: the waits in front of the rounds count on it
        }
    };

#ifdef MJ_DIAGNOSTIC   // clock probe / phase ablations: separate diagnostic build only (make DIAG=1), never in the product
    const int dm = a.debug_mask;
    if (dm & 1) for (int i = lane; i < G::STRIP_BYTES / 2; i += 64) s_strip[i] = 0;
    if (a.debug == 12 && wave_in_wg >= 2) { for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(127); }     // waves 2, 3 start ~4 us (half a strip) late
    if (a.debug == 13 && (wave_in_wg & 1)) { for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(127); }    // odd waves instead
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t dbg_wait = 0, dbg_acc[6] = {0, 0, 0, 0, 0, 0}, dbg_last = dbg_t0;
#define MJ_STAMP(i) do { if (a.debug == 10) { uint64_t s_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s_) :: "memory"); dbg_acc[i] += s_ - dbg_last; dbg_last = s_; } } while (0)
#endif
    // Jobs are handed out dynamically, one per ticket and wavefront.  Why not a fixed share per wave: the four workgroups that
    // share a CU are not served alike — the issue arbiter prefers the OLDEST wave of a SIMD, so with equal shares the first
    // workgroup of a CU finished at 60 % of the launch and the last one ran its final fifth alone on the CU, one wave per
    // SIMD (wave end times 2.7 .. 4.6 ms, mean 3.7: profiles/r04a_stage2_wave_end_times.txt).  A job's strips are vertically
    // adjacent and go through one wave one after the other, so the column runs that share a 64-byte sector are written by one
    // CU microseconds apart and merge in its XCD's L2.  The next job's ticket is drawn a job ahead (the atomic's latency is
    // never waited for); one counter word takes ~90 tickets per microsecond, a 1080p job (17 strips) asks for 30.
    // (a ticket is `jobs_per_ticket` consecutive jobs: layouts with little work per job — greyscale: 8 blocks per strip — would
    // otherwise ask the counter for more tickets than it serves, 88 per microsecond)
    const uint32_t n_jobs = (uint32_t)total_jobs;
    uint32_t ticket = src.take(src.draw());
    if (ticket >= src.n_tickets) return;
    uint32_t job = src.first_job(ticket);
    if (job >= n_jobs) return;            // (sources that hand out what is left: nothing was)
    const DevImage *qt_owner = nullptr;
    uint4 cw[G::ROUNDS];
  for (;;) {        // (gated sources only come round again: a job whose blocks were not there yet when its turn to be fetched came)
    if (!src.wait_ready(job)) return;     // (gated across workgroups: the job was given up to the clean-up launch, and so is the rest)
    fetch_first(job_of(job), cw);
    bool stalled = false;
    if constexpr (!SEAMS) {
        // as many stores behind the first fetch as every later fetch has behind it (see the store phase): otherwise the
        // loop entry is the path "no store after the loads" and the wait in front of phase A becomes vmcnt(0) for every strip
        constexpr int NT0 = (G::MW * G::TMW * G::MH * NC / 16 + 63) / 64;
        unsigned char *dump0 = a.dump + ((size_t)(dump_slot & 4095) * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < NT0; ++t) *reinterpret_cast<volatile u32x4_a4 *>(dump0) = u32x4_a4{0u, 0u, 0u, (uint32_t)t};
    }
    // Two loops.  Outer: the job — everything that takes divisions, descriptor loads or 64-bit products is done here, once
    // per job (17 strips for 1080p).  Inner: the job's strips, top to bottom — from one to the next the coefficient pointer,
    // the first MCU row and the output offset move by constants and the lanes' load offsets stay what they are.  (The kernel
    // is bound by instruction issue; the compiler's per-strip code for "which strip is next and where does it live" was a
    // quarter of the scalar and a tenth of the vector instructions.)
    for (;;) {                                      // tickets
      const uint32_t ticket_v = src.draw();     // the ticket after this one
      const uint32_t job_end = src.end_job(ticket, n_jobs);
      for (;;) {                                    // the ticket's jobs
        const Job jo = job_of(job);                 // (its first strip's coefficients are already on their way)
        const DevImage *im_g = jo.im;
        const ConstImage im = cimg(im_g);
        const int W = T ? im->height : im->width, H = T ? im->width : im->height;
        const int mch_o = im->mcu_count_h;              // MCUs per row of the ORIGINAL image (coefficient raster)
        const int mcv_k = jo.mcv;                        // MCU rows of the image the kernel sees
        const int mcu_x = jo.mcu_x;
        const int row_elems = jo.row_elems;
        const int64_t block_off = im->block_off;
        const uint16_t *qbase = a.qt;
        const int q0i = im->qt_index[0] * 64, q1i = im->qt_index[NC == 3 ? 1 : 0] * 64, q2i = im->qt_index[NC == 3 ? 2 : 0] * 64;
        if (im_g != qt_owner) {           // wave-uniform, rare: stage this image's tables (3 x 128 B) into the wave's LDS
            qt_owner = im_g;
            const int c = lane >> 4, part = lane & 15;     // lanes 0..47: 3 tables x 16 pieces of 8 bytes
            if (c < 3) {
                const int qi = c == 0 ? q0i : (c == 1 ? q1i : q2i);
                reinterpret_cast<uint2 *>(s_qt)[c * 16 + part] = reinterpret_cast<const uint2 *>(qbase + qi)[part];
            }
        }
        const uint32_t n_strips = (uint32_t)jo.n_strips;
        // this lane's coefficient rows relative to a strip's first MCU row: byte offsets (k * row + block b, row j), one per round
        uint32_t voff[G::ROUNDS];
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) {
            const int bt = min(r * 8 + grp, G::NBT - 1);
            const int k = bt / G::NB, b = bt - k * G::NB;
            voff[r] = (uint32_t)(__mul24(k, row_elems) + b * 64 + j * 8) * 2u;
        }
        const int16_t *cptr = jo.cfirst;            // wave-uniform: the current strip's first MCU row
        int y_first = jo.y_first;
        const int64_t rgb_off = im->rgb_off;
        const int hnc_i = H * NC;
        // first byte of this MCU column's pixel columns in the image
        unsigned char *const col_dst = a.rgb + rgb_off + (int64_t)(mcu_x * G::MW) * hnc_i;
      for (uint32_t si = 0;; ++si) {
        const int n_valid = min(G::TMW, mcv_k - y_first);
        // first block of strip MCU k in the coefficient store
        auto mcu_block = [&](int k) -> int64_t {
            return block_off + (T ? (int64_t)(mcu_x * mch_o + y_first + k) : (int64_t)((y_first + k) * mch_o + mcu_x)) * G::NB;
        };

#ifdef MJ_DIAGNOSTIC
        MJ_STAMP(5);          // loop head (and, first time round, everything before the loop)
        if (a.debug == 8 || a.debug == 9) {   // how long does this strip's prefetch (9: and the previous strip's stores) still take here?
            uint64_t s0, s1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0) :: "memory");
            if (a.debug == 8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1) :: "memory");
            dbg_wait += s1 - s0;
        }
#endif
        // ================= phase A: blocks ==================
        // One block's result row, packed, into the strip; t[] is what level 1 or 2 computed, ow the packed int16 row.
        auto store_row = [&](int k, int b, const uint4 &ow) {
            *reinterpret_cast<uint4 *>(s_strip + k * G::MCU_STRIDE + b * 64 + j * 8) = ow;
        };
        // ---- level 1: fp32, ROUNDS rounds of 8 blocks
        uint64_t susp_bits = 0;       // bit bt set = block bt of the strip needs the exact routine (level 3)
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) {
#ifdef MJ_DIAGNOSTIC
            if (a.debug == 1 || (dm & 1)) break;
#endif
            const int bt = min(r * 8 + grp, G::NBT - 1);
            const bool real = G::NBT % 8 == 0 || r * 8 + grp < G::NBT;       // (4:1:1: the last round is half empty)
            const int k = bt / G::NB, b = bt - k * G::NB;
            const int qc = (NC == 1 || b < G::NBY) ? 0 : b - G::NBY + 1;
            const int shift = qc == 0 ? 128 : 0;          // chroma stays centred in the strip
            const uint4 qw = *reinterpret_cast<const uint4 *>(s_qt + qc * 64 + j * 8);
            const uint32_t p0 = deq2(cw[r].x, qw.x), p1 = deq2(cw[r].y, qw.y), p2 = deq2(cw[r].z, qw.z), p3 = deq2(cw[r].w, qw.w);
            const uint32_t ac = (j == 0 ? (p0 & 0xFFFF0000u) : p0) | p1 | p2 | p3;   // any AC coefficient of this row
            const uint64_t acb = __ballot(ac != 0);
            float f[8], t[8];
            f[0] = cvt_lo16(p0); f[1] = cvt_hi16(p0); f[2] = cvt_lo16(p1); f[3] = cvt_hi16(p1);
            f[4] = cvt_lo16(p2); f[5] = cvt_hi16(p2); f[6] = cvt_lo16(p3); f[7] = cvt_hi16(p3);
            // A = sum of |coefficient| over the block: the scale of the fp32 error bound
            const float asum = group_sum8(((__builtin_fabsf(f[0]) + __builtin_fabsf(f[1])) + (__builtin_fabsf(f[2]) + __builtin_fabsf(f[3]))) +
                                          ((__builtin_fabsf(f[4]) + __builtin_fabsf(f[5])) + (__builtin_fabsf(f[6]) + __builtin_fabsf(f[7]))));
            idct8f(f, t);                                  // lane v: t[x] = sum_u K[x][u] B[u][v]
#pragma unroll
            for (int x = 0; x < 8; ++x) scrf[x * 16 + j] = t[x];
            {
                const float4 lo = *reinterpret_cast<const float4 *>(scrf + j * 16), hi = *reinterpret_cast<const float4 *>(scrf + j * 16 + 4);
                f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;   // lane x: row x
            }
            idct8f(f, t);                                  // lane x: t[y] = out[x][y]
            // round(t) + shift as the low 16 bits of  t + (1.5 * 2^23 + shift)  (one rounding, to an integer; the int16
            // wrap of :1573 comes with taking 16 bits); its distance from t is the distance to the nearest integer
            const float magic = 12582912.0f + (float)shift;
            uint32_t rb[8];
            float err = 0.0f;
#pragma unroll
            for (int y = 0; y < 8; y += 2) {
                const float r0 = t[y] + magic, r1 = t[y + 1] + magic;
                err = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t[y] - (r0 - magic)), __builtin_fabsf(t[y + 1] - (r1 - magic))), err);   // v_max3_f32
                rb[y] = __builtin_bit_cast(uint32_t, r0);
                rb[y + 1] = __builtin_bit_cast(uint32_t, r1);
            }
            uint4 ow;
            ow.x = __builtin_amdgcn_perm(rb[1], rb[0], 0x05040100u);
            ow.y = __builtin_amdgcn_perm(rb[3], rb[2], 0x05040100u);
            ow.z = __builtin_amdgcn_perm(rb[5], rb[4], 0x05040100u);
            ow.w = __builtin_amdgcn_perm(rb[7], rb[6], 0x05040100u);
            bool flagged = real && err >= (0.5f - kTie0) - kTieA * asum;
            if (acb != ~0ull) {                            // some group's block has no AC coefficient at all (wave-uniform test)
                // DC-only blocks need no sum: every sample is round(DC*q * T[0,0,0,0]) and T[0,0,0,0] is a hair above 1/8, so
                // the product rounds half AWAY from zero (equal to the reference over the whole int16 range)
                const bool dconly = ((acb >> (lane & 56)) & 0xFF) == 0;
                const int dc = __shfl(lo16(p0), lane & 56);
                const int sg = dc >> 31, ad = (dc ^ sg) - sg;
                const int vdc = ((((ad + 4) >> 3) ^ sg) - sg + shift) & 0xFFFF;
                const uint32_t vv = (uint32_t)vdc | ((uint32_t)vdc << 16);
                ow.x = dconly ? vv : ow.x; ow.y = dconly ? vv : ow.y; ow.z = dconly ? vv : ow.z; ow.w = dconly ? vv : ow.w;
                flagged = flagged && !dconly;
            }
            if (real) store_row(k, b, ow);
            if constexpr (SEAMS) {          // how the levels are used (mj_plan_idct_levels; seam-output kernels only: the tests' path)
                uint64_t fb = __ballot(flagged);
                fb |= fb >> 4; fb |= fb >> 2; fb |= fb >> 1; fb &= 0x0101010101010101ull;       // one bit per block with a flagged row
                const uint64_t rb8 = __ballot(real && j == 0);
                if (lane == 0) {
                    unsigned long long *cnt = a.level_counts;
                    atomicAdd(cnt, (unsigned long long)__builtin_popcountll(rb8));
                    if (fb) atomicAdd(cnt + 1, (unsigned long long)__builtin_popcountll(fb));
                }
            }
            // ---- level 2 (about one round in eight on noisy images): the groups whose block failed do it again in fp64
            if (__ballot(flagged) != 0) {
#ifdef MJ_DIAGNOSTIC
                if (a.debug == 5 || (dm & 16)) continue;                // timing only (wrong pixels): what level 2 costs
#endif
                int gf = flagged ? 1 : 0;                  // any lane of my group?
                gf |= __builtin_amdgcn_update_dpp(0, gf, 0xB1, 0xF, 0xF, true);
                gf |= __builtin_amdgcn_update_dpp(0, gf, 0x4E, 0xF, 0xF, true);
                gf |= __builtin_amdgcn_update_dpp(0, gf, 0x141, 0xF, 0xF, true);
                const bool mine = gf != 0;
                double fd[8], td[8];
                fd[0] = (double)lo16(p0); fd[1] = (double)hi16(p0); fd[2] = (double)lo16(p1); fd[3] = (double)hi16(p1);
                fd[4] = (double)lo16(p2); fd[5] = (double)hi16(p2); fd[6] = (double)lo16(p3); fd[7] = (double)hi16(p3);
                idct8(fd, td);
#pragma unroll
                for (int x = 0; x < 8; ++x) scr[x * 9 + j] = td[x];
#pragma unroll
                for (int v = 0; v < 8; ++v) fd[v] = scr[j * 9 + v];
                idct8(fd, td);
                int o[8];
                double errd = 0.0;
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const double rr = __builtin_rint(td[y]);
                    errd = fmax(errd, __builtin_fabs(td[y] - rr));
                    o[y] = (int)(int16_t)((int)(int16_t)(int)rr + shift);
                }
                const uint64_t sb = __ballot(mine && errd > (0.5 - 9.5367431640625e-07));
                if constexpr (SEAMS) {
                    const uint64_t s3 = __ballot(mine && j == 0 && ((sb >> (lane & 56)) & 0xFF) != 0);
                    if (lane == 0 && s3) atomicAdd(a.level_counts + 2, (unsigned long long)__builtin_popcountll(s3));
                }
                uint4 ow2;
                ow2.x = (uint32_t)(o[0] & 0xFFFF) | ((uint32_t)o[1] << 16);
                ow2.y = (uint32_t)(o[2] & 0xFFFF) | ((uint32_t)o[3] << 16);
                ow2.z = (uint32_t)(o[4] & 0xFFFF) | ((uint32_t)o[5] << 16);
                ow2.w = (uint32_t)(o[6] & 0xFFFF) | ((uint32_t)o[7] << 16);
                if (mine) store_row(k, b, ow2);
                if (sb != 0) {
#pragma unroll
                    for (int g8 = 0; g8 < 8; ++g8)
                        if ((sb >> (8 * g8)) & 0xFF) susp_bits |= 1ull << (r * 8 + g8);
                }
            }
        }
        susp_bits = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)susp_bits) |
                    ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(susp_bits >> 32)) << 32);
#ifdef MJ_DIAGNOSTIC
        MJ_STAMP(0);          // phase A rounds
#endif

        // ---- level 3, rare: blocks with a sample too close to a rounding boundary even in fp64 -> exact-order recompute
        while (susp_bits) {
            const int bt = __builtin_ctzll(susp_bits);
            susp_bits &= susp_bits - 1;
            const int k = bt / G::NB, b = bt - k * G::NB;
            if (k >= n_valid) continue;
            const int64_t blk = mcu_block(k) + b;
            const int qi = (NC == 1 || b < G::NBY) ? q0i : (b == G::NBY ? q1i : q2i);
            block_exact(a.coef + blk * 64, qbase + qi, a.idct_tt, s_strip + k * G::MCU_STRIDE + b * 64, nullptr, T, NC == 3 && b >= G::NBY);
        }
        if constexpr (SEAMS) {        // the :872 seam, from the finished strip (seam order is the original [x][y], level shift on every component)
            if (a.idct_out) {
#pragma unroll
                for (int r = 0; r < G::ROUNDS; ++r) {
                    const int bt = r * 8 + grp;
                    const int k = bt / G::NB, b = bt - k * G::NB;
                    const int back = (NC == 1 || b < G::NBY) ? 0 : 128;
                    if (bt < G::NBT && k < n_valid) {
                        const int16_t *row = s_strip + k * G::MCU_STRIDE + b * 64 + j * 8;
                        int16_t *io = a.idct_out + (mcu_block(k) + b) * 64;
#pragma unroll
                        for (int y = 0; y < 8; ++y) io[T ? y * 8 + j : j * 8 + y] = (int16_t)(row[y] + back);   // T: lane x' = original y
                    }
                }
            }
        }

        // the next strip's coefficient rows are requested now, into the registers phase A has just finished with;
        // they are consumed one iteration later, so HBM latency hides behind the pixel phase
#ifdef MJ_DIAGNOSTIC
        if (dm & 32) { if (si + 1 >= n_strips) { if (!Src::kSingleJobs && job + 1 < job_end) ++job; else { ticket = src.take(ticket_v); job = ticket < src.n_tickets ? src.first_job(ticket) : n_jobs; } } } else
#endif
        if (si + 1 < n_strips) {                     // the strip below: same column, TMW MCU rows further down
            const unsigned char *cn = reinterpret_cast<const unsigned char *>(cptr + (int64_t)G::TMW * row_elems);
            const int nv = min(G::TMW, mcv_k - (y_first + G::TMW));
            if (nv == G::TMW) {
#pragma unroll
                for (int r = 0; r < G::ROUNDS; ++r) {
#ifdef MJ_X_SPARSE_LD
                    const int bt_x = min(r * 8 + grp, G::NBT - 1), b_x = bt_x - (bt_x / G::NB) * G::NB;
                    const unsigned char *src_x = (NC == 3 && b_x >= G::NBY && j >= 4) ? a.dump + kDumpZeroLine : cn + voff[r];
                    cw[r] = MJ_COEF_LOAD(reinterpret_cast<const uint4 *>(src_x));
#else
                    cw[r] = MJ_COEF_LOAD(reinterpret_cast<const uint4 *>(cn + voff[r]));
#endif
                    asm volatile("" ::: "memory");
                }
            } else {                                 // the column's last strip has fewer MCUs: the missing ones repeat the first
#pragma unroll
                for (int r = 0; r < G::ROUNDS; ++r) {
                    const int bt = min(r * 8 + grp, G::NBT - 1);
                    const int k = bt / G::NB, b = bt - k * G::NB;
                    cw[r] = MJ_COEF_LOAD(reinterpret_cast<const uint4 *>(cn + (k < nv ? voff[r] : (uint32_t)(b * 64 + j * 8) * 2u)));
                    asm volatile("" ::: "memory");
                }
            }
        } else {                                     // the job ends: the first strip of this wave's next job
            // (a ticket of a source that hands out single jobs names ANY job — the next one's number may be below this one's)
            if (!Src::kSingleJobs && job + 1 < job_end) ++job;
            else { ticket = src.take(ticket_v); job = ticket < src.n_tickets ? src.first_job(ticket) : n_jobs; }
            if (job < n_jobs) {
                if (src.ready(job)) fetch_first(job_of(job), cw);
                else stalled = true;                 // (gated sources: its blocks are still being decoded — fetched at the top, after a wait)
            }
        }
#ifdef MJ_DIAGNOSTIC
        MJ_STAMP(1);          // level 3, next strip's geometry and fetch
#endif
        // ================= phase B: pixels ==================
        // A lane is one pixel column of one MCU (64 lanes = MW columns x TML MCUs); where the strip is SV times as tall as that
        // (MCUs of 8 pixel rows: FGeo), it does SV MCUs one after the other — MCU pk0 + sv * TML of the strip in turn sv — and
        // the store phase behind the turns moves the whole strip's runs.
        {
            constexpr int NBYTES = G::MH * NC;
            unsigned char *const s_out = wave_lds + G::STRIP_BYTES;
            uint32_t ob[(NBYTES + 3) / 4];
            constexpr int RGB = G::MH / 8;               // bits of a turn's `rg`: one per 8-row half of the MCU
            int act_m = 0, slow_m = 0, rg_m = 0;         // per turn: bit sv = lane active / slow; RGB bits from bit RGB * sv = rg
#pragma unroll
          for (int sv = 0; sv < G::SV; ++sv) {
            const int pk = pk0 + sv * G::TML;
#ifdef MJ_DIAGNOSTIC
            const bool have = pk < n_valid && a.debug != 2 && !(dm & 2);
#else
            const bool have = pk < n_valid;
#endif
            const int gx = mcu_x * G::MW + px, gy0 = (y_first + pk) * G::MH;
            const int16_t *mt = s_strip + pk * G::MCU_STRIDE;
            const int nrows = min(G::MH, H - gy0);
            unsigned char *dst = col_dst + (int64_t)px * hnc_i + gy0 * NC;

            if constexpr (SEAMS) {
                if (have && gx < W)
                    // planes are always the original x-major (W,H,C): transposed, this lane walks along the original x
                    pixel_run_exact<HS, VS, NC, T>(mt, px, dst, nrows,
                                                   a.planes ? a.planes + (im->pix_off + (T ? (int64_t)gy0 * W + gx : (int64_t)gx * H + gy0)) * NC : nullptr,
                                                   T ? W * NC : NC);
            } else {
#pragma unroll
                for (int i = 0; i < (NBYTES + 3) / 4; ++i) ob[i] = 0;
                bool slow = false;
                int rg = 0;
                if (have) {
                if constexpr (NC == 3) {
                    // chroma source rows sx0, sx0+1 of this lane's column, as floats
                    const int sx0 = src_pos<HS>(px);
                    const int sx1 = sx0 < 7 ? sx0 + 1 : 7;
                    // (Cb-128, Cr-128) pairs: the two chroma planes ride in the two halves of packed-fp32 registers, so
                    // one v_pk_* instruction serves both components.
                    // The packed int16 rows stay in 8 (16 with a second source row) registers; each half of the column
                    // converts only the source samples it touches, which keeps the live set under the 128-VGPR budget
                    const int16_t *cbp = mt + G::NBY * 64, *crp = cbp + 64;
                    const uint4 ba = *reinterpret_cast<const uint4 *>(cbp + sx0 * 8), ra = *reinterpret_cast<const uint4 *>(crp + sx0 * 8);
                    const uint32_t bw[4] = {ba.x, ba.y, ba.z, ba.w}, rw[4] = {ra.x, ra.y, ra.z, ra.w};
                    uint32_t bw2[4] = {0, 0, 0, 0}, rw2[4] = {0, 0, 0, 0};
                    if constexpr (HS > 1) {
                        const uint4 bb = *reinterpret_cast<const uint4 *>(cbp + sx1 * 8), rb = *reinterpret_cast<const uint4 *>(crp + sx1 * 8);
                        bw2[0] = bb.x; bw2[1] = bb.y; bw2[2] = bb.z; bw2[3] = bb.w;
                        rw2[0] = rb.x; rw2[1] = rb.y; rw2[2] = rb.z; rw2[3] = rb.w;
                    }
                    auto pairA = [&](int i) { return f32x2{(float)((i & 1) ? hi16(bw[i >> 1]) : lo16(bw[i >> 1])), (float)((i & 1) ? hi16(rw[i >> 1]) : lo16(rw[i >> 1]))}; };
                    auto pairB = [&](int i) { return f32x2{(float)((i & 1) ? hi16(bw2[i >> 1]) : lo16(bw2[i >> 1])), (float)((i & 1) ? hi16(rw2[i >> 1]) : lo16(rw2[i >> 1]))}; };
                    // largest |Cb-128|, |Cr-128| among the source samples (an upsampled value lies between its sources)
                    // and largest |remainder| of the green term: both decide, once per lane, whether fp32 was exact
                    float crange = 0.0f;
                    // |Cb - 128| = 125 somewhere in the run is the B tie (below).  An upsampled value lies between its sources, so
                    // it is ruled out once for the whole run on the packed int16 source samples: all of them inside (-125, 125),
                    // all above 125 or all below -125
                    bool tie125;
                    {
                        typedef short s16x2 __attribute__((ext_vector_type(2)));
                        auto pmax = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); };
                        auto pmin = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); };
                        uint32_t mx = pmax(pmax(bw[0], bw[1]), pmax(bw[2], bw[3])), mn = pmin(pmin(bw[0], bw[1]), pmin(bw[2], bw[3]));
                        if constexpr (HS > 1) {
                            mx = pmax(mx, pmax(pmax(bw2[0], bw2[1]), pmax(bw2[2], bw2[3])));
                            mn = pmin(mn, pmin(pmin(bw2[0], bw2[1]), pmin(bw2[2], bw2[3])));
                        }
                        const int hi_ = max(lo16(mx), hi16(mx)), lo_ = min(lo16(mn), hi16(mn));
                        tie125 = !((hi_ < 125 && lo_ > -125) || lo_ > 125 || hi_ < -125);
                    }
                    constexpr float MAGIC = 12582912.0f;                       // 1.5 * 2^23: x + MAGIC rounds x to an integer
#pragma unroll
                    for (int by = 0; by < G::MH / 8; ++by) {
                        const int yb = T ? (px >> 3) * VS + by : by * HS + (px >> 3);   // original block order (:875)
                        const uint4 yw = *reinterpret_cast<const uint4 *>(mt + yb * 64 + (px & 7) * 8);
                        const uint32_t ywd[4] = {yw.x, yw.y, yw.z, yw.w};
                        // source samples this half of the column interpolates between
                        const int s_lo = (G::SUB && VS > 1) ? src_pos<VS>(by * 8) : 0;
                        f32x2 cA[8], cB[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const bool used = !G::SUB || VS == 1 || (i >= s_lo && i <= src_pos<VS>(by * 8 + 7) + 1);
                            if (used) {
                                cA[i] = pairA(i);
                                crange = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cA[i].x), __builtin_fabsf(cA[i].y)), crange);
                                if constexpr (HS > 1) {
                                    cB[i] = pairB(i);
                                    crange = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cB[i].x), __builtin_fabsf(cB[i].y)), crange);
                                }
                            }
                        }
                        // (cb, cr) = (Cb - 128, Cr - 128) of pixel i of this half
                        auto chroma_of = [&](int i) -> f32x2 {
                            if constexpr (G::SUB) {
                                // sum(n_i v_i)/15 is never within 1/30 of a half-integer and the fp32 evaluation is
                                // within 0.012 of it for any int16 inputs, so rintf() returns the reference's value
                                const int y = by * 8 + i;
                                const int sy0 = src_pos<VS>(y);
                                const int sy1 = sy0 < 7 ? sy0 + 1 : 7;
                                const float4 wq = s_wts[px * G::WTS_ROW + y];
                                f32x2 sv = cA[sy0] * wq.x;
                                sv = __builtin_elementwise_fma(cA[sy1], f32x2{wq.y, wq.y}, sv);
                                if constexpr (HS > 1) {
                                    sv = __builtin_elementwise_fma(cB[sy0], f32x2{wq.z, wq.z}, sv);
                                    sv = __builtin_elementwise_fma(cB[sy1], f32x2{wq.w, wq.w}, sv);
                                }
                                return f32x2{__builtin_rintf(sv.x), __builtin_rintf(sv.y)};
                            } else {
                                return cA[i];
                            }
                        };
                        float remmax = 0.0f;                                       // largest |remainder| of the green term in this half
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int y = by * 8 + i;
                            // Y + MAGIC as a float, straight from the integer: MAGIC = 1.5 * 2^23 is 0x4B400000 and its
                            // neighbours within +-2^22 are its bit pattern plus the distance
                            const int Yi = (i & 1) ? hi16(ywd[i >> 1]) : lo16(ywd[i >> 1]);
                            const float Ym = __builtin_bit_cast(float, 0x4B400000 + Yi);
                            const f32x2 C = chroma_of(i);
                            // Colour (jpeg_decoder.py:1693-1700) in fp32 where that is exact.
                            // B, R: 1.772 cb = 443 cb / 250 and 1.402 cr = 701 cr / 500 hit an exact .5 first at |cb| = 125,
                            // |cr| = 250 and are otherwise >= 0.002 away from one, far more than the fp32 constants are off;
                            // one fma rounds  cb * 1.772 + (Y + MAGIC)  straight to  MAGIC + Y + round(1.772 cb).
                            const f32x2 br2 = __builtin_elementwise_fma(C, f32x2{1.772f, 1.402f}, f32x2{Ym, Ym}) - MAGIC;   // (B, R)
                            // G: N = 17207 cb + 35707 cr is an exact fp32 integer for |c| < 250; q = round(N / 50000) may
                            // be off by one only when the remainder is within 1.6 of +-25000, which also covers the ties.
                            // qm = MAGIC + q comes out of one fma; (Y + MAGIC) - (q + MAGIC) = Y - q exactly.
                            const f32x2 n2 = C * f32x2{17207.0f, 35707.0f};               // both products exact
                            const float N = n2.x + n2.y;                                  // exact: |N| < 2^24
                            const float qm = __builtin_fmaf(N, 2e-5f, MAGIC);
                            const float Gf = Ym - qm;
                            const float rem = __builtin_fmaf(-50000.0f, qm - MAGIC, N);
                            remmax = __builtin_fmaxf(__builtin_fabsf(rem), remmax);
                            const int o0 = 3 * y, o1 = 3 * y + 1, o2 = 3 * y + 2;
                            ob[o0 >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(br2.y, o0 & 3, ob[o0 >> 2]);
                            ob[o1 >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(Gf, o1 & 3, ob[o1 >> 2]);
                            ob[o2 >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(br2.x, o2 & 3, ob[o2 >> 2]);
                        }
                        rg |= remmax >= 24998.5f ? (1 << by) : 0;      // this half of the column has a pixel whose green must be redone
                    }
                    slow |= crange >= 250.0f || tie125;
                } else {
                    const uint4 yw = *reinterpret_cast<const uint4 *>(mt + (px & 7) * 8);
                    const uint32_t ywd[4] = {yw.x, yw.y, yw.z, yw.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float Yf = (float)((i & 1) ? hi16(ywd[i >> 1]) : lo16(ywd[i >> 1]));
                        ob[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(Yf, i & 3, ob[i >> 2]);
                    }
                }
                }   // have
                const bool active = have && gx < W;
                act_m |= active ? 1 << sv : 0;
                slow_m |= slow ? 1 << sv : 0;
                rg_m |= rg << (RGB * sv);
                // the lane's bytes of this MCU into the staging area, in run order: column px's run is its TMW MCUs back to back
                {
                    unsigned char *mine = s_out + (px * G::TMW + pk) * NBYTES;
                    if constexpr (NBYTES % 16 == 0) {
#pragma unroll
                        for (int i = 0; i < NBYTES / 16; ++i)
                            reinterpret_cast<uint4 *>(mine)[i] = make_uint4(ob[4 * i], ob[4 * i + 1], ob[4 * i + 2], ob[4 * i + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < NBYTES / 8; ++i) reinterpret_cast<uint2 *>(mine)[i] = make_uint2(ob[2 * i], ob[2 * i + 1]);
                    }
                }
            }
          }   // sv
            if constexpr (!SEAMS) {
#ifdef MJ_DIAGNOSTIC
                MJ_STAMP(2);      // pixel arithmetic
#endif
#ifdef MJ_DIAGNOSTIC
                const bool dbg_nostore = a.debug == 3 || (dm & 4);
                if (dbg_nostore) { uint32_t acc = 0;
#pragma unroll
                    for (int i = 0; i < (NBYTES + 3) / 4; ++i) acc ^= ob[i];
                    if (acc == 0x12345678u && slow_m) *(col_dst) = 1; }
#else
                constexpr bool dbg_nostore = false;
#endif
                if (!dbg_nostore) {
                // ---- stores.  A lane holds NBYTES consecutive bytes of one image column per MCU and the TMW MCUs of a column
                // are one contiguous run; written lane by lane, one store instruction would touch 64 separate
                // 16-byte pieces.  So the wave's bytes go through LDS (the transpose scratch is free now) and come back
                // as 16-byte pieces in run order: consecutive lanes write consecutive addresses.
                // The NT store instructions below are executed for EVERY strip, branch-free: pieces that must not land in
                // the image (columns past the right edge; all of them when the strip takes the per-lane stores further
                // down) go to this workgroup's dump line instead.  Why: vmcnt counts loads and stores together, in order,
                // and the compiler's s_waitcnt in front of the next strip's phase A allows as many younger operations as
                // the path with the FEWEST stores issues behind the coefficient prefetch — a branch around the stores
                // made that zero, and every strip waited for its predecessor's pixels to reach L2 (1.1 ms per launch).
                // bytes of a column run that lie inside the image (the bottom strip of an image whose height is no multiple of
                // the strip's: its last piece is stored as the 16 bytes that END at the image's edge — rewriting a few bytes of its
                // predecessor with the same values — so the store stays one 16-byte instruction per piece)
                constexpr int RUN = G::TMW * NBYTES, NPIECE = G::MW * RUN / 16;
                const int runv = min(RUN, (H - y_first * G::MH) * NC);
                const bool staged = runv >= 16 && (hnc_i & 3) == 0 && (rgb_off & 3) == 0 && __ballot((act_m & slow_m) != 0) == 0;
                {
                    static_assert(RUN % 16 == 0 && NBYTES % 8 == 0, "column runs are whole 16-byte pieces");
                    if constexpr (NC == 3) {
                        // green again where the fp32 quotient may be one off or sits on a tie (about 6 % of the strips have such a
                        // pixel in some lane): patched into the staged bytes, all in LDS
                        if (staged && __builtin_amdgcn_ballot_w64(act_m != 0 && rg_m != 0) != 0) {
#pragma unroll
                            for (int sv = 0; sv < G::SV; ++sv) {
                                const int pk = pk0 + sv * G::TML, rg = (rg_m >> (RGB * sv)) & ((1 << RGB) - 1);
                                if (((act_m >> sv) & 1) && rg != 0)
                                    green_fix_lds<HS, VS, T>(lds_off(s_strip + pk * G::MCU_STRIDE), lds_off(s_wts + px * G::WTS_ROW),
                                                             lds_off(s_out + (px * G::TMW + pk) * NBYTES), px, rg);
                            }
                        }
                    }
                    // Every lane stores one 16-byte piece per instruction, whatever the strip looks like.  Pieces that do not exist
                    // — columns past the right edge (columns are active or inactive as a whole: column c exists iff
                    // mcu_x*MW + c < W), bytes past the bottom edge, piece numbers past the last — become copies of an
                    // existing piece (the same bytes to the same address twice is harmless); a strip that takes the per-lane
                    // stores below sends its NT instructions to this workgroup's dump line instead.
                    const int ncol = min(G::MW, W - mcu_x * G::MW);
                    unsigned char *sbase = col_dst + y_first * (G::MH * NC);
                    // (the piece geometry is recomputed from the lane number every strip: as loop invariants the compiler
                    // keeps them in registers it does not have, and a spill reload is a vector-memory load that waits —
                    // vmcnt is in order — for the coefficient prefetch and for the previous piece's store)
                    int lane_o = lane;
                    asm volatile("" : "+v"(lane_o));
                    lane_o &= 63;
                    constexpr int NT = (NPIECE + 63) / 64;
                    uint4 pv[NT];
                    uint32_t doff[NT];                    // byte offset of piece t from sbase
                    const uint32_t hnc = (uint32_t)hnc_i;
                    if (runv == RUN && ncol == G::MW) {   // whole runs (wave-uniform; LDS reads may sit behind branches, stores may not)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {    // all reads first: one LDS round trip, not one per store
                            const int pce = (t * 64 + lane_o) % NPIECE;
                            const int c = (pce * 16) / RUN;
                            doff[t] = (uint32_t)c * hnc + (uint32_t)((pce * 16) - c * RUN);
                            pv[t] = *reinterpret_cast<const uint4 *>(s_out + 16 * pce);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int pce = (t * 64 + lane_o) % NPIECE;
                            const int c = min((pce * 16) / RUN, ncol - 1);
                            const int o = min((pce * 16) % RUN, runv - 16);       // dword aligned: runv is a multiple of 4 here
                            doff[t] = (uint32_t)c * hnc + (uint32_t)o;
                            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(s_out + c * RUN + o);
                            pv[t] = make_uint4(v.x, v.y, v.z, v.w);
                        }
                    }
                    unsigned char *stbase = staged ? sbase : a.dump + (size_t)(dump_slot & 4095) * 1024;     // wave-uniform
#ifdef MJ_DIAGNOSTIC
                    if (a.debug == 7 || (dm & 8)) stbase = a.dump + (size_t)(dump_slot & 4095) * 1024;    // timing only: every piece to the dump line
#endif
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        uint32_t off = staged ? doff[t] : (uint32_t)lane_o * 16u;
#ifdef MJ_DIAGNOSTIC
                        if (a.debug == 7 || (dm & 8)) off = (uint32_t)lane_o * 16u;
#endif
                        *reinterpret_cast<u32x4_a4 *>(stbase + off) = u32x4_a4{pv[t].x, pv[t].y, pv[t].z, pv[t].w};
                    }
                }
#ifdef MJ_DIAGNOSTIC
                MJ_STAMP(3);      // staging through LDS + the NT stores
#endif
                if (!staged && act_m != 0) {
                    // the per-lane way (rare: a lane that fp32 cannot decide, an image whose columns are not dword aligned): every
                    // MCU of the lane's turns from its staged bytes — or, where those are not final, through the exact routine
#pragma unroll
                    for (int sv = 0; sv < G::SV; ++sv) {
                        if (!((act_m >> sv) & 1)) continue;
                        const int pk = pk0 + sv * G::TML, gy0 = (y_first + pk) * G::MH, nrows = min(G::MH, H - gy0);
                        unsigned char *dst = col_dst + (int64_t)px * hnc_i + gy0 * NC;
                        const unsigned char *mine = s_out + (px * G::TMW + pk) * NBYTES;
                        if (((slow_m >> sv) & 1) || ((rg_m >> (RGB * sv)) & ((1 << RGB) - 1)) != 0 || nrows != G::MH || ((uintptr_t)dst & 3) != 0) {
                            pixel_run_exact<HS, VS, NC, T>(s_strip + pk * G::MCU_STRIDE, px, dst, nrows, nullptr, 0);
                        } else if (NBYTES % 16 == 0 && ((uintptr_t)dst & 15) == 0) {
#pragma unroll
                            for (int i = 0; i < NBYTES / 16; ++i) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(mine)[i];
                        } else if (NBYTES % 8 == 0 && ((uintptr_t)dst & 7) == 0) {
#pragma unroll
                            for (int i = 0; i < NBYTES / 8; ++i) reinterpret_cast<uint2 *>(dst)[i] = reinterpret_cast<const uint2 *>(mine)[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < NBYTES / 4; ++i) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(mine)[i];
                        }
                    }
                }
                }   // !dbg_nostore
            }
        }
#ifdef MJ_DIAGNOSTIC
        MJ_STAMP(4);              // slow-path pixels, green patches
#endif
        // the strip is private to this wave and LDS operations of one wave complete in order: no barrier
        if (si + 1 >= n_strips) break;
        y_first += G::TMW;
        cptr += (int64_t)G::TMW * row_elems;
      }
        if (Src::kSingleJobs || job >= job_end || stalled) break;       // (the next job belongs to the next ticket — or there is none)
      }
      if (job >= n_jobs || stalled) break;
    }
    if (job >= n_jobs) break;
  }
#ifdef MJ_DIAGNOSTIC
    if (a.debug == 10 && lane == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dump) + 393216 + 8;
        for (int i = 0; i < 6; ++i) atomicAdd(o + i, (unsigned long long)dbg_acc[i]);
    }
    if ((a.debug == 8 || a.debug == 9) && lane == 0) {   // sums over all waves: cycles in that wait, cycles in the kernel, waves
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dump) + 393216;
        atomicAdd(o, (unsigned long long)dbg_wait);
        atomicAdd(o + 1, (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_t0));
        atomicAdd(o + 2, 1ull);
    }
    if (a.debug == 11 && lane == 0) {    // when does every wave start and finish?  (100 MHz wall clock: balance of the persistent grid)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dump + (2u << 20)) + ((uint32_t)dump_slot * 4 + (uint32_t)wave_in_wg) * 4;
        o[0] = dbg_r0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |          // HW_REG_HW_ID
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);   // HW_REG_XCC_ID
        o[3] = 0;
    }
    if (a.debug == 4 && dump_slot == 7 && wave_in_wg == 0 && lane == 0) {   // diagnostic only: shader clock vs 100 MHz wall clock
        uint64_t *o = reinterpret_cast<uint64_t *>(a.rgb);
        o[0] = __builtin_amdgcn_s_memtime() - dbg_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
#endif
}

}  // namespace rfast
}  // namespace mj
