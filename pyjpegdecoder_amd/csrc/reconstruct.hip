// Stage 2 — dequantise + 8x8 inverse DCT + chroma upsample + YCbCr->RGB, fused, on gfx950.
//
// Replaces, per MCU, jpeg_decoder.py:869 (undo_zigzag * quantization_table), InverseDCT.__call__
// (:1561-1573), the block placement (:875-879), ResizeGrid.__call__ (:1588-1626), the store (:889-891),
// the crop (:1373) and YCbCr_to_RGB (:1683-1700) / the greyscale clip (:1384-1386).
//
// Work unit: one MCU per wavefront pass (grid-stride).  HBM traffic is the algorithmic minimum: each
// coefficient block is read once (128 B, one line per wave) and each output pixel is
// written once; everything in between lives in registers and LDS.
//
// Bit-exactness:
//   * IDCT: out[x,y] = np.sum(block * T[x,y]) is NumPy's pairwise sum: eight running sums r[v] over
//     u = 0..7 (flat index u*8+v), then ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), float64, products and sums
//     rounded separately.  Lane (x*8+y) reproduces exactly that sequence, skipping terms whose
//     coefficient is zero (x + 0.0 == x, so the skipped sequence is bitwise the same).  The table T is
//     the reference's own 4096 doubles, transposed to [u*8+v][x*8+y] so a wave reads 512 contiguous bytes.
//   * upsample: round(sum(n_i*v_i)/15) with the taps captured from the reference; exact in integers
//     because a multiple of 1/15 is never a half-integer.
//   * colour: evaluated in integers — 1.402 c = 701c/500, 1.772 c = 443c/250, 0.34414 a + 0.71414 b =
//     (17207a + 35707b)/50000 — which equals the float64 result whenever the real value is not an exact
//     half-integer (distance to the nearest tie >= 2e-5 >> float64 error ~1e-11); exact ties (and
//     out-of-range chroma) take the reference's float64 expression verbatim, contraction disabled.
#include "mijpeg_internal.h"
#include "upsample_taps.h"

#pragma clang fp contract(off)

namespace mj {

namespace {

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// floor((t) / den) for |t| < bias*den
template <int DEN, int BIAS>
__device__ __forceinline__ int floordiv(int t) {
    return (int)((unsigned)(t + DEN * BIAS) / (unsigned)DEN) - BIAS;
}

// YCbCr_to_RGB (:1693-1700) exactly as written, float64, no contraction.
__device__ __noinline__ void ycc_to_rgb_f64(int Y, int Cb, int Cr, int &R, int &G, int &B) {
    double y = (double)Y, cb = (double)Cb - 128.0, cr = (double)Cr - 128.0;
    double r = y + 1.402 * cr;
    double g = (y - 0.34414 * cb) - 0.71414 * cr;
    double b = y + 1.772 * cb;
    r = fmin(fmax(r, 0.0), 255.0);
    g = fmin(fmax(g, 0.0), 255.0);
    b = fmin(fmax(b, 0.0), 255.0);
    R = (int)__builtin_rint(r);
    G = (int)__builtin_rint(g);
    B = (int)__builtin_rint(b);
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void ycc_to_rgb(int Y, int Cb, int Cr, int &R, int &G, int &B) {
    int cb = Cb - 128, cr = Cr - 128;
    bool slow = (unsigned)(cb + 2048) > 4096u || (unsigned)(cr + 2048) > 4096u;
    int tr = 701 * cr + 250;                       // round(701 cr / 500)
    int qr = floordiv<500, 8192>(tr);
    int tb = 443 * cb + 125;                       // round(443 cb / 250)
    int qb = floordiv<250, 8192>(tb);
    int tg = 17207 * cb + 35707 * cr + 25000;      // round((17207 cb + 35707 cr) / 50000), subtracted
    int qg = floordiv<50000, 4096>(tg);
    // exact half-integers: remainder 0 after adding the half
    slow |= (tr - qr * 500 == 0) | (tb - qb * 250 == 0) | (tg - qg * 50000 == 0);
    if (slow) {
        ycc_to_rgb_f64(Y, Cb, Cr, R, G, B);
        return;
    }
    // G = Y - g where g is rounded to nearest; round(Y - g_real) = Y - round_half_down(g_real) — no tie here,
    // so nearest(−g) = −nearest(g)
    R = clamp255(Y + qr);
    B = clamp255(Y + qb);
    int tgn = -(17207 * cb + 35707 * cr) + 25000;
    G = clamp255(Y + floordiv<50000, 4096>(tgn));
}

template <int HS, int VS, int NC>
struct Geo {
    static constexpr int NBY = HS * VS;
    static constexpr int NB = NC == 1 ? 1 : NBY + 2;
    static constexpr int MW = NC == 1 ? 8 : 8 * HS;
    static constexpr int MH = NC == 1 ? 8 : 8 * VS;
    static constexpr int NPIX = MW * MH;
    static constexpr int PPL = NPIX / 64;           // pixels per lane
    static constexpr bool SUB = NC == 3 && NBY > 1; // chroma needs upsampling
};

}  // namespace

template <int HS, int VS, int NC, int LAYOUT>
__global__ __launch_bounds__(256) void k_reconstruct(ReconArgs a) {
    using G = Geo<HS, VS, NC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *s_tt = reinterpret_cast<double *>(smem);                                  // 32 KiB
    uint32_t *s_taps = reinterpret_cast<uint32_t *>(smem + 64 * 64 * sizeof(double)); // NPIX words (or 64)
    int16_t *s_mcu_all = reinterpret_cast<int16_t *>(smem + 64 * 64 * sizeof(double) + 256 * sizeof(uint32_t));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    int16_t *s_mcu = s_mcu_all + wave * (G::NB * 64);

    for (int i = tid; i < 64 * 64; i += 256) s_tt[i] = a.idct_tt[i];
    if constexpr (G::SUB) {
        const uint32_t *taps = HS == 4 ? UP_TAPS2_32x8 : ((HS == 2 && VS == 2) ? UP_TAPS_16x16 : (HS == 2 ? UP_TAPS_16x8 : UP_TAPS_8x16));
        for (int i = tid; i < G::NPIX; i += 256) s_taps[i] = taps[i];
    }
    __syncthreads();

    // lane n = u*8+v holds block[u][v] (reference [x][y] order); blocks are stored [v][u] by stage 1
    const int src_of_lane = (lane & 7) * 8 + (lane >> 3);
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < a.total_mcus; g += n_waves) {
        // ---- which image / MCU (wave-uniform)
        int img;
        int m;
        if (a.uniform_geometry) {
            img = (int)(g / a.mcus_per_image);
            m = (int)(g - (int64_t)img * a.mcus_per_image);
        } else {
            int lo = 0, hi = a.n_images;            // largest img with mcu_prefix[img] <= g
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (a.mcu_prefix[mid] <= g) lo = mid; else hi = mid;
            }
            img = lo;
            m = (int)(g - a.mcu_prefix[img]);
        }
        const DevImage *im = a.images + img;
        const int W = im->width, H = im->height;
        const int mcu_y = m / im->mcu_count_h, mcu_x = m - mcu_y * im->mcu_count_h;
        const int64_t blk0 = im->block_off + (int64_t)m * G::NB;
        const int16_t *cp = a.coef + blk0 * 64 + src_of_lane;

        int craw[G::NB];
#pragma unroll
        for (int b = 0; b < G::NB; ++b) craw[b] = cp[b * 64];

#pragma unroll
        for (int b = 0; b < G::NB; ++b) {
            const int comp = (NC == 1 || b < G::NBY) ? 0 : b - G::NBY + 1;
            const int q = a.qt[im->qt_index[comp] * 64 + src_of_lane];
            const int dn = (int)(int16_t)(craw[b] * q);                  // int16 * int16 -> int16 (:869)
            const uint64_t mask = __ballot(dn != 0);

            double r[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                r[v] = 0.0;
                const uint32_t cm = (uint32_t)((mask >> v) & 0x0101010101010101ull ? 1 : 0);
                if (cm) {
                    double t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = s_tt[(u * 8 + v) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if ((mask >> (u * 8 + v)) & 1) {
                            const int c = __builtin_amdgcn_readlane(dn, u * 8 + v);
                            const double p = (double)c * t[u];
                            r[v] = r[v] + p;
                        }
                    }
                }
            }
            const double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            const int val = (int)(int16_t)((int)(int16_t)(int)__builtin_rint(s) + 128);   // :1573
            s_mcu[b * 64 + lane] = (int16_t)val;
            if (a.idct_out) a.idct_out[(blk0 + b) * 64 + lane] = (int16_t)val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- pixels of the MCU: PPL consecutive pixels per lane along the contiguous axis
        unsigned char bytes[G::PPL * NC];
        int px[G::PPL], py[G::PPL];
#pragma unroll
        for (int j = 0; j < G::PPL; ++j) {
            const int p = lane * G::PPL + j;
            int x, y;
            if (LAYOUT == MJ_LAYOUT_XMAJOR) { x = p / G::MH; y = p % G::MH; }
            else                            { y = p / G::MW; x = p % G::MW; }
            px[j] = x; py[j] = y;
            const int yblk = (NC == 1) ? 0 : (y >> 3) * HS + (x >> 3);
            const int Yv = s_mcu[yblk * 64 + (x & 7) * 8 + (y & 7)];
            int Cbv = 0, Crv = 0;
            if constexpr (NC == 3) {
                const int16_t *cbp = s_mcu + G::NBY * 64, *crp = cbp + 64;
                if constexpr (G::SUB) {
                    const uint32_t tp = s_taps[x * G::MH + y];
                    int sb = 0, sr = 0;
                    if constexpr (HS == 4) {                                    // 4:1:1: two taps over 31
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int idx = (tp >> (11 * t)) & 63, w = (tp >> (11 * t + 6)) & 31;
                            sb += w * cbp[idx];
                            sr += w * crp[idx];
                        }
                        Cbv = (int)(int16_t)floordiv<62, 65536>(2 * sb + 31);   // round(s/31), never a tie (31 is odd)
                        Crv = (int)(int16_t)floordiv<62, 65536>(2 * sr + 31);
                    } else {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const int idx = (tp >> (10 * t)) & 63, w = (tp >> (10 * t + 6)) & 15;
                            sb += w * cbp[idx];
                            sr += w * crp[idx];
                        }
                        Cbv = (int)(int16_t)floordiv<30, 65536>(2 * sb + 15);   // round(s/15), never a tie
                        Crv = (int)(int16_t)floordiv<30, 65536>(2 * sr + 15);
                    }
                } else {
                    Cbv = cbp[x * 8 + y];
                    Crv = crp[x * 8 + y];
                }
            }
            const int gx = mcu_x * G::MW + x, gy = mcu_y * G::MH + y;
            if (a.planes && gx < W && gy < H) {
                int16_t *pl = a.planes + (im->pix_off + (int64_t)gx * H + gy) * NC;
                pl[0] = (int16_t)Yv;
                if constexpr (NC == 3) { pl[1] = (int16_t)Cbv; pl[2] = (int16_t)Crv; }
            }
            if constexpr (NC == 3) {
                int R, Gc, B;
                ycc_to_rgb(Yv, Cbv, Crv, R, Gc, B);
                bytes[j * 3 + 0] = (unsigned char)R;
                bytes[j * 3 + 1] = (unsigned char)Gc;
                bytes[j * 3 + 2] = (unsigned char)B;
            } else {
                bytes[j] = (unsigned char)clamp255(Yv);
            }
        }
        __builtin_amdgcn_wave_barrier();   // all reads of s_mcu done before the next pass overwrites it

        // ---- store
        const int gx0 = mcu_x * G::MW + px[0], gy0 = mcu_y * G::MH + py[0];
        int64_t off0;
        bool full;
        if (LAYOUT == MJ_LAYOUT_XMAJOR) {
            off0 = im->rgb_off + ((int64_t)gx0 * H + gy0) * NC;
            full = gx0 < W && gy0 + G::PPL <= H;
        } else {
            off0 = im->rgb_off + ((int64_t)gy0 * W + gx0) * NC;
            full = gy0 < H && gx0 + G::PPL <= W;
        }
        constexpr int NBYTES = G::PPL * NC;
        unsigned char *dst = a.rgb + off0;
        if (NBYTES % 4 == 0 && full && ((uintptr_t)dst & 3) == 0) {
#pragma unroll
            for (int k = 0; k < NBYTES / 4; ++k) {
                uint32_t wv = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | ((uint32_t)bytes[4 * k + 3] << 24);
                reinterpret_cast<uint32_t *>(dst)[k] = wv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::PPL; ++j) {
                const int gx = mcu_x * G::MW + px[j], gy = mcu_y * G::MH + py[j];
                if (gx < W && gy < H) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) dst[j * NC + c] = bytes[j * NC + c];
                }
            }
        }
    }
}

template <int HS, int VS, int NC>
static hipError_t launch_t(hipStream_t stream, const ReconArgs &a) {
    using G = Geo<HS, VS, NC>;
    const size_t lds = 64 * 64 * sizeof(double) + 256 * sizeof(uint32_t) + (size_t)4 * G::NB * 64 * sizeof(int16_t);
    int64_t want = (a.total_mcus + 3) / 4;
    int64_t cap = 256 * 4;   // 256 CUs x 4 workgroups (LDS: ~36 KiB each)
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    if (a.layout == MJ_LAYOUT_XMAJOR)
        hipLaunchKernelGGL((k_reconstruct<HS, VS, NC, MJ_LAYOUT_XMAJOR>), dim3(blocks), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((k_reconstruct<HS, VS, NC, MJ_LAYOUT_ROWMAJOR>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// ---- any sampling factors 1..4 per component (three components): the layouts outside the common five.
// The reference upsamples every component whose MCU shape differs from the largest one (:882-883) through ResizeGrid
// (:1588-1626) — scipy's griddata on a Delaunay triangulation of the source grid.  On a regular grid each triangle is half of
// a unit cell, so output sample (x, y) of an (8 hmax x 8 vmax) MCU lies at source position (x (sw-1)/(dw-1), y (sh-1)/(dh-1))
// and takes the barycentric weights of the three corners of the half-cell it falls in; which diagonal cuts a cell is
// captured from the reference (UP_DIAG, tools/make_layout_goldens.py: the formula reproduces griddata's operator for all 84
// pairs of factors).  Weights are integers over D = (dw-1)(dh-1) (a factor 1 where the dimension does not change): D is odd,
// sum(n_i v_i) / D is never a half-integer, and the float64 result of the reference rounds the way the integers do.
// One MCU per wavefront pass, exact-order IDCT (the reference's summation order), integer colour: correct first, it is the
// path of rare files.  Luma may be the upsampled component (Y 1x1 under 2x2 chroma).
__device__ __forceinline__ int generic_sample(const int16_t *plane, int sw, int sh, int dw, int dh, int hi, int vi, int x, int y) {
    if (sw == dw && sh == dh) return plane[x * sh + y];
    const int dxx = sw == dw ? 1 : dw - 1, dyy = sh == dh ? 1 : dh - 1;
    int cx = x, rx = 0, cy = y, ry = 0;
    if (sw != dw) { const int ax = x * (sw - 1); cx = ax / dxx; rx = ax - cx * dxx; }
    if (sh != dh) { const int ay = y * (sh - 1); cy = ay / dyy; ry = ay - cy * dyy; }
    const int D = dxx * dyy, FX = rx * dyy, FY = ry * dxx;
    const int x1 = min(cx + 1, sw - 1), y1 = min(cy + 1, sh - 1);
    const int v00 = plane[cx * sh + cy], v10 = plane[x1 * sh + cy], v01 = plane[cx * sh + y1], v11 = plane[x1 * sh + y1];
    const bool anti = rx != 0 && ry != 0 && ((UP_DIAG[(hi - 1) * 4 + (vi - 1)][min(cx, sw - 2)] >> min(cy, sh - 2)) & 1u);
    int s;
    if (!anti) s = FX >= FY ? (D - FX) * v00 + (FX - FY) * v10 + FY * v11 : (D - FY) * v00 + (FY - FX) * v01 + FX * v11;
    else s = FX + FY <= D ? (D - FX - FY) * v00 + FX * v10 + FY * v01 : (FX + FY - D) * v11 + (D - FY) * v10 + (D - FX) * v01;
    // round(s / D), never a tie: floor((2s + D) / 2D) with a bias that keeps the dividend positive (|s| < 961 * 2^15)
    return (int)(int16_t)((int)((unsigned)(2 * s + D + 2 * D * 65536) / (unsigned)(2 * D)) - 65536);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_reconstruct_generic(ReconArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *s_tt = reinterpret_cast<double *>(smem);                                  // 32 KiB
    int16_t *s_mcu_all = reinterpret_cast<int16_t *>(smem + 64 * 64 * sizeof(double));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    int16_t *s_mcu = s_mcu_all + wave * (kMaxBlocksPerMcu * 64);                      // component planes, one after the other
    for (int i = tid; i < 64 * 64; i += 256) s_tt[i] = a.idct_tt[i];
    __syncthreads();
    const int src_of_lane = (lane & 7) * 8 + (lane >> 3);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < a.total_mcus; g += n_waves) {
        int img, m;
        if (a.uniform_geometry) {
            img = (int)(g / a.mcus_per_image);
            m = (int)(g - (int64_t)img * a.mcus_per_image);
        } else {
            int lo = 0, hi = a.n_images;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (a.mcu_prefix[mid] <= g) lo = mid; else hi = mid;
            }
            img = lo;
            m = (int)(g - a.mcu_prefix[img]);
        }
        const DevImage *im = a.images + img;
        const int W = im->width, H = im->height, bpm = im->blocks_per_mcu;
        const int MW = 8 * im->hmax, MH = 8 * im->vmax;
        const int mcu_y = m / im->mcu_count_h, mcu_x = m - mcu_y * im->mcu_count_h;
        const int64_t blk0 = im->block_off + (int64_t)m * bpm;
        __builtin_amdgcn_wave_barrier();   // the previous pass has read its planes
        for (int b = 0; b < bpm; ++b) {
            const int comp = im->blk_comp[b];
            const int q = a.qt[im->qt_index[comp] * 64 + src_of_lane];
            const int dn = (int)(int16_t)(a.coef[(blk0 + b) * 64 + src_of_lane] * q);      // int16 * int16 -> int16 (:869)
            const uint64_t mask = __ballot(dn != 0);
            double r[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) r[v] = 0.0;
#pragma unroll 1
            for (int u = 0; u < 8; ++u) {                // running sums r[v] over u in order (NumPy's pairwise sum)
                const uint32_t rowbits = (uint32_t)(mask >> (u * 8)) & 0xFFu;
                if (rowbits == 0) continue;
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    if ((rowbits >> v) & 1) {
                        const int c = __builtin_amdgcn_readlane(dn, u * 8 + v);
                        const double p = (double)c * s_tt[(u * 8 + v) * 64 + lane];
                        r[v] = r[v] + p;
                    }
                }
            }
            const double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            const int val = (int)(int16_t)((int)(int16_t)(int)__builtin_rint(s) + 128);   // :1573
            // block `rep` of its component sits at (8 bx, 8 by) of the component's MCU, block_y, block_x = divmod(rep, h) (:875)
            const int rep = b - im->comp_first[comp], hc = im->comp_h[comp], sh = 8 * im->comp_v[comp];
            const int by = rep / hc, bx = rep - by * hc;
            s_mcu[im->comp_first[comp] * 64 + (bx * 8 + (lane >> 3)) * sh + by * 8 + (lane & 7)] = (int16_t)val;
            if (a.idct_out) a.idct_out[(blk0 + b) * 64 + lane] = (int16_t)val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int npix = MW * MH;
        for (int p = lane; p < npix; p += 64) {
            int x, y;
            if (LAYOUT == MJ_LAYOUT_XMAJOR) { x = p / MH; y = p - x * MH; }
            else                            { y = p / MW; x = p - y * MW; }
            const int gx = mcu_x * MW + x, gy = mcu_y * MH + y;
            if (gx >= W || gy >= H) continue;
            int v3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v3[c] = generic_sample(s_mcu + im->comp_first[c] * 64, 8 * im->comp_h[c], 8 * im->comp_v[c], MW, MH, im->comp_h[c], im->comp_v[c], x, y);
            if (a.planes) {
                int16_t *pl = a.planes + (im->pix_off + (int64_t)gx * H + gy) * 3;
                pl[0] = (int16_t)v3[0]; pl[1] = (int16_t)v3[1]; pl[2] = (int16_t)v3[2];
            }
            int R, Gc, B;
            ycc_to_rgb(v3[0], v3[1], v3[2], R, Gc, B);
            unsigned char *dst = a.rgb + im->rgb_off + (LAYOUT == MJ_LAYOUT_XMAJOR ? ((int64_t)gx * H + gy) : ((int64_t)gy * W + gx)) * 3;
            dst[0] = (unsigned char)R; dst[1] = (unsigned char)Gc; dst[2] = (unsigned char)B;
        }
    }
}

hipError_t launch_reconstruct_generic(hipStream_t stream, const ReconArgs &a) {
    const size_t lds = 64 * 64 * sizeof(double) + (size_t)4 * kMaxBlocksPerMcu * 64 * sizeof(int16_t);
    const int64_t want = (a.total_mcus + 3) / 4, cap = 256 * 4;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    if (a.layout == MJ_LAYOUT_XMAJOR) hipLaunchKernelGGL((k_reconstruct_generic<MJ_LAYOUT_XMAJOR>), dim3(blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((k_reconstruct_generic<MJ_LAYOUT_ROWMAJOR>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_reconstruct(hipStream_t stream, const ReconArgs &a, int hmax, int vmax, int ncomp) {
    if (ncomp == 1) return launch_t<1, 1, 1>(stream, a);
    if (hmax == 1 && vmax == 1) return launch_t<1, 1, 3>(stream, a);
    if (hmax == 2 && vmax == 1) return launch_t<2, 1, 3>(stream, a);
    if (hmax == 1 && vmax == 2) return launch_t<1, 2, 3>(stream, a);
    if (hmax == 2 && vmax == 2) return launch_t<2, 2, 3>(stream, a);
    if (hmax == 4 && vmax == 1) return launch_t<4, 1, 3>(stream, a);      // 4:1:1 (this kernel only: no fast form)
    return hipErrorInvalidValue;
}


// ---- MJ_LAYOUT_PLANAR_XMAJOR / _ROWMAJOR: (.., 3) -> (3, ..) per image.  A thread moves four pixels: three dwords in,
// one dword to each plane (an image's pixel count need not be a multiple of four, nor its offset of four bytes: the
// ragged ends go byte by byte).
__global__ __launch_bounds__(256) void k_planes_from_interleaved(const DevImage *__restrict__ images, const uint8_t *__restrict__ src,
                                                                 uint8_t *__restrict__ dst) {
    const DevImage im = images[blockIdx.y];
    const int64_t n = (int64_t)im.width * im.height;
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;        // first of this thread's four pixels
    if (q >= n) return;
    const uint8_t *s = src + im.rgb_off + 3 * q;
    uint8_t *d = dst + im.rgb_off + q;
    if (q + 4 <= n && ((im.rgb_off | n) & 3) == 0) {
        const uint32_t a = reinterpret_cast<const uint32_t *>(s)[0], b = reinterpret_cast<const uint32_t *>(s)[1], c = reinterpret_cast<const uint32_t *>(s)[2];
        // bytes: a = R0 G0 B0 R1, b = G1 B1 R2 G2, c = B2 R3 G3 B3
        const uint32_t r = (a & 0xFFu) | ((a >> 24) << 8) | (((b >> 16) & 0xFFu) << 16) | (((c >> 8) & 0xFFu) << 24);
        const uint32_t g = ((a >> 8) & 0xFFu) | ((b & 0xFFu) << 8) | ((b >> 24) << 16) | (((c >> 16) & 0xFFu) << 24);
        const uint32_t bl = ((a >> 16) & 0xFFu) | (((b >> 8) & 0xFFu) << 8) | ((c & 0xFFu) << 16) | ((c >> 24) << 24);
        *reinterpret_cast<uint32_t *>(d) = r;
        *reinterpret_cast<uint32_t *>(d + n) = g;
        *reinterpret_cast<uint32_t *>(d + 2 * n) = bl;
    } else {
        for (int i = 0; i < 4 && q + i < n; ++i) {
            d[i] = s[3 * i]; d[n + i] = s[3 * i + 1]; d[2 * n + i] = s[3 * i + 2];
        }
    }
}

hipError_t launch_planes_from_interleaved(hipStream_t stream, const DevImage *images, int n_images, int64_t max_pixels,
                                          const uint8_t *interleaved, uint8_t *planar) {
    if (n_images == 0 || max_pixels == 0) return hipSuccess;
    for (int i0 = 0; i0 < n_images; i0 += 65535) {          // grid.y limit
        const int ny = n_images - i0 < 65535 ? n_images - i0 : 65535;
        hipLaunchKernelGGL(k_planes_from_interleaved, dim3((unsigned)((max_pixels + 1023) / 1024), (unsigned)ny), dim3(256), 0, stream,
                           images + i0, interleaved, planar);
    }
    return hipGetLastError();
}

}  // namespace mj
