/*
 * jpegenc.c — deterministic baseline-JPEG WRITER for synthetic test and benchmark inputs.
 *
 * This is input-generation tooling, not part of the decode path and not derived from the
 * reference (the reference has no encoder).  It writes ITU-T T.81 baseline (SOF0) files with
 * the Annex-K Huffman tables, libjpeg-style quality scaling of the Annex-K quantisation
 * tables, optional chroma subsampling and optional DRI/RSTn restart markers, so that tests
 * and bench.py can build the workloads SURVEY.md §8(d) names on a box that has no Pillow.
 *
 * Image content family: R = 128+100 sin(x/9+y/17), G = 128+100 cos(x/13),
 * B = 128+90 sin(y/7), plus N(0,12^2) noise from a seeded splitmix64 stream, clipped to u8.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC tools/jpegenc.c -o tools/libjpegenc.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "std_tables.h"

static const uint8_t ZIGZAG_NAT[64] = { /* zig-zag index -> natural (row-major v*8+u) index */
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

/* ---------------------------------------------------------------- RNG + synthetic content */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static float GAUSS_LUT[4096];
static int gauss_ready = 0;
static void gauss_init(void) {
    /* inverse-CDF table of N(0,1) at 4096 mid-points (Acklam's rational approximation) */
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                               3.754408661907416e+00};
    for (int i = 0; i < 4096; i++) {
        double p = (i + 0.5) / 4096.0, x;
        if (p < 0.02425) {
            double q = sqrt(-2 * log(p));
            x = (((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1);
        } else if (p <= 1 - 0.02425) {
            double q = p - 0.5, r = q * q;
            x = (((((a[0]*r+a[1])*r+a[2])*r+a[3])*r+a[4])*r+a[5])*q / (((((b[0]*r+b[1])*r+b[2])*r+b[3])*r+b[4])*r+1);
        } else {
            double q = sqrt(-2 * log(1 - p));
            x = -(((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1);
        }
        GAUSS_LUT[i] = (float)x;
    }
    gauss_ready = 1;
}

static inline uint8_t clip_u8(float v) {
    int i = (int)lrintf(v);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

/* rgb: row-major [H][W][3] */
void mjenc_synth_rgb(uint64_t seed, int W, int H, float noise_sigma, uint8_t *rgb) {
    if (!gauss_ready) gauss_init();
    uint64_t s = seed * 0x2545F4914F6CDD1Dull + 0x1234567ull;
    float *gx = (float *)malloc(sizeof(float) * (size_t)W);
    for (int x = 0; x < W; x++) gx[x] = 128.f + 100.f * cosf((float)x / 13.f);
    /* small per-image phase so that different seeds differ in the smooth part too */
    float ph = (float)(seed % 64) * 0.1f;
    for (int y = 0; y < H; y++) {
        float by = 128.f + 90.f * sinf((float)y / 7.f + ph);
        uint8_t *row = rgb + (size_t)y * W * 3;
        for (int x = 0; x < W; x++) {
            uint64_t r = splitmix64(&s);
            float n0 = GAUSS_LUT[r & 4095] * noise_sigma;
            float n1 = GAUSS_LUT[(r >> 12) & 4095] * noise_sigma;
            float n2 = GAUSS_LUT[(r >> 24) & 4095] * noise_sigma;
            row[3*x+0] = clip_u8(128.f + 100.f * sinf((float)x / 9.f + (float)y / 17.f + ph) + n0);
            row[3*x+1] = clip_u8(gx[x] + n1);
            row[3*x+2] = clip_u8(by + n2);
        }
    }
    free(gx);
}

/* ---------------------------------------------------------------- bit writer */
typedef struct {
    uint8_t *p, *end;
    uint64_t acc;
    int nbits;
    int overflow;
} BitW;

static inline void put_byte(BitW *w, uint8_t b) {
    if (w->p < w->end) *w->p++ = b; else w->overflow = 1;
}
static inline void put_bits(BitW *w, uint32_t code, int n) {
    w->acc = (w->acc << n) | (code & ((1u << n) - 1));
    w->nbits += n;
    while (w->nbits >= 8) {
        uint8_t b = (uint8_t)(w->acc >> (w->nbits - 8));
        put_byte(w, b);
        if (b == 0xFF) put_byte(w, 0x00);
        w->nbits -= 8;
    }
}
static inline void flush_bits(BitW *w) {
    if (w->nbits > 0) put_bits(w, (1u << (8 - w->nbits)) - 1, 8 - w->nbits); /* pad with 1s */
    w->acc = 0; w->nbits = 0;
}

typedef struct { uint16_t code[256]; uint8_t len[256]; } HuffEnc;
static void huff_build(HuffEnc *h, const uint8_t *bits, const uint8_t *vals) {
    memset(h, 0, sizeof(*h));
    unsigned code = 0; int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l-1]; i++) { h->code[vals[k]] = (uint16_t)code++; h->len[vals[k]] = (uint8_t)l; k++; }
        code <<= 1;
    }
}

static inline int bit_category(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }

static void encode_block(BitW *w, const int16_t *zz, int *pred, const HuffEnc *dc, const HuffEnc *ac) {
    int diff = zz[0] - *pred; *pred = zz[0];
    int s = bit_category(diff);
    put_bits(w, dc->code[s], dc->len[s]);
    if (s) put_bits(w, (uint32_t)(diff < 0 ? diff - 1 : diff), s);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) { run++; continue; }
        while (run > 15) { put_bits(w, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        s = bit_category(v);
        int sym = (run << 4) | s;
        put_bits(w, ac->code[sym], ac->len[sym]);
        put_bits(w, (uint32_t)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run) put_bits(w, ac->code[0], ac->len[0]);
}

/* ---------------------------------------------------------------- FDCT + quantisation */
static double FD[8][8];
static int fd_ready = 0;
static void fd_init(void) {
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            FD[u][x] = 0.5 * (u == 0 ? sqrt(0.5) : 1.0) * cos((2 * x + 1) * u * M_PI / 16.0);
    fd_ready = 1;
}
/* in: 8x8 samples row-major [y][x] (level-shifted); out: zig-zag quantised coefficients */
static void fdct_quant(const float *in, const uint16_t *qt_zz, int16_t *zz) {
    double tmp[8][8], out[8][8];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            double s = 0; for (int x = 0; x < 8; x++) s += FD[u][x] * in[y*8+x];
            tmp[y][u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            double s = 0; for (int y = 0; y < 8; y++) s += FD[v][y] * tmp[y][u];
            out[v][u] = s;
        }
    for (int k = 0; k < 64; k++) {
        int nat = ZIGZAG_NAT[k];
        double c = out[nat >> 3][nat & 7] / qt_zz[k];
        zz[k] = (int16_t)(c < 0 ? -floor(-c + 0.5) : floor(c + 0.5));
    }
}

static void scale_qt(const uint8_t *base, int quality, uint16_t *out) {
    if (quality < 1) quality = 1; if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; i++) {
        int v = (base[i] * scale + 50) / 100;
        out[i] = (uint16_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

static void put_marker_seg(BitW *w, uint8_t m, const uint8_t *data, int len) {
    put_byte(w, 0xFF); put_byte(w, m);
    put_byte(w, (uint8_t)((len + 2) >> 8)); put_byte(w, (uint8_t)((len + 2) & 255));
    for (int i = 0; i < len; i++) put_byte(w, data[i]);
}

/*
 * subsamp: 0 = 4:4:4, 1 = 4:2:2 (h2v1), 2 = 4:2:0 (h2v2), 3 = 4:4:0 (h1v2), 4 = greyscale, 6 = 4:1:1 (h4v1).
 * rgb: row-major [H][W][3].  Returns bytes written, or -1 on overflow / bad argument.
 */
long mjenc_encode_rgb(const uint8_t *rgb, int W, int H, int quality, int subsamp,
                      int restart_interval, uint8_t *out, size_t cap) {
    if (!fd_ready) fd_init();
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || subsamp < 0 || subsamp > 6) return -1;
    const int non_interleaved = subsamp == 5;     /* 4:4:4 with one scan per component */
    if (non_interleaved) subsamp = 0;
    int ncomp = subsamp == 4 ? 1 : 3;
    int hs = subsamp == 6 ? 4 : ((subsamp == 1 || subsamp == 2) ? 2 : 1);     /* 6 = 4:1:1 (h4v1) */
    int vs = (subsamp == 2 || subsamp == 3) ? 2 : 1;
    if (ncomp == 1) { hs = vs = 1; }
    int mcu_w = 8 * hs, mcu_h = 8 * vs;
    int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
    int PW = mcus_x * mcu_w, PH = mcus_y * mcu_h;

    /* planes, padded by edge replication */
    float *Y = (float *)malloc(sizeof(float) * (size_t)PW * PH);
    float *Cb = NULL, *Cr = NULL;
    int CW = PW / hs, CH = PH / vs;
    if (ncomp == 3) {
        Cb = (float *)malloc(sizeof(float) * (size_t)PW * PH);
        Cr = (float *)malloc(sizeof(float) * (size_t)PW * PH);
    }
    for (int y = 0; y < PH; y++) {
        int sy = y < H ? y : H - 1;
        for (int x = 0; x < PW; x++) {
            int sx = x < W ? x : W - 1;
            const uint8_t *p = rgb + ((size_t)sy * W + sx) * 3;
            float r = p[0], g = p[1], b = p[2];
            Y[(size_t)y * PW + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
            if (ncomp == 3) {
                Cb[(size_t)y * PW + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                Cr[(size_t)y * PW + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
            }
        }
    }
    if (ncomp == 3 && (hs > 1 || vs > 1)) { /* box-filter downsample in place (top-left region) */
        for (int y = 0; y < CH; y++)
            for (int x = 0; x < CW; x++) {
                float sb = 0, sr = 0;
                for (int dy = 0; dy < vs; dy++)
                    for (int dx = 0; dx < hs; dx++) {
                        size_t i = (size_t)(y * vs + dy) * PW + (x * hs + dx);
                        sb += Cb[i]; sr += Cr[i];
                    }
                /* writing index (y*CW+x) <= every read index of later iterations */
                Cb[(size_t)y * CW + x] = sb / (hs * vs);
                Cr[(size_t)y * CW + x] = sr / (hs * vs);
            }
    }

    uint16_t qt[2][64];
    scale_qt(STD_QT_LUMA_ZZ, quality, qt[0]);
    scale_qt(STD_QT_CHROMA_ZZ, quality, qt[1]);
    HuffEnc hdc[2], hac[2];
    huff_build(&hdc[0], STD_DC_LUMA_BITS, STD_DC_LUMA_VALS);
    huff_build(&hdc[1], STD_DC_CHROMA_BITS, STD_DC_CHROMA_VALS);
    huff_build(&hac[0], STD_AC_LUMA_BITS, STD_AC_LUMA_VALS);
    huff_build(&hac[1], STD_AC_CHROMA_BITS, STD_AC_CHROMA_VALS);

    BitW w = {out, out + cap, 0, 0, 0};
    put_byte(&w, 0xFF); put_byte(&w, 0xD8);
    { static const uint8_t jfif[] = {'J','F','I','F',0, 1,1, 0, 0,1, 0,1, 0,0}; put_marker_seg(&w, 0xE0, jfif, sizeof(jfif)); }
    for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) {
        uint8_t seg[65]; seg[0] = (uint8_t)t;
        for (int i = 0; i < 64; i++) seg[1 + i] = (uint8_t)qt[t][i];
        put_marker_seg(&w, 0xDB, seg, 65);
    }
    {
        uint8_t seg[6 + 9]; int n = 0;
        seg[n++] = 8; seg[n++] = (uint8_t)(H >> 8); seg[n++] = (uint8_t)H; seg[n++] = (uint8_t)(W >> 8); seg[n++] = (uint8_t)W;
        seg[n++] = (uint8_t)ncomp;
        seg[n++] = 1; seg[n++] = (uint8_t)((hs << 4) | vs); seg[n++] = 0;
        if (ncomp == 3) { seg[n++] = 2; seg[n++] = 0x11; seg[n++] = 1; seg[n++] = 3; seg[n++] = 0x11; seg[n++] = 1; }
        put_marker_seg(&w, 0xC0, seg, n);
    }
    {
        const uint8_t *B[4] = {STD_DC_LUMA_BITS, STD_AC_LUMA_BITS, STD_DC_CHROMA_BITS, STD_AC_CHROMA_BITS};
        const uint8_t *V[4] = {STD_DC_LUMA_VALS, STD_AC_LUMA_VALS, STD_DC_CHROMA_VALS, STD_AC_CHROMA_VALS};
        const uint8_t ID[4] = {0x00, 0x10, 0x01, 0x11};
        for (int t = 0; t < (ncomp == 3 ? 4 : 2); t++) {
            uint8_t seg[1 + 16 + 256]; int n = 0, cnt = 0;
            seg[n++] = ID[t];
            for (int i = 0; i < 16; i++) { seg[n++] = B[t][i]; cnt += B[t][i]; }
            for (int i = 0; i < cnt; i++) seg[n++] = V[t][i];
            put_marker_seg(&w, 0xC4, seg, n);
        }
    }
    if (restart_interval > 0) {
        uint8_t seg[2] = {(uint8_t)(restart_interval >> 8), (uint8_t)restart_interval};
        put_marker_seg(&w, 0xDD, seg, 2);
    }
    if (non_interleaved) {
        /* three scans, one component each: the scan's MCU is one 8x8 block, blocks in raster order */
        float *P[3] = {Y, Cb, Cr};
        float blk[64]; int16_t zz[64];
        for (int c = 0; c < 3; c++) {
            uint8_t seg[6]; int n = 0;
            seg[n++] = 1; seg[n++] = (uint8_t)(c + 1); seg[n++] = c == 0 ? 0x00 : 0x11;
            seg[n++] = 0; seg[n++] = 63; seg[n++] = 0;
            put_marker_seg(&w, 0xDA, seg, n);
            int pred1 = 0, total = mcus_x * mcus_y, rst = 0;
            for (int m = 0; m < total; m++) {
                int x0 = (m % mcus_x) * 8, y0 = (m / mcus_x) * 8;
                for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) blk[y*8+x] = P[c][(size_t)(y0 + y) * PW + x0 + x];
                fdct_quant(blk, qt[c == 0 ? 0 : 1], zz);
                encode_block(&w, zz, &pred1, &hdc[c == 0 ? 0 : 1], &hac[c == 0 ? 0 : 1]);
                if (restart_interval > 0 && (m + 1) % restart_interval == 0 && m + 1 != total) {
                    flush_bits(&w);
                    put_byte(&w, 0xFF); put_byte(&w, (uint8_t)(0xD0 + (rst & 7))); rst++;
                    pred1 = 0;
                }
            }
            flush_bits(&w);
        }
        put_byte(&w, 0xFF); put_byte(&w, 0xD9);
        free(Y); free(Cb); free(Cr);
        return w.overflow ? -1 : (long)(w.p - out);
    }
    {
        uint8_t seg[1 + 6 + 3]; int n = 0;
        seg[n++] = (uint8_t)ncomp;
        seg[n++] = 1; seg[n++] = 0x00;
        if (ncomp == 3) { seg[n++] = 2; seg[n++] = 0x11; seg[n++] = 3; seg[n++] = 0x11; }
        seg[n++] = 0; seg[n++] = 63; seg[n++] = 0;
        put_marker_seg(&w, 0xDA, seg, n);
    }

    int pred[3] = {0, 0, 0};
    int total = mcus_x * mcus_y, rst = 0;
    float blk[64]; int16_t zz[64];
    for (int m = 0; m < total; m++) {
        int my = m / mcus_x, mx = m % mcus_x;
        for (int by = 0; by < vs; by++)
            for (int bx = 0; bx < hs; bx++) {
                int x0 = mx * mcu_w + bx * 8, y0 = my * mcu_h + by * 8;
                for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) blk[y*8+x] = Y[(size_t)(y0 + y) * PW + x0 + x];
                fdct_quant(blk, qt[0], zz);
                encode_block(&w, zz, &pred[0], &hdc[0], &hac[0]);
            }
        if (ncomp == 3) {
            float *P[2] = {Cb, Cr};
            int stride = (hs > 1 || vs > 1) ? CW : PW;
            for (int c = 0; c < 2; c++) {
                int x0 = mx * 8, y0 = my * 8;
                for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) blk[y*8+x] = P[c][(size_t)(y0 + y) * stride + x0 + x];
                fdct_quant(blk, qt[1], zz);
                encode_block(&w, zz, &pred[1 + c], &hdc[1], &hac[1]);
            }
        }
        if (restart_interval > 0 && (m + 1) % restart_interval == 0 && m + 1 != total) {
            flush_bits(&w);
            put_byte(&w, 0xFF); put_byte(&w, (uint8_t)(0xD0 + (rst & 7))); rst++;
            pred[0] = pred[1] = pred[2] = 0;
        }
    }
    flush_bits(&w);
    put_byte(&w, 0xFF); put_byte(&w, 0xD9);
    free(Y); free(Cb); free(Cr);
    return w.overflow ? -1 : (long)(w.p - out);
}

/*
 * Batch: image i uses seed seed0+i.  Files are written back to back into `blob`; offsets[i]..offsets[i+1]
 * delimit file i.  `stride` = per-image scratch capacity (bytes) used while encoding in parallel.
 * Returns total bytes or -1.
 */
long mjenc_synth_batch(int n, uint64_t seed0, int W, int H, float noise_sigma, int quality, int subsamp,
                       int restart_interval, uint8_t *blob, size_t cap, uint64_t *offsets) {
    size_t stride = (size_t)W * H * 3 + 65536;
    long *sizes = (long *)malloc(sizeof(long) * (size_t)n);
    uint8_t **bufs = (uint8_t **)calloc((size_t)n, sizeof(uint8_t *));
    int bad = 0;
    if (!gauss_ready) gauss_init();
    if (!fd_ready) fd_init();
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        uint8_t *rgb = (uint8_t *)malloc((size_t)W * H * 3);
        uint8_t *tmp = (uint8_t *)malloc(stride);
        mjenc_synth_rgb(seed0 + (uint64_t)i, W, H, noise_sigma, rgb);
        long sz = mjenc_encode_rgb(rgb, W, H, quality, subsamp, restart_interval, tmp, stride);
        free(rgb);
        if (sz < 0) { free(tmp); tmp = NULL; }
        else tmp = (uint8_t *)realloc(tmp, (size_t)sz);
        sizes[i] = sz; bufs[i] = tmp;
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        offsets[i] = off;
        if (sizes[i] < 0 || off + (size_t)sizes[i] > cap) { bad = 1; break; }
        memcpy(blob + off, bufs[i], (size_t)sizes[i]);
        off += (size_t)sizes[i];
    }
    offsets[n] = off;
    for (int i = 0; i < n; i++) free(bufs[i]);
    free(bufs); free(sizes);
    return bad ? -1 : (long)off;
}

/* Heterogeneous batch (bench.py `mixed_content`): file i draws, from its seed, a JPEG quality in {50, 60, 75, 85, 90, 95},
 * two noise levels out of {0, 5, 12, 40, 80} and a split row: the rows above it carry the first noise level, those below
 * the second — so the MCU rows (= restart segments) of one file, and the files among themselves, differ several-fold
 * in bits.  Same outputs as mjenc_synth_batch. */
long mjenc_synth_mixed_batch(int n, uint64_t seed0, int W, int H, int subsamp, int restart_interval, uint8_t *blob, size_t cap,
                             uint64_t *offsets) {
    static const int kQ[6] = {50, 60, 75, 85, 90, 95};
    static const float kS[5] = {0.0f, 5.0f, 12.0f, 40.0f, 80.0f};
    size_t stride = (size_t)W * H * 3 + 65536;
    long *sizes = (long *)malloc(sizeof(long) * (size_t)n);
    uint8_t **bufs = (uint8_t **)calloc((size_t)n, sizeof(uint8_t *));
    int bad = 0;
    if (!gauss_ready) gauss_init();
    if (!fd_ready) fd_init();
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        uint64_t h = (seed0 + (uint64_t)i) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const int q = kQ[h % 6];
        const float sa = kS[(h >> 8) % 5], sb = kS[(h >> 16) % 5];
        const int split = (int)((h >> 24) % (uint64_t)(H + 1));
        uint8_t *rgb = (uint8_t *)malloc((size_t)W * H * 3), *rgb2 = (uint8_t *)malloc((size_t)W * H * 3);
        uint8_t *tmp = (uint8_t *)malloc(stride * 3);              /* quality 95 on sigma 80: well above 1 byte per pixel */
        mjenc_synth_rgb(seed0 + (uint64_t)i, W, H, sa, rgb);
        mjenc_synth_rgb(seed0 + (uint64_t)i, W, H, sb, rgb2);
        memcpy(rgb + (size_t)split * W * 3, rgb2 + (size_t)split * W * 3, (size_t)(H - split) * W * 3);
        long sz = mjenc_encode_rgb(rgb, W, H, q, subsamp, restart_interval, tmp, stride * 3);
        free(rgb); free(rgb2);
        if (sz < 0) { free(tmp); tmp = NULL; }
        else tmp = (uint8_t *)realloc(tmp, (size_t)sz);
        sizes[i] = sz; bufs[i] = tmp;
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        offsets[i] = off;
        if (sizes[i] < 0 || off + (size_t)sizes[i] > cap) { bad = 1; break; }
        memcpy(blob + off, bufs[i], (size_t)sizes[i]);
        off += (size_t)sizes[i];
    }
    offsets[n] = off;
    for (int i = 0; i < n; i++) free(bufs[i]);
    free(bufs); free(sizes);
    return bad ? -1 : (long)off;
}

