/*
 * jpeg_oracle.c — CPU restatement of the reference's per-MCU decode path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (pyjpegdecoder_amd/) never does.  Parity status: PINNED — every function below is checked
 * against vectors captured from the reference itself (tools/make_goldens.py imports
 * /root/reference/jpeg_decoder.py in the build container and writes the files under tests/golden), see
 * tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it restates (all into /root/reference/jpeg_decoder.py).
 * Array conventions are the reference's: 8x8 blocks are [x][y] = [horizontal][vertical] (F4 in
 * SURVEY.md), planes are x-major.
 *
 * Build (oracle/Makefile): gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC
 * Floating point must be IEEE double with no contraction: the IDCT's rounding at exact ties depends
 * on the precise sequence of roundings.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * undo_zigzag  (:1648-1662).  The literal matrix there is rows = vertical, then `.T`, so
 * out[x][y] = block[ZZ[y][x]] with ZZ the standard zig-zag index grid (:430-437).
 */
static const uint8_t ZZ_GRID[8][8] = {
    { 0,  1,  5,  6, 14, 15, 27, 28},
    { 2,  4,  7, 13, 16, 26, 29, 42},
    { 3,  8, 12, 17, 25, 30, 41, 43},
    { 9, 11, 18, 24, 31, 40, 44, 53},
    {10, 19, 23, 32, 39, 45, 52, 54},
    {20, 22, 33, 38, 46, 51, 55, 60},
    {21, 34, 37, 47, 50, 56, 59, 61},
    {35, 36, 48, 49, 57, 58, 62, 63}};

void orc_undo_zigzag(const int16_t *zz, int16_t *xy /* [8][8] as x*8+y */) {
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) xy[x * 8 + y] = zz[ZZ_GRID[y][x]];
}

/* ------------------------------------------------------------------------------------------------
 * InverseDCT.idct_table  (:1541-1553):
 *   T[x,y,u,v] = 0.25 * Cu * Cv * cos((2x+1)*pi*u/16) * cos((2y+1)*pi*v/16), evaluated left to right
 *   in Python floats (IEEE double), Cu = 2**-0.5 for u == 0 else 1.0.
 * Python's math.cos is libm cos(); `(2*x + 1) * pi * u / 16` is ((int*pi)*u)/16 in doubles.
 */
void orc_idct_table(double *T /* [8][8][8][8] */) {
    const double pi = 3.141592653589793; /* math.pi */
    const double isq2 = pow(2.0, -0.5);  /* 2**(-0.5) == 0.7071067811865476 */
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++)
            for (int u = 0; u < 8; u++)
                for (int v = 0; v < 8; v++) {
                    double Cu = u == 0 ? isq2 : 1.0, Cv = v == 0 ? isq2 : 1.0;
                    double a = ((double)(2 * x + 1) * pi) * (double)u / 16.0;
                    double b = ((double)(2 * y + 1) * pi) * (double)v / 16.0;
                    double t = 0.25 * Cu;
                    t = t * Cv;
                    t = t * cos(a);
                    t = t * cos(b);
                    T[((x * 8 + y) * 8 + u) * 8 + v] = t;
                }
}

/* ------------------------------------------------------------------------------------------------
 * Dequantise + IDCT of one block.
 *   :869   block = undo_zigzag(block) * quantization_table      (int16 * int16 -> int16, wraps)
 *   :1566-1573  output[x,y] = np.sum(block * T[x,y], dtype=float64); np.round(...).astype(int16) + 128
 * np.sum over the 64 contiguous doubles uses NumPy's pairwise summation kernel, which for n = 64
 * (< its 128-element block) is: eight running sums r[j] = a[j], then r[j] += a[i+j] for i = 8,16,..,56,
 * combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))  (pinned by tests/golden/idct_*.npz).
 * qt_xy is the reference's quantization_tables[id] ([x][y], :454-462).
 */
void orc_dequant_idct(const int16_t *zz, const int16_t *qt_xy, const double *T,
                      int16_t *deq_xy /* may be NULL */, int16_t *out_xy) {
    int16_t blk[64];
    orc_undo_zigzag(zz, blk);
    for (int k = 0; k < 64; k++) blk[k] = (int16_t)((int32_t)blk[k] * (int32_t)qt_xy[k]);
    if (deq_xy) memcpy(deq_xy, blk, sizeof(blk));
    for (int xy = 0; xy < 64; xy++) {
        const double *t = T + xy * 64;
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = (double)blk[j] * t[j];
        for (int i = 8; i < 64; i += 8)
            for (int j = 0; j < 8; j++) r[j] += (double)blk[i + j] * t[i + j];
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        double q = nearbyint(s); /* np.round: half to even */
        out_xy[xy] = (int16_t)((int16_t)q + 128);
    }
}

/* Same, on an already de-zigzagged and dequantised int16 [x][y] block (InverseDCT.__call__ alone). */
void orc_idct_xy(const int16_t *blk, const double *T, int16_t *out_xy) {
    for (int xy = 0; xy < 64; xy++) {
        const double *t = T + xy * 64;
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = (double)blk[j] * t[j];
        for (int i = 8; i < 64; i += 8)
            for (int j = 0; j < 8; j++) r[j] += (double)blk[i + j] * t[i + j];
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        out_xy[xy] = (int16_t)((int16_t)nearbyint(s) + 128);
    }
}

/* ------------------------------------------------------------------------------------------------
 * ResizeGrid.__call__  (:1588-1626): linear interpolation of a component MCU [sw][sh] onto
 * sample_shape [dw][dh] through scipy griddata, then np.round -> int16.  griddata's effect for a
 * fixed pair of shapes is a fixed sparse operator W (<= 3 taps/row, weights n/15)
 * captured from the reference into tests/golden/upsample_W_*.npy and passed in here.
 * sum(n_i * v_i)/15 can never be a half-integer (15 is odd), so the float64 result of griddata
 * (accurate to ~1e-14) rounds the same way as this exact form.
 */
void orc_upsample(const int16_t *in, int n_in, const int16_t *W /* [n_out][n_in] numerators; a row sums to its denominator */,
                  int n_out, int16_t *out) {
    for (int o = 0; o < n_out; o++) {
        long acc = 0, den = 0;              /* a product of (8n - 1)s — 15, 31, 15 * 23 ... — odd in every case */
        const int16_t *w = W + (size_t)o * n_in;
        for (int k = 0; k < n_in; k++) { acc += (long)w[k] * in[k]; den += w[k]; }
        out[o] = (int16_t)nearbyint((double)acc / (double)den);
    }
}

/* ------------------------------------------------------------------------------------------------
 * YCbCr_to_RGB  (:1683-1700), float64, expression order as written, clip then np.round then uint8.
 * ycc: [n][3] int16;  rgb: [n][3] uint8.
 */
void orc_ycbcr_to_rgb(const int16_t *ycc, long n, uint8_t *rgb) {
    for (long i = 0; i < n; i++) {
        double Y = (double)ycc[3 * i], Cb = (double)ycc[3 * i + 1], Cr = (double)ycc[3 * i + 2];
        double R = Y + 1.402 * (Cr - 128.0);
        double G = (Y - 0.34414 * (Cb - 128.0)) - 0.71414 * (Cr - 128.0);
        double B = Y + 1.772 * (Cb - 128.0);
        double c[3] = {R, G, B};
        for (int k = 0; k < 3; k++) {
            double v = c[k] < 0.0 ? 0.0 : (c[k] > 255.0 ? 255.0 : c[k]);
            rgb[3 * i + k] = (uint8_t)nearbyint(v);
        }
    }
}

/* Greyscale tail (:1384-1386): clip int16 to 0..255, cast to uint8. */
void orc_grey_to_u8(const int16_t *y, long n, uint8_t *out) {
    for (long i = 0; i < n; i++) out[i] = (uint8_t)(y[i] < 0 ? 0 : (y[i] > 255 ? 255 : y[i]));
}

/* ------------------------------------------------------------------------------------------------
 * Frame / scan description handed over by the caller (what start_of_scan :505-632 has prepared).
 */
typedef struct {
    int32_t width, height, ncomp;
    int32_t hs[3], vs[3];          /* sampling factors, frame order (Y, Cb, Cr) */
    int32_t qt_sel[3];             /* quantisation table slot per component (0..3) */
    int32_t dc_sel[3], ac_sel[3];  /* Huffman table slots per component (0..3) */
    int32_t restart_interval;
    int32_t mcu_count_h, mcu_count_v;
} OrcScan;

typedef struct {
    /* canonical code book: for each length 1..16 the first code, the count, and the index of its
       first symbol — equivalent to the reference's {codeword-string: value} dict (:366-377) */
    int32_t first_code[17];
    int32_t count[17];
    int32_t first_sym[17];
    uint8_t vals[256];
} OrcHuff;

void orc_build_huffman(const uint8_t *bits /* [16] */, const uint8_t *vals, OrcHuff *h) {
    int32_t code = 0, k = 0;
    memset(h, 0, sizeof(*h));
    for (int l = 1; l <= 16; l++) {
        code <<= 1;                       /* :370 */
        h->first_code[l] = code;
        h->count[l] = bits[l - 1];
        h->first_sym[l] = k;
        code += bits[l - 1];              /* :374, once per symbol */
        k += bits[l - 1];
    }
    memcpy(h->vals, vals, (size_t)(k > 256 ? 256 : k));
}

/* bits_generator / get_bits (:654-695): a FIFO of bits refilled ONE BYTE at a time only when the
 * request exceeds what is queued; after reading a 0xFF byte the next byte is skipped whatever it is
 * (:676-677); restart = drop queued bits, skip two bytes (:667-669). */
typedef struct {
    const uint8_t *file;
    int64_t size, pos;
    uint32_t acc; /* queued bits, right-aligned */
    int nbits;
    int overrun;
} OrcBits;

static inline uint32_t orc_get_bits(OrcBits *b, int amount) {
    while (amount > b->nbits) {
        if (b->pos >= b->size) { b->overrun = 1; return 0; } /* reference: IndexError */
        uint8_t byte = b->file[b->pos++];
        if (byte == 0xFF) b->pos++;
        b->acc = (b->acc << 8) | byte;
        b->nbits += 8;
    }
    if (amount == 0) return 0;
    uint32_t v = (b->acc >> (b->nbits - amount)) & ((1u << amount) - 1);
    b->nbits -= amount;
    return v;
}
static inline void orc_restart(OrcBits *b) { b->nbits = 0; b->acc = 0; b->pos += 2; }

/* next_huffval (:712-722): extend the codeword one bit at a time until it is a key of the table;
 * more than 16 bits -> CorruptedJpeg. Returns -1 for that, -2 on stream overrun. */
static inline int orc_next_huffval(OrcBits *b, const OrcHuff *h) {
    int32_t code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | (int32_t)orc_get_bits(b, 1);
        if (b->overrun) return -2;
        int32_t d = code - h->first_code[l];
        if (d >= 0 && d < h->count[l]) return h->vals[h->first_sym[l] + d];
    }
    return -1;
}

/* bin_twos_complement (:1636-1646) = JPEG EXTEND */
static inline int32_t orc_extend(uint32_t v, int n) {
    if (n == 0) return 0;
    return (v >> (n - 1)) ? (int32_t)v : (int32_t)v - ((1 << n) - 1);
}

/*
 * baseline_dct_scan, entropy part (:734-866, :894-900), interleaved scan of all frame components.
 * coef: [mcu_count][blocks_per_mcu][64] int16, zig-zag order — the array seen at :869 just before
 * undo_zigzag (golden G1), in decode order: MCU raster, component order, block_count (:805).
 * Returns 0 ok, 1 CorruptedJpeg (no code within 16 bits), 2 ran past end of file.
 * *end_pos = file_header after the scan.
 */
int orc_entropy_decode_baseline(const uint8_t *file, int64_t file_size, int64_t start, const OrcScan *sc,
                                const OrcHuff *dc_tabs /* [4] */, const OrcHuff *ac_tabs /* [4] */,
                                int16_t *coef, int64_t *end_pos) {
    OrcBits b = {file, file_size, start, 0, 0, 0};
    int16_t prev_dc[3] = {0, 0, 0};
    int64_t mcu_count = (int64_t)sc->mcu_count_h * sc->mcu_count_v;
    int16_t *blk = coef;
    for (int64_t mcu = 0; mcu < mcu_count;) {
        for (int c = 0; c < sc->ncomp; c++) {
            int repeat = sc->ncomp > 1 ? sc->hs[c] * sc->vs[c] : 1; /* :780-785 */
            const OrcHuff *hd = &dc_tabs[sc->dc_sel[c]], *ha = &ac_tabs[sc->ac_sel[c]];
            for (int r = 0; r < repeat; r++, blk += 64) {
                memset(blk, 0, 64 * sizeof(int16_t));
                int s = orc_next_huffval(&b, hd);                         /* :812 */
                if (s < 0) { *end_pos = b.pos; return s == -1 ? 1 : 2; }
                uint32_t raw = orc_get_bits(&b, s);                       /* :818 */
                if (b.overrun) { *end_pos = b.pos; return 2; }
                int16_t dcv = (int16_t)(orc_extend(raw, s) + prev_dc[c]); /* int16 arithmetic */
                prev_dc[c] = dcv;
                blk[0] = dcv;
                int index = 1;
                while (index < 64) {                                      /* :834 */
                    int hv = orc_next_huffval(&b, ha);
                    if (hv < 0) { *end_pos = b.pos; return hv == -1 ? 1 : 2; }
                    if (hv == 0) break;                                   /* EOB :849 */
                    index += hv >> 4;                                     /* :853-856 */
                    if (index >= 64) break;
                    int n = hv & 15;
                    if (n > 0) {
                        raw = orc_get_bits(&b, n);
                        if (b.overrun) { *end_pos = b.pos; return 2; }
                        blk[index] = (int16_t)orc_extend(raw, n);
                    }
                    index++;
                }
            }
        }
        mcu++;
        if (sc->restart_interval > 0 && mcu % sc->restart_interval == 0 && mcu != mcu_count) { /* :898 */
            orc_restart(&b);
            prev_dc[0] = prev_dc[1] = prev_dc[2] = 0;
        }
    }
    *end_pos = b.pos;
    return 0;
}

/*
 * Reconstruction of a whole interleaved baseline image from its coefficients:
 *   :869-891 dequant, IDCT, block placement in the component MCU (:875-879), upsample when the
 *   component's MCU shape differs from sample_shape (:882-883), store (:889-891);
 *   :1373 crop;  :1382-1386 colour conversion.
 * qt_xy: [4][64] reference-layout tables.  W_up[c]: upsample operator for component c or NULL.
 * planes_out: int16 [width][height][ncomp] (cropped image_array before colour conversion, golden G5);
 * idct_out: optional int16 [nblocks][64] ([x][y] per block, golden G3);
 * rgb_out: uint8 [width][height][3] (or [width][height] for greyscale), golden G6.
 */
int orc_reconstruct_baseline(const OrcScan *sc, const int16_t *coef, const int16_t *qt_xy, const double *T,
                             const int16_t *const *W_up, int16_t *idct_out, int16_t *planes_out,
                             uint8_t *rgb_out) {
    int hmax = 1, vmax = 1;
    for (int c = 0; c < sc->ncomp; c++) { if (sc->hs[c] > hmax) hmax = sc->hs[c]; if (sc->vs[c] > vmax) vmax = sc->vs[c]; }
    if (sc->ncomp == 1) { hmax = sc->hs[0]; vmax = sc->vs[0]; }
    /* sample_shape (:238-240) = max component MCU shape; for one component the scan MCU is 8x8 (:596) */
    int mw = sc->ncomp > 1 ? 8 * hmax : 8, mh = sc->ncomp > 1 ? 8 * vmax : 8;
    int bpm = 0;
    for (int c = 0; c < sc->ncomp; c++) bpm += sc->ncomp > 1 ? sc->hs[c] * sc->vs[c] : 1;
    int64_t mcu_count = (int64_t)sc->mcu_count_h * sc->mcu_count_v;
    int W = sc->width, H = sc->height, nc = sc->ncomp;
    int status = 0;

#pragma omp parallel for schedule(static)
    for (int64_t mcu = 0; mcu < mcu_count; mcu++) {
        int mcu_y = (int)(mcu / sc->mcu_count_h), mcu_x = (int)(mcu % sc->mcu_count_h);
        const int16_t *blk = coef + mcu * bpm * 64;
        int16_t comp_mcu[32 * 32], up[32 * 32], out8[64];
        int64_t bidx = mcu * bpm;
        for (int c = 0; c < nc; c++) {
            int h = nc > 1 ? sc->hs[c] : 1, v = nc > 1 ? sc->vs[c] : 1;
            int cw = 8 * h, ch = 8 * v;
            for (int r = 0; r < h * v; r++, blk += 64, bidx++) {
                orc_dequant_idct(blk, qt_xy + 64 * sc->qt_sel[c], T, NULL, out8);
                if (idct_out) memcpy(idct_out + bidx * 64, out8, sizeof(out8));
                int by = r / h, bx = r % h; /* :875 */
                for (int x = 0; x < 8; x++)
                    for (int y = 0; y < 8; y++) comp_mcu[(bx * 8 + x) * ch + (by * 8 + y)] = out8[x * 8 + y];
            }
            const int16_t *src = comp_mcu;
            if (cw != mw || ch != mh) { /* :882 */
                if (!W_up || !W_up[c]) { status = 3; continue; }
                orc_upsample(comp_mcu, cw * ch, W_up[c], mw * mh, up);
                src = up;
            }
            for (int x = 0; x < mw; x++) {
                int gx = mcu_x * mw + x;
                if (gx >= W) break;
                for (int y = 0; y < mh; y++) {
                    int gy = mcu_y * mh + y;
                    if (gy >= H) break;
                    planes_out[((int64_t)gx * H + gy) * nc + c] = src[x * mh + y];
                }
            }
        }
    }
    if (status) return status;
    if (rgb_out) {
        if (nc == 3) orc_ycbcr_to_rgb(planes_out, (long)W * H, rgb_out);
        else orc_grey_to_u8(planes_out, (long)W * H, rgb_out);
    }
    return 0;
}

/* ================================================================================================
 * Progressive scans  (progressive_dct_scan, :908-1304).
 *
 * Coefficient state between scans lives, in the reference, inside image_array at pixel coordinates
 * (:1029, :1225); here it is the equivalent per-block store  coef[block][64]  in zig-zag order, blocks in
 * the same interleaved order as the baseline seam (MCU raster, component, block_count), so that after the
 * last scan orc_reconstruct_baseline() performs the reference's final pass (:1306-1362: in-place dequant,
 * IDCT, per-block resize, store) unchanged.
 *
 * One call = one SOS.  Restatement notes:
 *   - DC first (:1010-1029): value = (EXTEND(bits) + previous_dc) stored << Al (int16); DC refine (:1036-1038):
 *     value |= bit << Al.
 *   - AC first (:1122-1256): run/size symbols; size 0 with run 15 = 16 zeros; size 0 otherwise = EOB run
 *     (1 << r) + bits(r) (r = 0: run of 1, no bits); value stored << Al; a band finished normally advances the
 *     block counter by one, then the EOB run is added (:1240-1250).
 *   - AC refine (:1176-1292): zero runs count only zero coefficients, non-zero ones passed on the way are
 *     queued; a new value is placed at the next zero position; queued coefficients then receive one bit each,
 *     applied as  value |= bit << Al  on the two's-complement int16 — NOT the spec's "subtract for negative
 *     values"  — and the EOB-run walk queues every non-zero coefficient of the remaining bands.
 *   - restarts are count driven (:1050-1053, :1297-1298); only DC-first scans reset the predictors.
 * Scan geometry as prepared by start_of_scan (:591-621): interleaved DC scans walk MCUs; single-component scans
 * walk the component's own 8x8 blocks, ceil(W_c/8) per row.
 * Returns 0 ok, 1 bad Huffman code, 2 ran past the end of file, 4 coefficient index out of range (reference:
 * IndexError), 5 scan form the reference mishandles and this oracle refuses (single-component DC scan of a
 * subsampled-frame component with h or v > 1).
 */
typedef struct {
    int32_t width, height, ncomp_frame;
    int32_t hs[3], vs[3];            /* frame sampling */
    int32_t n_scan_comp;             /* components in this scan */
    int32_t scan_comp[3];            /* frame index (0..2) of each scan component */
    int32_t dc_sel[3], ac_sel[3];    /* per scan component */
    int32_t ss, se, ah, al;
    int32_t restart_interval;
    int32_t mcu_count_h, mcu_count_v;   /* of THIS scan (:609-621) */
} OrcProgScan;

typedef struct {
    int hmax, vmax, bpm, mcus_x, mcus_y, first_blk[3];
} OrcGrid;

static void orc_grid(const OrcProgScan *s, OrcGrid *g) {
    g->hmax = 1; g->vmax = 1;
    if (s->ncomp_frame > 1)
        for (int c = 0; c < s->ncomp_frame; c++) { if (s->hs[c] > g->hmax) g->hmax = s->hs[c]; if (s->vs[c] > g->vmax) g->vmax = s->vs[c]; }
    g->bpm = 0;
    for (int c = 0; c < s->ncomp_frame; c++) { g->first_blk[c] = g->bpm; g->bpm += s->ncomp_frame > 1 ? s->hs[c] * s->vs[c] : 1; }
    int mw = 8 * g->hmax, mh = 8 * g->vmax;
    g->mcus_x = (s->width + mw - 1) / mw; g->mcus_y = (s->height + mh - 1) / mh;
}

/* block (bx, by) of frame component c -> index into coef[] (interleaved order) */
static inline int64_t orc_block_index(const OrcProgScan *s, const OrcGrid *g, int c, int bx, int by) {
    int h = s->ncomp_frame > 1 ? s->hs[c] : 1, v = s->ncomp_frame > 1 ? s->vs[c] : 1;
    int mx = bx / h, my = by / v;
    return ((int64_t)my * g->mcus_x + mx) * g->bpm + g->first_blk[c] + (by % v) * h + (bx % h);
}

int orc_progressive_scan(const uint8_t *file, int64_t file_size, int64_t start, const OrcProgScan *s,
                         const OrcHuff *dc_tabs, const OrcHuff *ac_tabs, int16_t *coef, int64_t *end_pos) {
    OrcBits b = {file, file_size, start, 0, 0, 0};
    OrcGrid g;
    orc_grid(s, &g);
    const int64_t mcu_count = (int64_t)s->mcu_count_h * s->mcu_count_v;
    const int refining = s->ah != 0;
    int status = 0;

    if (s->ss == 0 && s->se == 63) { /* --------------------------------- one component of a non-interleaved BASELINE file:
        baseline_dct_scan (:734-866, :894-900) with a single scan component; the scan's MCU is one 8x8 block (:612-619) */
        if (s->n_scan_comp != 1) { *end_pos = b.pos; return 5; }
        const int c = s->scan_comp[0];
        if (s->ncomp_frame > 1 && (s->hs[c] > 1 || s->vs[c] > 1)) { *end_pos = b.pos; return 5; }   /* the reference breaks here (:882-891) */
        int16_t prev = 0;
        for (int64_t mcu = 0; mcu < mcu_count;) {
            int16_t *blk = coef + orc_block_index(s, &g, c, (int)(mcu % s->mcu_count_h), (int)(mcu / s->mcu_count_h)) * 64;
            int sz = orc_next_huffval(&b, &dc_tabs[s->dc_sel[0]]);
            if (sz < 0) { *end_pos = b.pos; return sz == -1 ? 1 : 2; }
            uint32_t raw = orc_get_bits(&b, sz);
            if (b.overrun) { *end_pos = b.pos; return 2; }
            prev = (int16_t)(orc_extend(raw, sz) + prev);                                /* :818-820 */
            blk[0] = prev;
            int index = 1;
            while (index < 64) {                                                         /* :834-866 */
                int hv = orc_next_huffval(&b, &ac_tabs[s->ac_sel[0]]);
                if (hv < 0) { *end_pos = b.pos; return hv == -1 ? 1 : 2; }
                if (hv == 0x00) break;                                                   /* :849 */
                index += hv >> 4;
                if (index >= 64) break;                                                  /* :855-856 */
                int n = hv & 0x0F;
                if (n > 0) {
                    uint32_t vb = orc_get_bits(&b, n);
                    if (b.overrun) { *end_pos = b.pos; return 2; }
                    blk[index] = (int16_t)orc_extend(vb, n);
                }
                index += 1;
            }
            mcu++;
            if (s->restart_interval > 0 && mcu % s->restart_interval == 0 && mcu != mcu_count) {   /* :898-900 */
                orc_restart(&b);
                prev = 0;
            }
        }
        *end_pos = b.pos;
        return 0;
    }

    if (s->ss == 0) { /* ------------------------------------------------ DC scan (:974-1057) */
        int16_t prev[3] = {0, 0, 0};
        for (int64_t mcu = 0; mcu < mcu_count;) {
            for (int i = 0; i < s->n_scan_comp; i++) {
                const int c = s->scan_comp[i];
                const int h = s->ncomp_frame > 1 ? s->hs[c] : 1, v = s->ncomp_frame > 1 ? s->vs[c] : 1;
                const int repeat = s->n_scan_comp > 1 ? h * v : 1;
                if (s->n_scan_comp == 1 && (h > 1 || v > 1)) { *end_pos = b.pos; return 5; }
                for (int r = 0; r < repeat; r++) {
                    int bx, by;
                    if (s->n_scan_comp > 1) { bx = (int)(mcu % s->mcu_count_h) * h + r % h; by = (int)(mcu / s->mcu_count_h) * v + r / h; }
                    else { bx = (int)(mcu % s->mcu_count_h); by = (int)(mcu / s->mcu_count_h); }
                    int16_t *blk = coef + orc_block_index(s, &g, c, bx, by) * 64;
                    if (!refining) {
                        int sz = orc_next_huffval(&b, &dc_tabs[s->dc_sel[i]]);
                        if (sz < 0) { *end_pos = b.pos; return sz == -1 ? 1 : 2; }
                        uint32_t raw = orc_get_bits(&b, sz);
                        if (b.overrun) { *end_pos = b.pos; return 2; }
                        int16_t dcv = (int16_t)(orc_extend(raw, sz) + prev[i]);
                        prev[i] = dcv;
                        blk[0] = (int16_t)((int32_t)dcv << s->al);
                    } else {
                        uint32_t bit = orc_get_bits(&b, 1);
                        if (b.overrun) { *end_pos = b.pos; return 2; }
                        blk[0] = (int16_t)(blk[0] | (int16_t)(bit << s->al));
                    }
                }
            }
            mcu++;
            if (s->restart_interval > 0 && mcu % s->restart_interval == 0 && mcu != mcu_count) {
                orc_restart(&b);
                if (!refining) prev[0] = prev[1] = prev[2] = 0;
            }
        }
        *end_pos = b.pos;
        return 0;
    }

    /* ---------------------------------------------------------------- AC scan (:1060-1298) */
    const int c = s->scan_comp[0];
    const OrcHuff *ht = &ac_tabs[s->ac_sel[0]];
    int64_t eob_run = 0;
    int16_t *queue[64];        /* to_refine within one band: pointers to coefficients awaiting a correction bit */
    int nq = 0;
#define BLK(m) (coef + orc_block_index(s, &g, c, (int)((m) % s->mcu_count_h), (int)((m) / s->mcu_count_h)) * 64)
#define REFINE_QUEUE()                                                          \
    do {                                                                        \
        for (int qi = 0; qi < nq; qi++) {                                       \
            uint32_t bit = orc_get_bits(&b, 1);                                 \
            if (b.overrun) { *end_pos = b.pos; return 2; }                      \
            *queue[qi] = (int16_t)(*queue[qi] | (int16_t)(bit << s->al));       \
        }                                                                       \
        nq = 0;                                                                 \
    } while (0)

    for (int64_t mcu = 0; mcu < mcu_count;) {
        int16_t *blk = BLK(mcu);
        int index = s->ss;
        while (index <= s->se) {
            int hv = orc_next_huffval(&b, ht);
            if (hv < 0) { *end_pos = b.pos; return hv == -1 ? 1 : 2; }
            int run = hv >> 4, size = hv & 15, zero_run = 0;
            if (hv == 0) { eob_run = 1; break; }
            else if (hv == 0xF0) zero_run = 16;
            else if (size == 0) {
                uint32_t bits = orc_get_bits(&b, run);
                if (b.overrun) { *end_pos = b.pos; return 2; }
                eob_run = ((int64_t)1 << run) + bits;
                break;
            } else zero_run = run;

            if (!refining) index += zero_run;                        /* :1177-1179 */
            else {
                while (zero_run > 0) {                               /* :1184-1193 */
                    if (index > 63) { *end_pos = b.pos; return 4; }
                    if (blk[index] == 0) zero_run--;
                    else { if (nq < 64) queue[nq++] = &blk[index]; }
                    index++;
                }
            }
            if (size > 0) {                                          /* :1201-1228 */
                uint32_t raw = orc_get_bits(&b, size);
                if (b.overrun) { *end_pos = b.pos; return 2; }
                int32_t val = orc_extend(raw, size);
                if (index > 63) { *end_pos = b.pos; return 4; }
                if (refining) {
                    while (blk[index] != 0) {
                        if (nq < 64) queue[nq++] = &blk[index];
                        index++;
                        if (index > 63) { *end_pos = b.pos; return 4; }
                    }
                }
                blk[index] = (int16_t)(val << s->al);
                index++;
            }
            if (refining) REFINE_QUEUE();                            /* :1231-1232 */
        }
        if (index > s->se) mcu++;                                    /* :1240-1245 */
        if (!refining) { mcu += eob_run; eob_run = 0; }              /* :1248-1250 */
        else {
            while (eob_run > 0) {                                    /* :1259-1276 */
                if (mcu >= mcu_count) { status = 4; break; }         /* reference: index error past the array */
                blk = BLK(mcu);
                if (blk[index] != 0) {
                    /* the reference queues these and reads all their bits after the walk (:1286); nothing else
                       reads the stream in between, so taking each bit right away is the same sequence */
                    uint32_t bit = orc_get_bits(&b, 1);
                    if (b.overrun) { *end_pos = b.pos; return 2; }
                    blk[index] = (int16_t)(blk[index] | (int16_t)(bit << s->al));
                }
                index++;
                if (index > s->se) { eob_run--; mcu++; index = s->ss; }
            }
            if (status) { *end_pos = b.pos; return status; }
            REFINE_QUEUE();                                          /* :1285-1286 */
        }
        if (s->restart_interval > 0 && mcu % s->restart_interval == 0 && mcu != mcu_count) orc_restart(&b);
    }
#undef BLK
#undef REFINE_QUEUE
    *end_pos = b.pos;
    return 0;
}
