/*
 * mijpeg.h — C ABI of libmijpeg.so: the MI355X (gfx950) replacement for the per-MCU hot path of
 * tbpaolini/PyJpegDecoder.
 *
 * The reference is one Python class with no FFI; the seam this library plugs into is method level
 * (SURVEY.md §8b).  Each entry point names the reference code it replaces (all citations are into
 * /root/reference/jpeg_decoder.py):
 *
 *   mj_plan_create / mj_plan_execute    JpegDecoder.baseline_dct_scan       :697-906   (Huffman decode,
 *                                       + InverseDCT.__call__               :1561-1573  dequantise, IDCT,
 *                                       + ResizeGrid.__call__               :1588-1626  upsample,
 *                                       + end_of_image crop / YCbCr_to_RGB  :1373-1386, :1683-1700  colour)
 *   (progressive batches: stage 1 is JpegDecoder.progressive_dct_scan :908-1304, one launch per scan ordinal,
 *    stage 2 is the final pass :1306-1362)
 *   mj_decode_baseline_batch            the same, one call (create + execute + sync + read back)
 *   mj_idct_batch                       the same minus the entropy decoder: caller supplies the zig-zag
 *                                       coefficients seen at :869 (BASELINE.json configs[1], "host Huffman")
 *
 * Inputs are exactly what start_of_scan (:505-650) has prepared when it calls the scan decoder: the
 * file bytes, the offset of the first entropy-coded byte, per-component table selectors, the DHT
 * BITS/HUFFVAL lists, the DQT tables, the restart interval and the MCU geometry — plus the offsets of
 * the restart segments, because stage 1 decodes one restart segment per wavefront.
 *
 * Conventions
 *   - plain C, no torch / numpy types; every pointer is either host or device memory as said by the
 *     accompanying MJ_MEM_* flag; the caller owns every buffer it passes and the library never frees it.
 *   - every function returns MJ_OK (0) or a negative MJ_ERR_*; mj_last_error() gives the text.
 *   - per-image decode outcomes go to status[] (MJ_ST_*), they are not API errors.
 *   - one context per GPU per host thread; a context is not thread safe.
 *   - mj_plan_execute is asynchronous on the plan's stream; everything else is synchronous at return.
 *   - there is NO CPU fallback anywhere in this library.
 */
#ifndef MIJPEG_H
#define MIJPEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJ_VERSION 1

/* API results */
#define MJ_OK               0
#define MJ_ERR_INVALID     (-1)   /* bad argument / inconsistent description */
#define MJ_ERR_HIP         (-2)   /* a HIP runtime call failed (no GPU, OOM, launch failure, ...) */
#define MJ_ERR_UNSUPPORTED (-3)   /* sampling layout / feature outside the MI355X path */

/* per-image status */
#define MJ_ST_OK          0
#define MJ_ST_BAD_CODE    1   /* no Huffman code within 16 bits: the reference raises CorruptedJpeg (:718-719) */
#define MJ_ST_OVERRUN     2   /* a segment needed more bits than it holds (reference: IndexError / garbage)     */
#define MJ_ST_DESYNC      3   /* a segment's MCUs end before its next RSTn: the reference's count-driven restart
                                 (:667-669, :898-900) would lose synchronisation here                           */

#define MJ_ST_TAIL        4   /* MJ_FLAG_GPU_SEGMENT only: the scan is not followed by EOI (more scans, DNL, ...): the
                                 host-side marker loop (:78-110) has to segment this file                          */
#define MJ_ST_UNCONVERGED 5   /* long restart segments (files without DRI) are cut into pieces whose decoder states are found
                                 by a fixed number of synchronisation rounds on the device; this image's states had not
                                 settled when the rounds were over (a pathological or crafted stream): nothing is wrong
                                 with the file — decode it again in a plan created with MJ_FLAG_NO_SYNC                */

#define MJ_ST_INTERNAL    6   /* a wavefront of a fused launch gave up waiting for its workgroup's decoder wavefronts (a bound on
                                 what is otherwise a spin loop: cannot happen; the image's pixels are not valid)        */

/* memory spaces */
#define MJ_MEM_NONE   0
#define MJ_MEM_HOST   1
#define MJ_MEM_DEVICE 2

/* pixel layouts of the RGB / grey output */
#define MJ_LAYOUT_XMAJOR   0   /* reference image_array: (W, H, C), x-major */
#define MJ_LAYOUT_ROWMAJOR 1   /* (H, W, C) */
#define MJ_LAYOUT_PLANAR_XMAJOR   2   /* (C, W, H): the components of the array the reference leaves at :1373-1386 as
                                         three planes per image (one for greyscale), image after image              */
#define MJ_LAYOUT_PLANAR_ROWMAJOR 3   /* (C, H, W) */

/* flags of mj_batch.flags */
#define MJ_FLAG_KEEP_COEF    1u   /* keep the zig-zag coefficient array (:869 seam) readable after execute */
#define MJ_FLAG_KEEP_PLANES  2u   /* also produce the cropped int16 YCbCr planes (:1373 seam)               */
#define MJ_FLAG_KEEP_IDCT    4u   /* also produce the per-block IDCT output (:872 return value)             */
#define MJ_FLAG_EXACT_ONLY   8u   /* stage 2: use only the exact-order fp64 summation (no fast path)        */
#define MJ_FLAG_SPEC_REFINE 16u   /* progressive AC refinement per ITU-T T.81 G.1.2.3 (move negative values away
                                     from zero) instead of the reference's `|=` on two's complement (:1114)  */

#define MJ_FLAG_GPU_SEGMENT 32u  /* baseline batches: find the RSTn markers and the end of the scan on the GPU (what
                                     _parse.py's find_entropy_end / find_restart_segments do on the host).  Every
                                     image then has n_segments = 1 and seg_begin/seg_end = first entropy-coded byte /
                                     any bound at or behind the end of the scan (e.g. the end of the file); the blob
                                     must be 16-byte aligned with 16 readable bytes behind blob_len.  (The marker scan
                                     and stage 0 read it inside [seg_begin, seg_end) + 16 bytes and the lane forms of
                                     stage 1 read stage 0's stream, so the 512-byte rule of MJ_MEM_DEVICE blobs below
                                     does not apply; a plan that takes the wave form of stage 1 — small batches,
                                     unusual sampling layouts — reads the blob itself, further ahead, and works on a
                                     padded copy of it made at plan creation when the blob lacks those 512 bytes.)  */

#define MJ_FLAG_NO_SYNC     64u  /* never cut restart segments into synchronised pieces: one serial walk per segment (the
                                     fallback for images that came back MJ_ST_UNCONVERGED)                          */

typedef struct mj_context mj_context;
typedef struct mj_plan mj_plan;

/* One DHT table (:293-324): BITS[16] then HUFFVAL. */
typedef struct {
    uint8_t bits[16];
    uint8_t vals[256];
} mj_huff_spec;

/* One image = one interleaved baseline scan (or a single-component greyscale scan). */
typedef struct {
    int32_t width, height;        /* image_width, image_height (:160-169)                                    */
    int32_t ncomp;                /* 1 or 3 (:177-183)                                                        */
    int32_t hs[3], vs[3];         /* sampling factors in frame order Y, Cb, Cr (:205-207)                     */
    int32_t qt_sel[3];            /* index into mj_batch.qt of each component's table (:212)                  */
    int32_t dc_sel[3], ac_sel[3]; /* index into mj_batch.huff of each component's DC / AC table (:543-544)    */
    int32_t restart_interval;     /* MCUs per restart segment, 0 = none (:476)                                */
    int32_t mcu_count_h, mcu_count_v; /* (:609-619)                                                           */
    int32_t n_segments;           /* number of restart segments = ceil(mcu_count / restart_interval) or 1     */
    int64_t first_segment;        /* index of this image's first entry in seg_begin / seg_end                 */
} mj_image_desc;

/* One SOS of an image that is decoded scan by scan: what start_of_scan (:505-650) hands to progressive_dct_scan
 * (:908) for a progressive (SOF2) image, or to baseline_dct_scan (:697) for a baseline image with one scan per
 * component (then ss = 0, se = 63, ah = al = 0, n_comp = 1, no component subsampled).
 * Scans must be listed image by image, in file order. */
typedef struct {
    int32_t image;                /* index into mj_batch.images                                               */
    int32_t n_comp;               /* components in the scan (:530)                                            */
    int32_t comp[3];              /* their positions in the frame (0 = Y, 1 = Cb, 2 = Cr)                     */
    int32_t dc_sel[3], ac_sel[3]; /* per scan component: index into mj_batch.huff (:543-544)                  */
    int32_t ss, se, ah, al;       /* spectral selection, successive approximation (:559-562)                  */
    int32_t restart_interval;     /* value in force for this scan (:501-502)                                  */
    int32_t mcu_count_h, mcu_count_v; /* of this scan (:609-621): interleaved MCUs or the component's 8x8 blocks */
    int32_t n_segments;           /* restart segments of this scan                                            */
    int64_t first_segment;        /* index of the scan's first entry in seg_begin / seg_end                   */
} mj_scan_desc;

typedef struct {
    int32_t n_images;
    const mj_image_desc *images;          /* host */

    const uint8_t *blob;                  /* the file bytes of all images, back to back or not              */
    int64_t blob_len;
    int32_t blob_mem;                     /* MJ_MEM_HOST or MJ_MEM_DEVICE (device: 4-byte aligned, must stay valid for the
                                             plan, and blob_len must include at least 512 readable bytes behind the last
                                             segment's end: the stage-1 bit readers fetch ahead; checked at plan creation.
                                             Host blobs are uploaded with that slack added.) */

    int64_t n_segments;                   /* total entries of the two arrays below                          */
    const int64_t *seg_begin;             /* host: blob offset of the first entropy byte of each segment; list them in
                                             ascending blob order (any order works, but only ordered batches take the
                                             lane-parallel stage 1)                                          */
    const int64_t *seg_end;               /* host: blob offset one past its last entropy byte (= position of
                                             the RSTn / next marker)                                         */
    int32_t n_huff;
    const mj_huff_spec *huff;             /* host */
    int32_t n_qt;
    const uint16_t *qt;                   /* host: n_qt tables of 64 entries in zig-zag (file) order (:454)  */

    int32_t layout;                       /* MJ_LAYOUT_*                                                     */
    uint32_t flags;                       /* MJ_FLAG_*                                                       */

    int32_t n_scans;                      /* 0 = single-scan baseline batch; > 0 = scan-by-scan batch: every image is
                                             progressive (SOF2) or non-interleaved baseline                    */
    const mj_scan_desc *scans;            /* host; the images' own n_segments / first_segment / table selectors
                                             are ignored in a scan-by-scan batch                              */
} mj_batch;

/* Sizes and per-image offsets of a plan's outputs (all outputs are packed image after image). */
typedef struct {
    int64_t total_blocks;                 /* coefficient blocks in the batch                                 */
    int64_t total_mcus;
    int64_t total_pixels;                 /* sum of width*height                                              */
    int64_t rgb_bytes;                    /* sum of width*height*ncomp                                        */
    int64_t entropy_bytes;                /* sum of segment lengths                                           */
} mj_plan_info;

/* ---- context ------------------------------------------------------------------------------------- */
int mj_create(int device_id, mj_context **out);
void mj_destroy(mj_context *ctx);
const char *mj_last_error(const mj_context *ctx);   /* ctx may be NULL: error of the failed mj_create */
int mj_version(void);
/* Make the context's stream wait for a hipEvent_t (passed as void*) recorded elsewhere — e.g. behind the upload of a
 * batch's files on a copy stream — before anything queued on it afterwards runs. */
int mj_context_wait_event(mj_context *ctx, void *hip_event);

/* ---- plan: upload once, execute many times (bench.py times mj_plan_execute only) ------------------ */
int mj_plan_create(mj_context *ctx, const mj_batch *batch, mj_plan **out);
void mj_plan_destroy(mj_plan *plan);
int mj_plan_get_info(const mj_plan *plan, mj_plan_info *info);
/* Which form of stage 1 the plan chose: one restart segment per wavefront, one per lane, long segments
 * cut into self-synchronised pieces, or scan by scan (progressive / non-interleaved); MJ_FORM_WG_TABLES is or-ed in when
 * the batch has more Huffman tables than LDS holds and every workgroup loads only the tables of its own images,
 * MJ_FORM_RESOLVED when the lane / synchronisation form decodes with the resolved 13-bit AC tables (huffman_lanes13.hip:
 * every batch whose distinct tables fit LDS in that format — at most 3 AC and 4 DC tables). */
#define MJ_FORM_WAVE      0
#define MJ_FORM_LANES     1
#define MJ_FORM_SYNC      2
#define MJ_FORM_SCANS     3
#define MJ_FORM_WG_TABLES 16
#define MJ_FORM_RESOLVED  32
#define MJ_FORM_COUNT_RESOLVED 128   /* MJ_FORM_SYNC: the counting walks run on resolved tables with a repair work list
                                        (huffman_sync.hip: k_count) instead of the classic rounds; MJ_FORM_SCANS: the first AC
                                        scans are walked in self-synchronising chunks (progressive_chunks.hip) */
#define MJ_FORM_FUSED     64   /* mj_plan_execute runs stages 1 and 2 as ONE launch (fused.hip): lane-walk wavefronts and
                                  reconstruction wavefronts side by side in one workgroup per CU.  Uniform x-major batches of
                                  4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 files with one restart interval per MCU row and the resolved
                                  tables; mj_plan_execute_stage1 / _stage2 of such a plan are the two launches as ever */
int mj_plan_stage1_form(const mj_plan *plan);
/* offsets (in elements of the respective output) of image i inside the packed outputs */
int mj_plan_image_offsets(const mj_plan *plan, int32_t image, int64_t *block_off, int64_t *rgb_off);

/* Launch stage 1 (Huffman) + stage 2 (dequant/IDCT/upsample/colour) on `stream` (a hipStream_t passed as
 * void*, NULL = the context's own stream).  `rgb_device` is a device buffer of rgb_bytes bytes, or NULL to
 * use a plan-owned one.  Asynchronous for every kind of batch: nothing in it waits for the device. */
int mj_plan_execute(mj_plan *plan, void *stream, uint8_t *rgb_device);
/* (A plan owns ONE coefficient store, one set of stage-1 scratch and one stage-2 work counter: executes of the same plan must
 * not overlap — queue them on one stream, or wait for mj_plan_sync before using another.  Different plans overlap freely.) */
/* The two stages separately (profiling, config 2). */
int mj_plan_execute_stage1(mj_plan *plan, void *stream);
int mj_plan_execute_stage2(mj_plan *plan, void *stream, uint8_t *rgb_device);
int mj_plan_sync(mj_plan *plan);   /* waits for this plan's latest execute (on whichever stream it went), not for other work on that stream */

/* Device pointers of plan-owned buffers (valid until mj_plan_destroy): zero-copy hand-off to torch etc. */
int mj_plan_device_buffers(mj_plan *plan, int16_t **coef, uint8_t **rgb, int16_t **planes, int16_t **idct);

/* Copy results to host memory (after mj_plan_sync).  Any pointer may be NULL. */
int mj_plan_read(mj_plan *plan, uint8_t *rgb_host, int16_t *coef_host, int16_t *planes_host,
                 int16_t *idct_host, int32_t *status_host);
/* Replace the plan's coefficient array (config 2: coefficients decoded elsewhere). mem = MJ_MEM_*. */
int mj_plan_write_coef(mj_plan *plan, const int16_t *coef, int32_t mem);

/* Test hook, host only (no GPU, no context): the form stage 1 would take for a batch with these restart-segment byte lengths —
 * csrc/form_select.h's rule, the one mj_plan_create applies.  traits: 1 a table serves as DC and AC table, 2 segments not in blob
 * order, 4 progressive, 8 a sampling layout outside the common five, 16 MJ_FLAG_GPU_SEGMENT, 32 ... with one segment per image,
 * 64 a DC size above 15, 128 MJ_FLAG_NO_SYNC, 256 per-workgroup table lists do not fit.  force: MJ_HUFFMAN's value or NULL;
 * forced_chunk: MJ_SYNC_CHUNK or 0.  out = { MJ_FORM_* (| MJ_FORM_WG_TABLES), chunk bytes, chunks, 1 if the segments would be
 * dealt out by length }. */
int mj_debug_stage1_form(const int32_t *seg_len, int64_t n_segs, uint64_t blob_len, int32_t n_huff, uint32_t traits, const char *force,
                         int32_t forced_chunk, int32_t out[4]);

/* Test hook, host only: the tables the counting walks of MJ_FORM_SYNC look symbols up in (csrc/huffman_sync.hip: k_count), as
 * mj_plan_create builds them for a batch with these n_huff <= 8 tables — roles[t] 1 = DC table, 2 = AC table —, wbits index bits.
 * *tab_bytes = bytes from one table to the next; out (may be NULL) receives n_huff * tab_bytes / 4 words.  MJ_ERR_UNSUPPORTED:
 * such a batch takes the classic rounds (a table in both roles, a DC size above 15, tables too large). */
int mj_debug_count_tables(const mj_huff_spec *huff, int32_t n_huff, const int32_t *roles, int32_t wbits, uint32_t *out, int64_t cap_words,
                          int32_t *tab_bytes);

/* Test hook, host only: how a fused launch (MJ_FORM_FUSED) would be cut for a batch of n_images images of segments_per_image
 * restart segments on a chip of `cus` CUs — out = { applies (LDS), images per workgroup and pass, producer wavefronts, lanes per
 * producer, consumer wavefronts beside them, bytes of LDS the producers take, passes per workgroup, workgroups }. */
int mj_debug_fused_shape(int32_t cus, int32_t n_ac, int32_t n_dc, int32_t ac_slot_bytes, int32_t hmax, int32_t vmax, int32_t transposed,
                         int32_t n_images, int32_t segments_per_image, int32_t want_consumers, int32_t out[8]);

/* Test hook, host only: would mj_plan_execute of such a batch be ONE fused launch (csrc/form_select.h: fused_applies, the rule
 * mj_plan_create applies before it asks mj_debug_fused_shape's question)?  layout: MJ_LAYOUT_*; every image mcus_per_row x
 * mcu_rows MCUs with one restart interval (0: none); traits: 1 not the lane form on resolved tables, 2 segments dealt out by length
 * (files of mixed content), 4 segments in another order, 8 images of several geometries, 16 a sampling layout outside the common
 * five, 32 progressive, 64 images with different restart intervals; flags: the plan's MJ_FLAG_*.
 * *mode_out = 0 the two launches, 1 fused with whole images per workgroup, 2 fused with the hand-off across workgroups. */
int mj_debug_fused_applies(int32_t layout, int32_t ncomp, int32_t hmax, int32_t vmax, int32_t mcus_per_row, int32_t mcu_rows,
                           int32_t restart_interval, int32_t n_images, uint32_t traits, uint32_t flags, int32_t *mode_out);

/* Test hook, host only: which scans of a progressive batch would be walked as scout + parts (csrc/form_select.h:
 * choose_prog_split, the rule mj_plan_create applies).  mode: MJ_PROG_SPLIT (1 = by the size of the batch); n_bands: band launches
 * per scan (MCU rows / rows per band); wave_slots: CUs x 32, 0 = MI355X's; parts: MJ_PROG_PARTS or 0 (not set: the rule may
 * choose); per scan its image, restart segments and entropy-coded bytes (< 0: not a refining AC scan of one component).
 * split_out[k] = 1: scan k is split; *parts_out = parts per band. */
int mj_debug_prog_split(int32_t mode, int32_t n_images, int32_t n_bands, int32_t wave_slots, int32_t parts, int32_t n_scans,
                        const int32_t *image, const int32_t *n_segments, const int64_t *bytes, uint8_t *split_out, int32_t *parts_out);

/* Test hook: every byte of the plan's coefficient store := byte_value (synchronous).  The parity tests poison the store in
 * front of a fused execute: a reconstruction wavefront that read a block before its decoder wavefront had written it would
 * show (a store that still holds the previous execute's blocks of the same files hides exactly that). */
int mj_plan_fill_coef(mj_plan *plan, int byte_value);

/* ---- one-shot conveniences ----------------------------------------------------------------------- */
/* create + execute + sync + read + destroy; rgb_out/status_out host, coef_out may be NULL. */
int mj_decode_baseline_batch(mj_context *ctx, const mj_batch *batch, uint8_t *rgb_out, int16_t *coef_out,
                             int32_t *status_out);
/* Stage 2 only on caller-supplied coefficients (host), descriptors as above (segment fields ignored). */
int mj_idct_batch(mj_context *ctx, const mj_batch *batch, const int16_t *coef, uint8_t *rgb_out);

/* ---- measurement --------------------------------------------------------------------------------- */
/* Average device time (ms) of each stage's kernel over `iters` back-to-back launches on the plan's
 * stream, measured with HIP events recorded on that stream. */
int mj_plan_time_stages(mj_plan *plan, int iters, uint8_t *rgb_device, float *stage1_ms, float *stage2_ms);
/* The same for the launches mj_plan_execute makes.  A plan whose execute is ONE fused launch (MJ_FORM_FUSED): front_ms = what
 * runs in front of it (marker scan with MJ_FLAG_GPU_SEGMENT, stage 0), main_ms = the fused launch.  Any other plan: the two
 * stages as mj_plan_time_stages reports them (front = stage 0+1, main = stage 2). */
int mj_plan_time_execute(mj_plan *plan, int iters, uint8_t *rgb_device, float *front_ms, float *main_ms);

/* Placement tuning of a plan that is executed many times into ONE output buffer (a service's output slot, a benchmark's step).
 * Where the plan's coefficient store lies relative to that buffer — physically: nothing the virtual addresses show — puts the
 * fused launch (MJ_FORM_FUSED) into one of two classes 8-9 % apart (profiles/r06_placement.txt).  Tries `candidates` (1..16)
 * stores — the plan's own, then fresh allocations —, each with a few timed executes into rgb_device on `stream` (NULL: the
 * context's), keeps the fastest and releases the others; then the same for the plan's stage-0 stream buffer (the other buffer
 * the launch's traffic runs through).  ms_out[candidates] (may be NULL): ms per execute of each coefficient store tried (0 = not
 * tried); *chosen (may be NULL): which one stayed; *best_ms (may be NULL): ms per execute with what the plan ends up with.  The
 * output buffer is the caller's: a caller that owns several can call this for each and keep the best pair (bench.py does).  Plans
 * that are not fused are left alone (all zeros).  Synchronous. */
int mj_plan_tune_placement(mj_plan *plan, void *stream, uint8_t *rgb_device, int32_t candidates, float *ms_out, int32_t *chosen,
                           float *best_ms);

/* The plain device-to-device copy the rooflines are held against (SURVEY 8d's second denominator): `bytes` (a multiple of 16)
 * copied `iters` times between two buffers of the context's own by a kernel that moves sixteen bytes per lane, in each of a few
 * launch shapes (workgroups per CU, loads in flight, temporal or not: what a copy reaches on this chip depends on them by
 * 20 %); device time per copy in ms of the BEST shape (HIP events on the context's stream).  2 * bytes / ms = the copy rate
 * the chip reaches between these two buffers — a ceiling to hold the decode kernels against, not an average. */
int mj_device_copy_rate(mj_context *ctx, int64_t bytes, int iters, float *ms_per_copy);
/* The shader clock (MHz) the chip held during the context's latest fused launch (MJ_FORM_FUSED) and that launch's duration as
 * its first workgroup saw it: the launch leaves the shader-clock counter and the 100 MHz counter at its start and end.  The
 * launch is power-limited on MI355X, so a time without its clock does not compare across boards.  Zeros: no fused launch yet.
 * Call after the launch has completed (mj_plan_sync). */
int mj_context_launch_clock(mj_context *ctx, float *shader_mhz, float *launch_ms);

/* Test and tuning switches, process-wide.  The defaults are what the library measured as best; the parity tests use the
 * switches to force every form of a stage through the same inputs, the probe scripts to sweep geometries.  The library does
 * NOT read them from the environment (a stray variable must not change how a production decode runs).  Read when a plan is
 * created (forms, orders, chunk sizes) or when it executes (lane geometry).  value NULL or "" = back to the default.
 *   MJ_HUFFMAN        wave | lanes | lanes11 | sync   stage-1 form              MJ_SEG_ORDER     blob | binned | striped
 *   MJ_SYNC_ROUNDS    0..64  repair rounds (classic) / chunks a repair lane may walk on (resolved); 0 = no repairs
 *   MJ_SYNC_CHUNK     256..65536 bytes   MJ_SYNC_WARM   run-up bytes in front of every chunk (half a chunk)
 *   MJ_SYNC_COUNT     classic | resolved   the counting walks of MJ_FORM_SYNC: the round-3 kernel and its repair rounds, or the
 *                     walk on resolved tables with a repair work list (default wherever the batch allows: <= 8 tables, one role each)
 *   MJ_SYNC_BITS      10..13  index bits of those tables (12, fewer if LDS asks for it)
 *   MJ_PROG_BANDS     0 | 1   MJ_PROG_ROWS  frame MCU rows per band   MJ_PROG_FAST  0 | 1 (0 = the general scan walk only)
 *   MJ_PROG_SPLIT     0 | 1 | 2 | 3  refining AC scans as scout + parts: never | by the size of the batch (all of them, then only each
 *                     image's largest with two parts, then none) | always | each image's largest
 *   MJ_PROG_PARTS     1..8  parts per band of a split scan (4)
 *   MJ_PROG_CHUNKS    0 | 1 | 2  the first AC scans of progressive files in self-synchronising chunks, one per lane, in front of the
 *                     band pipeline: never | from 2 048 images on | always;  MJ_PROG_CHUNK  128..65536 bytes per chunk (512)
 *   MJ_LANES_WAVES    1..16   MJ_LANES_PER_WAVE  1..64 (the 11-bit lane form reads 1 as 2)   MJ_LANES_RING  64 | 128
 *   MJ_STAGE2_CHUNK   1..4096 strips per stage-2 job
 *   MJ_FUSED          0 | 1  (0 = mj_plan_execute always launches the stages separately)
 *   MJ_FUSED_CONSUMERS 0..15  reconstruction wavefronts beside the lane walk of a fused launch (8, or as many as LDS allows)
 *   MJ_FUSED_PRODUCERS 1..8   walking wavefronts of a fused launch (fewer, fuller ones: 272 segments as 6 x 46 lanes instead of 8 x 34)
 *   MJ_FUSED_ACBITS    10..13  index bits of a fused launch's AC tables (12; 11 frees 16 KB of LDS for two more strips)
 *   MJ_FUSED_SIMD_SPLIT 0 | 1  the walking wavefronts on SIMDs 0-1, the reconstructing ones on SIMDs 2-3 (default 0: mixed)
 *                     (the last three: the balance experiments of profiles/r06_fused_balance.txt — measured, defaults unchanged)
 *   MJ_FUSED_PIECE    1..4096  MCUs per job of a row-major fused launch (its jobs are pieces of MCU rows; 20, rounded down to strips)
 *   MJ_FUSED_LUMA13   0 | 1  component 0's AC table of a fused launch with a 13-bit main level (default: only where the segments are
 *                     dealt out by length) or with 12 bits like the others
 *   MJ_FUSED_PATIENCE 0..1000000  polls before a consumer of a fused launch whose jobs cross workgroups gives a job up to the
 *                     clean-up launch (2000; 0: every job that is not ready at once — the tests' way into that path)
 * Returns MJ_ERR_INVALID for a name that is none of these AND for a value outside the range or word list given here (a probe
 * sweep must not report the default under another label); the option then keeps what it had.  Values are copied when they are
 * read: setting an option from one thread while another creates or executes a plan is safe (that plan sees the old or the new
 * value, each option read once).  mj_plan_stage1_form() reports what a plan ended up with. */
int mj_set_option(const char *name, const char *value);
/* The value an option holds ("" = the default) into value_out[cap]; MJ_ERR_INVALID for a name that is no option. */
int mj_get_option(const char *name, char *value_out, int32_t cap);

/* Diagnostic of the fast stage 2 (reference :1561-1573): of the blocks the plan's latest stage-2 execute WITH seam outputs
 * (MJ_FLAG_KEEP_IDCT / MJ_FLAG_KEEP_PLANES) put through the IDCT, counts[0] = all of them, counts[1] = how many the fp32
 * first level could not decide (they went to the fp64 level), counts[2] = how many of those went on to the exact-order
 * routine.  Waits for that execute. */
int mj_plan_idct_levels(mj_plan *plan, uint64_t counts[3]);

/* ---- host-side helper (no GPU needed) -------------------------------------------------------------- */
/* The IDCT table as the library builds it: 4096 doubles laid out [u*8+v][x*8+y], the transpose of the
 * reference's InverseDCT.idct_table (:1541-1553).  Lets CPU tests pin it bit-for-bit. */
void mj_host_idct_table(double *tt);

/* ---- host front end (no GPU work, no context) -------------------------------------------------------
 * Header parse + batch assembly for the everyday case, on host threads: what pyjpegdecoder_amd/_parse.py
 * (parse_jpeg(headers_only=True): the reference's marker loop :78-110 and its SOF0/DHT/DQT/DRI/SOS handlers
 * :112-650, stopped at the SOS) and batch.prepare_batch do in Python, file by file.  For every file: walk the
 * segments in front of the scan, copy the file to blob + file_off[i], fill images[i] and its one byte range
 * (seg_begin[i] = first entropy-coded byte, seg_end[i] = end of the file: the arrays of a MJ_FLAG_GPU_SEGMENT
 * batch), and number the distinct Huffman / quantisation tables of the batch in order of first use.
 *
 * Accepted: SOF0, 8 bit, 1 or 3 components, a first SOS naming every frame component in frame order, with only
 * APPn / COM / DQT / DHT / DRI / SOF0 segments in front of it.  Everything else (progressive, scans of single
 * components, DNL, unknown markers, short or inconsistent headers) is not diagnosed here: the call returns
 * MJ_HOST_DECLINED with declined_file = the first such file and the caller runs the full marker loop, which
 * raises what the reference raises.  All arrays are the caller's (mj_image_desc images[n_files], int64
 * seg_begin/seg_end[n_files], huff[huff_cap], qt[qt_cap * 64]; 6 / 3 per file always suffice); file_off must be
 * ascending multiples of 4 with file_off[i] + sizes[i] <= file_off[i + 1] (blob_len for the last); the gaps and the
 * tail of the blob are zeroed (stage 1 reads ahead of a segment's end). */
#define MJ_HOST_DECLINED 1
typedef struct {
    int32_t n_files;
    const uint8_t *const *files;          /* in: the files' bytes                                             */
    const int64_t *sizes;                 /* in                                                                */
    const int64_t *file_off;              /* in: blob offset of every file                                     */
    uint8_t *blob;                        /* out: host buffer of blob_len bytes                                */
    int64_t blob_len;
    mj_image_desc *images;                /* out                                                               */
    int64_t *seg_begin, *seg_end;         /* out                                                               */
    mj_huff_spec *huff;                   /* out */
    int32_t huff_cap;
    uint16_t *qt;                         /* out: zig-zag order, like mj_batch.qt                              */
    int32_t qt_cap;
    int32_t n_threads;                    /* in: host threads to use (>= 1)                                    */
    int32_t n_huff, n_qt;                 /* out: distinct tables written                                      */
    int32_t declined_file;                /* out: -1, or the first file this front end does not take           */
    uint8_t *skip;                        /* in, optional: n_files bytes.  Non-NULL: files the front end does not take are
                                             marked 1 here and left out (their slot in the blob stays zeroed) instead of
                                             ending the call; the output arrays then hold the accepted files only, in order */
    int32_t n_accepted;                   /* out: files assembled (= n_files without `skip`)                   */
} mj_host_job;
int mj_host_assemble(mj_host_job *job);

#ifdef __cplusplus
}
#endif
#endif /* MIJPEG_H */
