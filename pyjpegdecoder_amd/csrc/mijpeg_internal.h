// Internal device/host structures of libmijpeg.so (gfx950 only).
#pragma once
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mijpeg.h"

namespace mj {

constexpr int kWave = 64;               // CDNA4 wavefront
constexpr int kLutBits = 9;             // primary Huffman LUT: codes up to 9 bits resolve in one LDS read
constexpr int kLutSize = 1 << kLutBits;
constexpr int kMaxBlocksPerMcu = 16;    // the common layouts have <= 6 (4:2:0 = 4 Y + Cb + Cr); T.81 allows 10 per interleaved scan, the reference does not check (:774-785)
constexpr int kMaxTabsPerImage = 6;     // DC+AC for up to three components

// One Huffman table on the device.  `lut` is staged into LDS by the stage-1 kernel; the long-code
// side (codes of 10..16 bits, a few percent of symbols) stays in global memory / L2.
struct DevHuff {
    uint16_t lut[kLutSize];   // (len << 8) | symbol for codes <= kLutBits bits, 0 = longer code
    int32_t first_code[17];   // canonical code book per length (jpeg_decoder.py:366-377)
    int32_t count[17];
    int32_t first_sym[17];
    uint8_t vals[256];
    uint8_t pad[4];
};
static_assert(sizeof(DevHuff) % 8 == 0, "DevHuff must keep 8-byte alignment in arrays");

// Per-image description used by both kernels.
struct DevImage {
    int32_t width, height, ncomp;
    int32_t hmax, vmax;               // MCU = (8*hmax) x (8*vmax) pixels
    int32_t blocks_per_mcu;
    int32_t mcu_count_h, mcu_count_v;
    int32_t restart_interval;
    int32_t n_tabs;                   // distinct Huffman tables of this image (<= kMaxTabsPerImage)
    int32_t tab_index[kMaxTabsPerImage];   // indices into the batch's DevHuff array
    // for each block of an MCU, in decode order (jpeg_decoder.py:774, :805):
    // (the common layouts have at most 6 blocks per MCU and the lane / synchronisation forms read the first 8 entries as one
    // 64-bit word; layouts with more blocks — kMaxBlocksPerMcu — are decoded by the wave form and k_reconstruct_generic)
    uint8_t blk_comp[kMaxBlocksPerMcu];     // component 0..2
    uint8_t blk_dc_slot[kMaxBlocksPerMcu];  // slot (0..n_tabs-1) of its DC table in the wave's LDS copy
    uint8_t blk_ac_slot[kMaxBlocksPerMcu];
    int32_t qt_index[3];              // per component, into the batch's quantisation tables
    uint8_t comp_h[4], comp_v[4];     // sampling factors per component (jpeg_decoder.py:205-207)
    uint8_t comp_first[4];            // first block of the component inside an MCU
    int32_t generic;                  // 1 = a layout outside the common ones: any factors 1..4 per component
    int32_t pad0;
    int64_t block_off;                // first coefficient block of this image in the packed coef array
    int64_t mcu_off;                  // first MCU of this image in the batch-wide MCU numbering
    int64_t rgb_off;                  // byte offset of this image in the packed RGB output
    int64_t pix_off;                  // pixel offset (for the planes output: *ncomp int16 each)
};

// One restart segment = the unit of work of one stage-1 wavefront.
struct DevSegment {
    int64_t begin;        // blob offset of the first entropy-coded byte
    int32_t len;          // bytes up to the terminating marker
    int32_t image;
    int32_t mcu0;         // first MCU (within the image) of the segment
    int32_t n_mcu;
    int32_t last;         // 1 = last segment of its image (trailing bytes are not a desync)
    int32_t pad;
};

// One image's byte range for the on-GPU restart-marker scan (MJ_FLAG_GPU_SEGMENT, destuff.hip)
struct DevScanJob {
    int64_t begin, end;   // blob offsets: first entropy-coded byte, a bound at or behind the end of the scan
    int64_t first_seg;    // index of the image's first DevSegment
    int32_t n_seg;        // ceil(mcu_count / restart_interval), 1 without restart interval
    int32_t image;
};

// One chunk of a restart segment's stage-0 stream (huffman_sync.hip): the unit of the synchronisation passes that
// find, inside long segments (files without restart markers), places where an independent decoder can start.
struct DevChunk {
    int32_t seg;          // index of the DevSegment
    int32_t j;            // chunk number within the segment: bits [j * cbits, (j + 1) * cbits)
};
// One piece of a long restart segment for stage 0 (destuff.hip): `len` source bytes from byte `off` of the segment
struct DevPiece {
    int32_t seg;
    int32_t first;        // index of the segment's first piece
    int32_t off, len;
};
// What a counting pass learns about a chunk, decoded from the entry state its predecessor handed over.
struct DevChunkOut {
    uint64_t entry;       // the entry state this record was computed from (a later round skips the chunk if unchanged)
    int32_t blocks;       // blocks completed in the chunk
    int32_t bnd_pos;      // bit position of the first MCU boundary inside the chunk, -1 if none
    int32_t bnd_blocks;   // blocks completed before that boundary
    int16_t dc_bnd[3];    // DC differences per component summed up to the boundary (int16 wrap, :818-820)
    int16_t dc_sum[3];    // ... over the whole chunk
};
static_assert(sizeof(DevChunkOut) == 32, "DevChunkOut layout");
// A virtual segment: a run of whole MCUs inside a restart segment with everything a decoder needs to start there.
struct DevVSeg {
    uint32_t voff0;       // byte offset of the parent segment's stream in the stage-0 buffer
    int32_t bit0, bit_end;    // first bit, one past the last bit (relative to the parent segment's stream)
    int32_t image, mcu0, n_mcu;
    int16_t pred[3];      // DC predictors at the first MCU
    int16_t last;         // 1 = runs to the end of its restart segment (padding bits may follow)
};
static_assert(sizeof(DevVSeg) == 32, "DevVSeg layout");

// One SOS of a progressive image (device copy of mj_scan_desc with table indices resolved).
struct DevProgScan {
    int32_t image, n_comp;
    int32_t comp[3];
    int32_t dc_tab[3], ac_tab[3];   // indices into the batch's DevHuff array
    int32_t ss, se, ah, al;
    int32_t mcu_count_h, mcu_count_v;
    int32_t level;                  // dependency level: scans of level L need the same rows of level L - 1 scans done
    int32_t split;                  // 1: a refining AC scan walked twice (progressive_fast.hip): a scout that only follows the bit
                                    // position, and — one launch behind, at level + 1 — a few walks per band that place
};

// What a scan's restart segment carries from one band of MCU rows to the next (progressive.hip).
struct DevProgState {
    uint64_t bb;
    int32_t pos, bc, pad;
    int32_t eobrun;
    int32_t pred[3];
    int32_t err;
    int32_t mcu_next;               // first MCU the next band starts with
    int32_t reserved;
};

// Where a split scan's scout stood at the first MCU of a part of a band (progressive_fast.hip): what the part's walk starts from.
struct DevProgSub {
    int32_t pos, eobrun, err;
    int32_t mcu;                    // the MCU this entry belongs to (bands alternate between two sets of entries: a stale one does not match)
};
constexpr int kProgSub = 8;         // parts per band: at most (entries per band and segment)

// One restart segment of one progressive scan.
struct DevProgSeg {
    int64_t begin;
    int32_t len;
    int32_t scan;         // index into the DevProgScan array
    int32_t mcu0, n_mcu;  // in units of the scan's own MCUs
    int32_t last;
    int32_t stream_slot;  // the segment's number in stage 0's stream buffer (progressive_fast.hip), blob order
};

}  // namespace mj

namespace mj {
// the first scans and the refining AC scans (progressive_fast.hip) read the stage-0 stream (destuff.hip) of the scans' segments — segment
// number DevProgSeg::stream_slot — and LUTs of kProgLutBits bits, (len << 8 | symbol) per entry, for every table
constexpr int kProgLutBits = 11;
hipError_t launch_progressive_fast(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevProgSeg *segs,
                                   int n_segs, const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                   const uint16_t *lut11p, int16_t *coef, int32_t *status, int spec_refine, int transposed,
                                   DevProgState *states, int step, int rows_per_band, int n_split = 0, DevProgSub *subs = nullptr, int parts = 1);
// The first AC scans of large progressive batches, cut into chunks (progressive_chunks.hip): one restart segment of such a scan
struct DevAcSeg {
    int32_t image;                // the image (status, geometry)
    int32_t comp;                 // the scan's component
    int32_t ss, se, al;
    int32_t table;                // index of the scan's AC table in the batch
    int32_t stream_slot;          // the segment's number in stage 0's stream buffer (seg_bits index)
    int32_t stream_dw;            // its first dword there: (begin >> 2) + stream_slot
    int32_t first_blk, n_blk;     // blocks of the scan (its own raster): the segment's first, how many
    int32_t mcu_count_h;          // blocks per row of the scan
    int32_t last;                 // the scan's last restart segment
    int32_t chunk0, n_chunks;     // its chunks in the (padded) chunk list
};
constexpr int kProgCanonBytes = 320;     // per table behind its 9-bit LUT (512 x uint16: len << 8 | symbol): uint16 limit[16], int16 base[16], uint8 vals[256]
hipError_t launch_progressive_chunks(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevAcSeg *segs, int n_segs,
                                     const uint8_t *tabs, const uint16_t *lut11p, const DevChunk *chunks, int64_t n_chunks, int cbits, uint64_t *exit_state,
                                     DevChunkOut *outs, void *items, int32_t *n_items, int32_t *owner, DevVSeg *vsegs, const DevImage *images,
                                     int16_t *coef, int32_t *status, int transposed, int max_links);
hipError_t launch_progressive_scan(hipStream_t stream, const uint8_t *blob, const DevProgSeg *segs, int n_segs,
                                   const DevProgScan *scans, const DevImage *images, const DevHuff *huff,
                                   int16_t *coef, int32_t *status, int spec_refine, int transposed, DevProgState *states,
                                   int step, int rows_per_band);
}

// stage-1 / stage-2 launchers (defined in huffman.hip / reconstruct.hip)
namespace mj {
hipError_t launch_huffman(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                          const DevImage *images, const DevHuff *huff, int16_t *coef, int32_t *status,
                          int lut_slots, int transposed);

#ifdef MJ_DIAGNOSTIC
void dbg_lanes_report();     // huffman_lanes.hip: prints and clears the in-loop stamps of MJ_DEBUG_STAGE1=3
void dbg_prog_report();      // progressive_fast.hip: ... of the refining walk
void dbg_prog_waves_report();       // progressive_fast.hip: when the waves of band launch MJ_DEBUG_PROG_STEP finished
void dbg_lanes13_waves_report();   // huffman_lanes13.hip: when every wave of the last launch finished
void dbg_fused_clear(uint8_t *dump);                   // fused.hip (MJ_DEBUG_FUSED): per-workgroup time stamps of the next launch ...
void dbg_fused_report(const uint8_t *dump, int n_wg);  // ... and their averages
#endif
// lane-parallel form: one restart segment per lane, all of the batch's tables (<= kMaxLaneTables) in LDS
constexpr int kLaneLutBits = 11;
constexpr int kMaxLaneTables = 8;
// per-workgroup table lists (batches with more tables than that): entries per workgroup; 8 or 16 of them are used
constexpr int kMaxWgTables = 16;
// stage 0 (destuff.hip): per restart segment, the bytes the bit reader keeps, as big-endian dwords at dword
// (begin >> 2) + segment index of `out_stream`; seg_bits[i] = 8 x kept bytes
hipError_t launch_destuff(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                          uint32_t *out_stream, int32_t *seg_bits);
hipError_t launch_scan_markers(hipStream_t stream, const uint8_t *blob, const DevScanJob *jobs, int n_jobs, DevSegment *segs,
                               int32_t *status);
// one counting round: entry == nullptr -> speculative round (every chunk assumes a block starts at its first bit);
// otherwise chunk c starts from entry[c - 1].  *changed counts the chunks whose exit state differs from prev_exit[c].
hipError_t launch_sync_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                             const DevImage *images, const DevHuff *huff, const uint16_t *lut11u, int n_huff,
                             const DevChunk *chunks, int64_t n_chunks, int cbits, const uint64_t *entry, uint64_t *exit_out,
                             DevChunkOut *outs, int32_t *changed, const int32_t *wg_tabs = nullptr, int wg_slots = 0, const int32_t *prev_changed = nullptr,
                             int warm_bits = -1);   // wg_tabs: per 256 chunks; warm_bits: run-up in front of every chunk, -1 = half a chunk
// the same on resolved tables (lutc: n_tabs tables of tab_bytes each, wbits index bits: plan_create.hip's build_count_tables): the first
// walk over every chunk, then — max_links > 0 — the list of chunks whose entry state was guessed wrong (items: 16 bytes per
// chunk, *n_items) and their repair, each lane walking on for at most max_links chunks (owner: 4 bytes per chunk of scratch).
// exit_state / outs as above.
hipError_t launch_count(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs, const DevImage *images,
                        const uint32_t *lutc, int tab_bytes, int n_tabs, int wbits, const DevChunk *chunks, int64_t n_chunks, int cbits,
                        int warm_bits, uint64_t *exit_state, DevChunkOut *outs, void *items, int32_t *n_items, int max_links, int32_t *owner);
// seg_chunk0[s] = the first chunk of restart segment s (n_segs + 1 entries)
hipError_t launch_build_vsegs(hipStream_t stream, const DevChunk *chunks, const int32_t *seg_chunk0, int64_t n_segs, const DevChunkOut *outs,
                              const DevSegment *segs, const int32_t *seg_bits, const DevImage *images, DevVSeg *vsegs,
                              const uint64_t *final_exit, int cbits, int32_t *status);
hipError_t launch_destuff_pieces(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, const DevPiece *pieces,
                                 int64_t n_pieces, int32_t *kept, uint32_t *out_stream, int32_t *seg_bits);
// lut11: (len << 8 | symbol) for tables used as DC tables, (len << 11 | run << 4 | size, EOB = run 64) for AC tables
hipError_t launch_huffman_lanes(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits,
                                const DevSegment *segs, int64_t n_segs,
                                const DevImage *images, const DevHuff *huff, const uint16_t *lut11, int n_huff,
                                int16_t *coef, int32_t *status, int transposed, const DevVSeg *vsegs = nullptr,
                                const int32_t *wg_tabs = nullptr, int wg_slots = 0);   // wg_tabs: [workgroups][kMaxWgTables]
// the same with resolved 13-bit AC tables (huffman_lanes13.hip); lut13: [n_ac][kLanes13SlotBytes / 4]: 8192 finished symbols + second-level tables for codes of 14..16 bits, lut11 supplies the DC
// tables; byte t of ac_slot_pk / dc_slot_pk = LDS slot of table t in that role, byte s of dc_tab_pk = table of DC slot s
constexpr int kLanes13SubTables = 144;                              // second-level tables per AC table (8 entries each)
constexpr int kLanes13SlotBytes = 8192 * 4 + kLanes13SubTables * 32;  // one AC table in that form
bool lanes13_fits(int n_ac, int n_dc);
hipError_t launch_huffman_lanes13(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs, int64_t n_segs,
                                  const DevImage *images, const DevHuff *huff, const uint16_t *lut11, const uint32_t *lut13,
                                  int n_ac, int n_dc, uint64_t ac_slot_pk, uint64_t dc_slot_pk, uint64_t dc_tab_pk,
                                  int16_t *coef, int32_t *status, int transposed, const DevVSeg *vsegs,
                                  const int32_t *by_length = nullptr, int order_mode = 0);   // by_length: segment numbers, longest first
// how that launch groups its units of work: lanes per wavefront (a workgroup = 4 waves = 4 x this many consecutive units)
int lanes_per_wave(int64_t n_segs, int n_slots);

// Test and tuning switches (mj_set_option, include/mijpeg.h): process-wide, set through the API only — the product library does
// not look at the environment for them (a stray variable must not change how a production decode runs); the diagnostic build
// (make DIAG=1) falls back to an environment variable of the same name so that the probe scripts can flip them per run.
// Returns the value or nullptr.
const char *opt(const char *name);
int set_opt(const char *name, const char *value);
int get_opt(const char *name, char *out, int cap);

constexpr size_t kStage2DumpBytes = 4096 * 1024;    // 1 KiB per workgroup of the largest persistent grid
// (MI355X: at most 768 workgroups use their KiB; two small areas further up have other uses)
constexpr size_t kDumpZeroLine = 0x180000;          // 16 bytes that stay zero (variant builds -DMJ_X_SPARSE_LD read them)
constexpr size_t kDumpClockWords = 0x1C0000;        // 4 x u64: shader-clock / 100 MHz stamps of the latest fused launch's workgroup 0 (mj_context_launch_clock)
struct ReconArgs {
    const DevImage *images;
    int32_t n_images;
    const int64_t *mcu_prefix;   // [n_images + 1]
    int64_t total_mcus;
    const int16_t *coef;
    const uint16_t *qt;          // [n_qt][64] natural order [v][u] (same layout as the coefficient blocks;
                                 // [u][v] like them when the plan is transposed)
    const double *idct_tt;       // [64 (u*8+v)][64 (x*8+y)] transposed reference table
    const uint32_t *up_taps;     // packed upsample taps, see reconstruct.hip
    uint8_t *rgb;
    uint8_t *dump;               // kStage2DumpBytes of writable memory: where stage 2's store instructions send the 16-byte
                                 // pieces that must not land in the image (lanes past an edge, strips that take the
                                 // slow stores) — the store phase stays branch-free, see reconstruct_fast.hip
    int16_t *planes;             // optional
    int16_t *idct_out;           // optional
    int32_t layout;
    int32_t exact_only;
    int32_t debug;               // diagnostic builds only (make DIAG=1, MJ_DEBUG_STAGE2 env): 1 = no IDCT rounds, 2 = no
                                 // pixel phase, 3 = no stores, 4 = in-kernel clock probe
    int32_t debug_mask;          // diagnostic builds only (MJ_DEBUG_MASK): ablations that combine — 1 no IDCT rounds (strip zeroed once),
                                 // 2 no pixel arithmetic, 4 no staging / stores, 8 stores to the dump line, 16 no level 2, 32 no coefficient loads
    // homogeneous-batch shortcut: all images share one geometry
    int32_t uniform_geometry;
    int32_t mcus_per_image;
    // fast form: the launch's strips are handed out in jobs of at most `chunk_strips` strips of one MCU column, one wavefront at
    // a time, through this ticket counter (zero before the launch; the wave that draws the launch's last ticket puts it back to zero)
    uint32_t *work_counter;
    int32_t chunk_strips;
    int32_t jobs_per_ticket;     // >= 1
    unsigned long long *level_counts;   // seam-output launches: blocks seen / sent to level 2 / sent to level 3 (mj_plan_idct_levels), per plan
};
hipError_t launch_reconstruct(hipStream_t stream, const ReconArgs &a, int hmax, int vmax, int ncomp);
// any sampling factors 1..4 per component (DevImage::generic): reconstruct.hip
hipError_t launch_reconstruct_generic(hipStream_t stream, const ReconArgs &a);
// fast form: strips of fast_tile_mcus() MCUs in column-major MCU order.  transposed = the kernel runs on the transposed
// image (blocks and tables stored [u][v]), so its x-major output is the row-major image (MJ_LAYOUT_ROWMAJOR)
int fast_tile_mcus(int hmax, int vmax, int ncomp, bool transposed);
// how many times the 64 lanes' MCUs a strip is tall (FGeo::SV): MCUs of 8 pixel rows whose column runs would be under 192 bytes
constexpr int strip_sv(int mw, int mh, int nc) { return mh != 8 ? 1 : ((64 / mw) * 8 * nc >= 192 ? 1 : ((64 / mw) * 8 * nc == 64 ? 4 : 2)); }
// jobs: pieces of at most a.chunk_strips strips of one MCU column, numbered image by image (job_prefix[i] = first job of image i)
hipError_t launch_reconstruct_fast(hipStream_t stream, const ReconArgs &a, int hmax, int vmax, int ncomp, bool transposed,
                                   const int64_t *job_prefix, int64_t total_jobs, int jobs_per_image);
// Stages 1 + 2 in one launch (fused.hip): producer wavefronts (the lane walk) and consumer wavefronts (stage 2's strip worker)
// in one workgroup per CU that takes whole images — `ipw` of them at a time, in `n_pass` passes when a workgroup's images hold
// more restart segments than its producers have lanes (8 x 64).  fused_shape() says how such a launch would be cut — ok =
// false: it does not apply (LDS) and the two launches stay.
constexpr int kFusedPassShift = 20;     // a producer's progress word: pass << 20 | MCUs of its segments complete in that pass
struct FusedShape {
    bool ok;
    int ipw;            // images per workgroup and pass
    int n_pass;         // passes of a workgroup: its "virtual workgroups" n_pass * wg + pass, one after the other
    int n_virt;         // virtual workgroups in all = ceil(images / ipw)
    int n_prod, lpw;    // producer wavefronts, lanes (restart segments) per producer wavefront
    int n_cons;         // consumer wavefronts beside the producers (every wavefront is one once the producers are through)
    int ring;           // bytes of stream window per lane
    int ac_total_bytes; // the AC tables in LDS, back to back (12-bit main levels — 13 where the plan says so — + their second-level tables)
    int dbits;          // index bits of the DC tables in LDS
    bool xwg;           // restart segments dealt out by length: one pool of jobs for the launch, hand-off across workgroups
    int n_wg;           // ... workgroups of such a launch (= CUs: all resident at once)
};
FusedShape fused_shape_x(int cus, int ac_total_bytes, int n_dc, int hmax, int vmax, bool transposed, int64_t n_segs, int want_consumers);
FusedShape fused_shape(int cus, int ac_total_bytes, int n_dc, int hmax, int vmax, bool transposed, int n_images, int spi, int want_consumers,
                       int want_producers = 0);
// spi: restart segments per image; restart_interval: MCUs per segment (any: a job's readiness is worked out per MCU)
hipError_t launch_fused(hipStream_t stream, const FusedShape &shape, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                        int64_t n_segs, const DevImage *images, const DevHuff *huff, const uint16_t *lut11, const uint32_t *lut13,
                        int n_ac, int n_dc, uint64_t ac_slot_pk, uint64_t dc_slot_pk, uint64_t dc_tab_pk, const int ac_off[4], const int ac_bits[4],
                        int16_t *coef, int32_t *status,
                        const ReconArgs &a, int hmax, int vmax, bool transposed, int spi, int restart_interval, int mcus_per_row, int mcu_rows,
                        const int64_t *job_prefix, int64_t total_jobs, int jobs_per_image, const int32_t *by_length = nullptr,
                        const int32_t *holder = nullptr, uint32_t *x_words = nullptr);
// (lut13 of launch_fused: the plan's fused tables, back to back — ac_off / ac_bits per LDS slot)
// Launch-geometry caches are per device: one process may hold contexts on several GPUs (mijpeg.h: one context per GPU per
// thread), and a function attribute set on one device says nothing about the next.  Contexts on two threads may make a
// kernel's FIRST launch at the same moment: the per-device flags are std::once_flag (OncePerDevice), the cached integers atomics.
constexpr int kMaxDevices = 64;
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < kMaxDevices ? d : 0; }
struct OncePerDevice {
    std::once_flag flag[kMaxDevices];
    template <class F> void run(F &&fn) { std::call_once(flag[current_device()], fn); }
};
struct IntPerDevice {         // 0 = not known yet; racing first uses compute and store the same value
    std::atomic<int> v[kMaxDevices];
    IntPerDevice() { for (auto &x : v) x.store(0, std::memory_order_relaxed); }
    template <class F> int get(F &&compute) {
        std::atomic<int> &x = v[current_device()];
        int r = x.load(std::memory_order_relaxed);
        if (r == 0) { r = compute(); x.store(r, std::memory_order_relaxed); }
        return r;
    }
};
inline int device_cus() {
    static IntPerDevice cus;
    return cus.get([] {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || c < 1) c = 256;
        return c;
    });
}
// MJ_LAYOUT_PLANAR_*: every image's interleaved pixels (x-major or row-major, as stage 2 wrote them) -> its three planes
hipError_t launch_planes_from_interleaved(hipStream_t stream, const DevImage *images, int n_images, int64_t max_pixels,
                                          const uint8_t *interleaved, uint8_t *planar);
// dst[0 .. bytes) = src[0 .. bytes), sixteen bytes per lane (bytes a multiple of 16): the plain copy the rooflines are held against
hipError_t launch_copy16(hipStream_t stream, const void *src, void *dst, int64_t bytes, int variant);     // variant 0 .. copy16_variants() - 1: launch shapes
int copy16_variants();
// p[0 .. n_words) = value, as a kernel (why not hipMemsetAsync: api.hip)
hipError_t launch_fill_words(hipStream_t stream, void *p, uint32_t value, int64_t n_words);
// 64-entry permutation of every block: dst[b*64 + i] = src[b*64 + table[i]]
hipError_t launch_permute_blocks(hipStream_t stream, const int16_t *src, int16_t *dst, int64_t n_blocks, int to_natural,
                                 int transposed);
}  // namespace mj
