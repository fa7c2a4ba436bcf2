// Host-side state of libmijpeg.so shared by its translation units: the device buffer cache, the context and the plan
// (api.hip: the C ABI and the launches of an execute; plan_create.hip: mj_plan_create; options.hip: the test / tuning switches).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "mijpeg_internal.h"
#include "form_select.h"

// Device buffers of destroyed plans, kept by the context for the next plan.  A decode service creates one plan per
// batch; hipMalloc / hipFree of its multi-gigabyte coefficient store every time costs more than the decode (the
// runtime hands freed memory back lazily: measured 300-700 ms stalls every few batches of 512 x 1080p), and
// hipFree waits for the device.  Sizes are rounded up to 1/8-octave steps so that batches of similar size reuse each
// other's buffers; the cache is bounded (a quarter of the device's memory, MJ_CACHE_MB overrides) and evicts the
// least recently released buffers.
struct DevBufferCache {
    struct Block { void *ptr; size_t size; uint64_t stamp; };
    std::vector<Block> free_blocks;
    std::vector<Block> live;            // handed out (size needed again at release)
    size_t cached_bytes = 0, limit_bytes = 0;
    uint64_t clock = 0;
    uint64_t n_hit = 0, n_miss = 0, n_evict = 0;    // MJ_CACHE_STATS=1 prints them when the context goes
    static size_t bucket(size_t n) {
        if (n <= 4096) return 4096;
        size_t p2 = (size_t)1 << (63 - __builtin_clzll((unsigned long long)n));
        const size_t step = p2 >> 3;
        return (n + step - 1) / step * step;
    }
    hipError_t get(void **out, size_t bytes) {
        const size_t want = bucket(bytes);
        int best = -1;
        for (int i = 0; i < (int)free_blocks.size(); ++i)
            if (free_blocks[i].size == want && (best < 0 || free_blocks[i].stamp > free_blocks[best].stamp)) best = i;
        if (best >= 0) {
            *out = free_blocks[best].ptr;
            cached_bytes -= want;
            ++n_hit;
            free_blocks.erase(free_blocks.begin() + best);
        } else {
            ++n_miss;
            hipError_t e = hipMalloc(out, want);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); trim(0); e = hipMalloc(out, want); }
            if (e != hipSuccess) return e;
        }
        live.push_back({*out, want, 0});
        return hipSuccess;
    }
    void put(void *ptr) {
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].ptr == ptr) {
                Block b = live[i];
                live[i] = live.back();
                live.pop_back();
                b.stamp = ++clock;
                free_blocks.push_back(b);
                cached_bytes += b.size;
                trim(limit_bytes);
                return;
            }
        (void)hipFree(ptr);             // not one of ours
    }
    // hand a live block back to the device instead of keeping it for the next plan (mj_plan_tune_placement's losing candidates)
    void drop(void *ptr) {
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].ptr == ptr) { live[i] = live.back(); live.pop_back(); break; }
        (void)hipFree(ptr);
    }
    void trim(size_t keep) {
        while (cached_bytes > keep && !free_blocks.empty()) {
            int old = 0;
            for (int i = 1; i < (int)free_blocks.size(); ++i)
                if (free_blocks[i].stamp < free_blocks[old].stamp) old = i;
            (void)hipFree(free_blocks[old].ptr);
            ++n_evict;
            cached_bytes -= free_blocks[old].size;
            free_blocks.erase(free_blocks.begin() + old);
        }
    }
};

struct mj_context {
    int device = 0;
    hipStream_t stream = nullptr;
    // plan creation clears a plan's big buffers here, beside whatever the context stream is running (another plan's
    // kernels, in a serving loop); the plan's first use waits for its `ready` event
    hipStream_t setup_stream = nullptr;
    // pinned staging for a plan's small uploads (descriptors, tables, segment lists: a few MB), so that they too are
    // queued on the setup stream instead of each blocking the host behind whatever the copy engine is busy with (in a
    // serving loop: the next batch's 300 MB of files).  A plan holds one arena until it is destroyed.
    struct Arena { char *base = nullptr; size_t cap = 0, used = 0; };
    std::vector<Arena> free_arenas;
    Arena *cur = nullptr;          // the arena of the plan being created
    int32_t *h_word = nullptr;     // pinned: where a stream hands one counter to the host (a pageable target would make the
                                   // copy synchronous for the whole device, i.e. wait for other plans' kernels on other streams)
    double *d_idct_tt = nullptr;   // [u*8+v][x*8+y], the reference's InverseDCT.idct_table transposed
    uint8_t *d_dump = nullptr;     // stage 2's dump lines (mj::kStage2DumpBytes)
    bool no_graph = false;         // MJ_NO_GRAPH at context creation: never replay captured graphs
    // Fused launches (fused.hip) of one context take turns: a fused workgroup wants a CU's whole LDS, so two plans' fused
    // launches on two streams cannot share the chip — left to the hardware queues they interleave workgroup by workgroup and
    // both finish later than one after the other would (three plans in flight: 6.58 ms per step against 5.74, round 5).  An
    // execute that contains a fused launch waits — its stream does, not the host — for the context's previous one.
    hipEvent_t fused_done = nullptr;
    hipStream_t fused_stream = nullptr;     // where the latest fused execute went (null: none yet)
    DevBufferCache cache;
    std::string err;
};

extern std::string g_create_err;     // what mj_last_error(NULL) returns (api.hip)

struct mj_plan {
    mj_context *ctx = nullptr;
    int32_t n_images = 0;
    int32_t layout = 0;
    uint32_t flags = 0;
    int hmax = 1, vmax = 1, ncomp = 3;
    int lut_slots = 1;
    bool uniform = false;
    // row-major plans run the fast stage 2 on the transposed problem: blocks and tables are then stored [u][v]
    bool transposed = false;
    bool generic = false;              // a sampling layout outside the common ones: wave form of stage 1, k_reconstruct_generic
    uint8_t *d_rgb_tmp = nullptr;  // planar layouts: the interleaved image stage 2 writes before the components are separated
    int64_t max_pixels = 0;        // largest width*height of the batch
    int32_t mcus_per_image = 0;
    mj_plan_info info{};
    std::vector<mj::DevImage> h_images;
    // device
    uint8_t *d_blob_owned = nullptr;
    const uint8_t *blob_src = nullptr;  // a caller's device blob that every execute copies into d_blob_owned first (wave form behind
    int64_t blob_src_len = 0;           // MJ_FLAG_GPU_SEGMENT: the kernel reads further ahead than that flag makes the caller pad)
    const uint8_t *d_blob = nullptr;
    mj::DevSegment *d_segs = nullptr;
    int64_t n_segs = 0;
    mj::DevImage *d_images = nullptr;
    mj::DevHuff *d_huff = nullptr;
    uint16_t *d_lut11 = nullptr;        // [n_huff][2048] primary LUTs of the lane-parallel stage-1 kernel
    // resolved 13-bit AC tables of the lane form's fast variant (huffman_lanes13.hip), when the batch's tables fit LDS that way
    int32_t *d_by_length = nullptr;     // restart segments, longest first (how the lane form deals them out to its waves)
    int seg_order_mode = 0;
    uint32_t *d_lut13 = nullptr;        // [n_ac13][8192]
    uint32_t *d_lut12 = nullptr;        // a fused launch's AC tables: back to back, 12-bit main levels (13 where it says so)
    int lutf_off[4] = {0, 0, 0, 0}, lutf_bits[4] = {12, 12, 12, 12}, lutf_total = 0;   // ... byte offset and index bits per LDS slot, bytes in all
    int n_ac13 = 0, n_dc13 = 0;
    uint64_t ac_slot_pk = 0, dc_slot_pk = 0, dc_tab_pk = 0;
    // batches with more tables than LDS holds (files with their own optimised tables): per workgroup, the tables its
    // units of work use — one list for the lane kernel's launch, one for the counting rounds (256 chunks per workgroup)
    int32_t *d_wg_tabs_lanes = nullptr, *d_wg_tabs_count = nullptr;
    int wg_slots_lanes = 0, wg_slots_count = 0;      // 8 or 16 LUTs per workgroup
    uint32_t *d_stream = nullptr;       // stage 0 output (destuff.hip): big-endian dwords per restart segment
    size_t stream_bytes = 0;            // ... its size (mj_plan_tune_placement tries other buffers of that size)
    int32_t *d_seg_bits = nullptr;      // [n_segs] bits per segment after stage 0
    // long restart segments (files without DRI): synchronisation passes + virtual segments (huffman_sync.hip)
    bool use_sync = false;
    int sync_rounds = 32;          // repair rounds of the synchronisation form queued per execute (MJ_SYNC_ROUNDS at plan creation: tests).
                                   // A round behind one that changed nothing returns at once, so the number only bounds the longest
                                   // chain of wrongly guessed entry states that still settles (flat image regions re-synchronise badly:
                                   // round 4 found a quarter of a synthetic batch's images unconverged after four rounds, none after six; an idle round costs ~2 us)
    int sync_chunk_bytes = 2048;
    int sync_warm_bits = -1;       // run-up in front of every chunk of the counting rounds (MJ_SYNC_WARM at plan creation; -1 = half a chunk)
    uint16_t *d_lut11u = nullptr;       // every table as len << 11 | run << 4 | size
    mj::DevChunk *d_chunks = nullptr;
    int64_t n_chunks = 0;
    uint64_t *d_stateA = nullptr, *d_stateB = nullptr;
    mj::DevChunkOut *d_couts = nullptr;
    mj::DevVSeg *d_vsegs = nullptr;
    int32_t *d_seg_chunk0 = nullptr;    // [n_segs + 1] first chunk of every restart segment
    uint32_t *d_lutc = nullptr;         // the counting walks' resolved tables (k_count), or null: the classic rounds (k_sync_count)
    int lutc_tab_bytes = 0, lutc_bits = 0;
    void *d_sync_items = nullptr;       // the repair launch's work list (16 bytes per chunk); its counter is d_changed[0]
    hipGraphExec_t graph_exec = nullptr;   // captured launches of one execute (see mj_plan_execute)
    hipStream_t graph_stream = nullptr;
    uint8_t *graph_rgb = nullptr;
    bool executed_once = false;
    mj_context::Arena arena;               // pinned staging of this plan's uploads (back to the context at destroy)
    hipEvent_t ready = nullptr;             // recorded behind the creation-time clears on the context's setup stream
    bool ready_done = false;
    hipStream_t ready_stream = nullptr;     // the stream that was made to wait for `ready`
    hipEvent_t done = nullptr;              // recorded behind the plan's latest execute: what mj_plan_sync / _read / _destroy wait for
    bool done_valid = false;
    bool last_was_graph = false;
    hipStream_t prev_stream = nullptr;      // of the last plain execute
    uint8_t *prev_rgb = nullptr;
    mj::DevPiece *d_pieces = nullptr;   // stage 0 of long segments, piece by piece
    int64_t n_pieces = 0;
    int32_t *d_piece_kept = nullptr;
    int32_t *d_changed = nullptr;
    mj::DevScanJob *d_jobs = nullptr;   // MJ_FLAG_GPU_SEGMENT: per-image byte ranges for the marker scan
    int n_jobs = 0;
    int n_huff = 0;
    bool use_lanes = false;
    // stages 1 + 2 in one launch (fused.hip) for mj_plan_execute, where the batch is of the kind it takes
    bool use_fused = false;
    mj::FusedShape fused{};
    int fused_spi = 0;                  // restart segments (= MCU rows) per image
    int32_t *d_holder = nullptr;        // fused, segments dealt out by length: per segment, the progress word of the wave that walks it
    uint32_t *d_xwords = nullptr;       // ... the launch's ticket counter (32 words: a line of its own) and the progress words
    // progressive batches: scans grouped by dependency level, one launch per level
    bool progressive = false;
    mj::DevProgScan *d_pscans = nullptr;
    mj::DevProgSeg *d_psegs = nullptr;
    mj::DevProgState *d_pstates = nullptr;   // per segment: what a scan carries from band to band
    int64_t n_psegs = 0;
    int prog_rows_per_band = 2, prog_steps = 0;
    bool prog_banded = false;
    // the first scans and the refining AC scans read the stage-0 stream of the scans' segments (progressive_fast.hip)
    mj::DevSegment *d_prog_dsegs = nullptr;  // d_psegs' byte ranges in the form stage 0 takes
    uint16_t *d_lut11p = nullptr;            // [n_huff][1 << kProgLutBits], (len << 8 | symbol)
    bool prog_fast = false;
    int64_t prog_rest_off = 0;               // banded: d_psegs[prog_rest_off..] are the segments of the scans progressive.hip walks
    mj::DevProgSub *d_psubs = nullptr;       // [n_split][2][kProgSub]: by segment, two sets (even and odd bands)
    int prog_parts = 4;                      // ... parts per band
    // the first AC scans of large batches in chunks (progressive_chunks.hip): d_psegs[n_psegs_wave..] are theirs, no wavefront walk takes them
    bool prog_chunks = false;
    int pc_chunk_bytes = 512;
    int64_t n_psegs_wave = 0;
    mj::DevAcSeg *d_acsegs = nullptr;
    int n_acsegs = 0;
    mj::DevChunk *d_pc_chunks = nullptr;
    int64_t n_pc_chunks = 0;
    uint8_t *d_pc_tabs = nullptr;
    uint64_t *d_pc_exit = nullptr;
    mj::DevChunkOut *d_pc_outs = nullptr;
    void *d_pc_items = nullptr;
    int32_t *d_pc_owner = nullptr;
    mj::DevVSeg *d_pc_vsegs = nullptr;
    int64_t n_split = 0;                     // banded: d_psegs[0..n_split) are the segments of the scans walked as scout + parts
    // (one launch per dependency level only — MJ_PROG_BANDS=0; the band pipeline orders d_psegs by length instead)
    std::vector<int64_t> ordinal_seg_off;   // [n_ordinals + 1] into d_psegs
    std::vector<int64_t> ordinal_kind_off;  // [n_ordinals][4]: within a level, where the segments of each kind of scan start
    uint16_t *d_qt = nullptr;
    int64_t *d_mcu_prefix = nullptr;
    int64_t *d_job_prefix = nullptr;    // fast stage 2: first job of every image (+ total), then the kernel's ticket counter
    int64_t total_jobs = 0;
    int32_t jobs_per_image = 0;
    int32_t chunk_strips = 16;          // strips per job (a piece of one MCU column) of the fast stage 2
    int32_t jobs_per_ticket = 1;        // consecutive jobs a wave draws at once
    int16_t *d_tmp_coef = nullptr;      // staging for zig-zag <-> natural conversion
    int16_t *d_coef = nullptr;
    uint8_t *d_rgb = nullptr;       // plan-owned, allocated on first use
    int16_t *d_planes = nullptr;
    int16_t *d_idct = nullptr;
    int32_t *d_status = nullptr;
    uint8_t *last_rgb = nullptr;    // where the most recent execute wrote
};

// ---- plan_tables.hip: table building (host)
namespace mj {
// DHT -> canonical code book + 9-bit LUT (jpeg_decoder.py:366-377)
void build_dev_huff(const mj_huff_spec &spec, DevHuff &h);
bool build_resolved_tables(const mj_batch *b, const std::vector<int> &role, uint64_t ac_pk, int n_ac, const int ab_of_slot[4], int fixed_slot_bytes,
                           std::vector<uint32_t> &out, int slot_off[4], int &total_bytes);
bool build_count_tables(const mj_batch *b, const std::vector<int> &role, int W, std::vector<uint32_t> &out, int &tab_bytes);
// ---- plan_progressive.hip: the progressive side of mj_plan_create
struct ProgScans { std::vector<DevProgScan> pscans; std::vector<DevProgSeg> psegs; };
int plan_progressive_scans(mj_context *ctx, const mj_batch *b, mj_plan *p, ProgScans &S, int64_t &entropy_bytes);
int plan_progressive_upload(mj_context *ctx, const mj_batch *b, mj_plan *p, ProgScans &S);
}  // namespace mj

// (one definition per translation unit: they return through the caller's frame)
namespace {

int fail(mj_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define MJ_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((ctx), MJ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// a host array into a device buffer from the context's cache — through the plan's pinned arena on the setup stream where it fits
template <typename T>
int upload(mj_context *ctx, T **dst, const T *src, size_t n, size_t pad_bytes = 0) {
    MJ_HIP(ctx, ctx->cache.get((void **)dst, n * sizeof(T) + pad_bytes + 16));
    if (pad_bytes) MJ_HIP(ctx, hipMemsetAsync((char *)*dst + n * sizeof(T), 0, pad_bytes, ctx->setup_stream));
    const size_t bytes = n * sizeof(T);
    if (!bytes) return MJ_OK;
    mj_context::Arena *a = ctx->cur;
    const size_t at = a ? (a->used + 63) & ~(size_t)63 : 0;
    if (a && at + bytes <= a->cap) {
        memcpy(a->base + at, src, bytes);
        a->used = at + bytes;
        MJ_HIP(ctx, hipMemcpyAsync(*dst, a->base + at, bytes, hipMemcpyHostToDevice, ctx->setup_stream));
    } else {
        MJ_HIP(ctx, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));      // big (a host blob) or no arena: the plain way
    }
    return MJ_OK;
}

}  // namespace
