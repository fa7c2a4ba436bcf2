// C ABI of libmijpeg.so: context, plan (upload once / execute many), one-shot helpers.
// Host-side only; the kernels are in huffman.hip and reconstruct.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "mijpeg_internal.h"

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
    DevBufferCache cache;
    std::string err;
};

static std::string g_create_err;

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
    uint32_t *d_lut12 = nullptr;        // the same with a 12-bit main level, [n_ac13][lut12_slot_bytes / 4] (fused launches)
    int lut12_slot_bytes = 0;
    int n_ac13 = 0, n_dc13 = 0;
    uint64_t ac_slot_pk = 0, dc_slot_pk = 0, dc_tab_pk = 0;
    // batches with more tables than LDS holds (files with their own optimised tables): per workgroup, the tables its
    // units of work use — one list for the lane kernel's launch, one for the counting rounds (256 chunks per workgroup)
    int32_t *d_wg_tabs_lanes = nullptr, *d_wg_tabs_count = nullptr;
    int wg_slots_lanes = 0, wg_slots_count = 0;      // 8 or 16 LUTs per workgroup
    uint32_t *d_stream = nullptr;       // stage 0 output (destuff.hip): big-endian dwords per restart segment
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

// InverseDCT.idct_table (jpeg_decoder.py:1541-1553): 0.25*Cu*Cv*cos((2x+1)*pi*u/16)*cos((2y+1)*pi*v/16),
// left to right in IEEE doubles with libm cos (= Python's math.cos).  Stored transposed: [u*8+v][x*8+y].
void build_idct_tt(double *tt) {
    const double pi = 3.141592653589793;
    const double isq2 = pow(2.0, -0.5);
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++)
            for (int u = 0; u < 8; u++)
                for (int v = 0; v < 8; v++) {
                    volatile double t = 0.25 * (u == 0 ? isq2 : 1.0);
                    t = t * (v == 0 ? isq2 : 1.0);
                    volatile double ca = cos(((double)(2 * x + 1) * pi) * (double)u / 16.0);
                    volatile double cb = cos(((double)(2 * y + 1) * pi) * (double)v / 16.0);
                    t = t * ca;
                    t = t * cb;
                    tt[(u * 8 + v) * 64 + (x * 8 + y)] = t;
                }
}

// DHT -> canonical code book + 9-bit LUT (jpeg_decoder.py:366-377)
void build_dev_huff(const mj_huff_spec &spec, mj::DevHuff &h) {
    memset(&h, 0, sizeof(h));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        code <<= 1;
        h.first_code[l] = code;
        h.count[l] = spec.bits[l - 1];
        h.first_sym[l] = k;
        for (int i = 0; i < spec.bits[l - 1] && k < 256; ++i, ++k, ++code) {
            h.vals[k] = spec.vals[k];
            if (l <= mj::kLutBits && code < (1 << l)) {
                int shift = mj::kLutBits - l;
                for (int f = 0; f < (1 << shift); ++f) {
                    int idx = (code << shift) | f;
                    if (h.lut[idx] == 0) h.lut[idx] = (uint16_t)((l << 8) | spec.vals[k]);   // first (shortest) key wins
                }
            }
        }
    }
}

// zig-zag index -> natural index v*8+u (row = vertical frequency); blocks and quantisation tables live on the
// device in this order (see huffman.hip / reconstruct_fast.hip)
const uint8_t kNatOfZz[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool sampling_class(const mj_image_desc &d, int &hmax, int &vmax) {
    if (d.ncomp == 1) { hmax = vmax = 1; return true; }
    if (d.ncomp != 3) return false;
    if (d.hs[1] != 1 || d.vs[1] != 1 || d.hs[2] != 1 || d.vs[2] != 1) return false;
    hmax = d.hs[0]; vmax = d.vs[0];
    return ((hmax == 1 || hmax == 2) && (vmax == 1 || vmax == 2)) || (hmax == 4 && vmax == 1);
}

// Any other three-component layout with factors 1..4 (4:1:0, 1x4, factors of 3, chroma above 1x1, luma below the chroma
// resolution ...): decoded by the wave form of stage 1 and k_reconstruct_generic.  The reference takes them all (:205-240).
bool generic_sampling(const mj_image_desc &d, int &hmax, int &vmax) {
    if (d.ncomp != 3) return false;
    hmax = vmax = 1;
    int blocks = 0;
    for (int c = 0; c < 3; ++c) {
        if (d.hs[c] < 1 || d.hs[c] > 4 || d.vs[c] < 1 || d.vs[c] > 4) return false;
        hmax = std::max(hmax, (int)d.hs[c]); vmax = std::max(vmax, (int)d.vs[c]);
        blocks += d.hs[c] * d.vs[c];
    }
    return blocks <= mj::kMaxBlocksPerMcu;
}

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


// Resolved AC tables (huffman_lanes13.hip's entry format) with AB index bits for every table of the batch used as an AC table:
// per table a main level of 2^AB entries — the FINISHED symbol wherever code + value bits fit the index (jpeg_decoder.py:834-866 and
// bin_twos_complement :1636-1646 evaluated here), else what the arithmetic step needs — and second-level tables of 2^(16 - AB)
// entries for the prefixes of longer codes.  fixed_slot_bytes: the stride of a table in `out` (0 = as small as the batch's codes
// allow: main level + the largest number of second-level tables any table needs; slot_bytes returns it).  false: does not fit.
bool build_resolved_tables(const mj_batch *b, const std::vector<int> &role, uint64_t ac_pk, int n_ac, int AB, int fixed_slot_bytes,
                           std::vector<uint32_t> &out, int &slot_bytes) {
    const int AS = 1 << AB, SUB = 1 << (16 - AB);
    const int res_limit = AB;
    int max_sub = 1;
    if (fixed_slot_bytes) {
        max_sub = (fixed_slot_bytes / 4 - AS) / SUB;
    } else {
        for (int t = 0; t < b->n_huff; ++t) {
            if (role[t] != 2) continue;
            std::vector<char> seen(AS, 0);
            int n = 1, code = 0, k = 0;
            for (int l = 1; l <= 16; ++l) {
                code <<= 1;
                for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                    if (code >= (1 << l) || l <= AB) continue;
                    const int prefix = code >> (l - AB);
                    if (!seen[prefix]) { seen[prefix] = 1; ++n; }
                }
            }
            max_sub = std::max(max_sub, n);
        }
    }
    const int SLOT = fixed_slot_bytes ? fixed_slot_bytes / 4 : ((AS + max_sub * SUB) * 4 + 15) / 16 * 4;
    if ((size_t)SLOT * 4 > 65535u) return false;                // (second-level tables are addressed by a 16-bit byte offset)
    slot_bytes = SLOT * 4;
    out.assign((size_t)n_ac * SLOT, 0xFFFFFFFFu);
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 2) continue;
        uint32_t *tab = out.data() + (size_t)((ac_pk >> (8 * t)) & 0xFF) * SLOT;
        // second-level tables behind the main one: for the 16 - AB bits that follow an AB-bit prefix of longer codes;
        // table 0 = "no such code" (where every other unset main entry points as well)
        int n_sub = 1;
        for (int i = 0; i < SUB; ++i) tab[AS + i] = 0x8000u;
        int code = 0, k = 0;
        auto put = [&](uint32_t *base, uint32_t first, uint32_t count, uint32_t entry) {     // the shortest code wins (first fit)
            for (uint32_t f = 0; f < count; ++f)
                if (base[first + f] == 0xFFFFFFFFu) base[first + f] = entry;
        };
        for (int l = 1; l <= 16; ++l) {
            code <<= 1;
            for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                if (code >= (1 << l)) continue;
                const int hv = b->huff[t].vals[k], run = hv >> 4, size = hv & 15;
                const uint32_t adv = hv == 0 ? 127u : 2u * (uint32_t)(run + 1);
                const uint32_t open_entry = ((uint32_t)(31 - size) << 24) | ((uint32_t)l << 16) | 0x8000u | ((hv == 0 ? 0u : (uint32_t)(run + 1)) << 8);   // value bits taken arithmetically
                if (l > AB) {
                    const uint32_t prefix = (uint32_t)code >> (l - AB);
                    uint32_t &m = tab[prefix];
                    if (m == 0xFFFFFFFFu) {                       // first long code under this prefix: a new table
                        if (n_sub >= max_sub) return false;
                        for (int j = 0; j < SUB; ++j) tab[AS + n_sub * SUB + j] = 0xFFFFFFFFu;
                        m = ((uint32_t)(AS * 4 + n_sub * SUB * 4) << 16) | 0xC000u;
                        ++n_sub;
                    }
                    if ((m & 0xC0FFu) != 0xC000u) continue;       // a shorter code owns the prefix (over-subscribed table)
                    uint32_t *sub = tab + ((m >> 16) / 4);
                    put(sub, ((uint32_t)code << (16 - l)) & (uint32_t)(SUB - 1), 1u << (16 - l), open_entry);
                } else if (hv == 0 ? l <= res_limit : l + size <= res_limit) {
                    const int n = hv == 0 ? 0 : size, rest = AB - l - n;
                    for (uint32_t vb = 0; vb < (1u << n); ++vb) {
                        // bin_twos_complement (:1636-1646): leading 1 = the value itself, leading 0 = value - (2^n - 1)
                        const int val = n == 0 ? 0 : ((vb >> (n - 1)) ? (int)vb : (int)vb - ((1 << n) - 1));
                        put(tab, (((uint32_t)code << n) | vb) << rest, 1u << rest,
                            ((uint32_t)(uint16_t)(int16_t)val << 16) | (adv << 8) | (uint32_t)(l + n));
                    }
                } else {
                    put(tab, (uint32_t)code << (AB - l), 1u << (AB - l), open_entry);
                }
            }
        }
        for (int i = 0; i < AS; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = ((uint32_t)(AS * 4) << 16) | 0xC000u;          // no such code: the empty second-level table
        for (int i = AS; i < SLOT; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x8000u;
    }
    return true;
}

}  // namespace

namespace mj {
namespace {
// name, and what a value must look like: one of `words` (separated by '|'), or an integer in [lo, hi] (a multiple of `step`)
struct OptionRule { const char *name; const char *words; int lo, hi, step; };
const OptionRule kOptionRules[] = {
    {"MJ_HUFFMAN", "wave|lanes|lanes11|sync", 0, 0, 1}, {"MJ_SEG_ORDER", "blob|binned|striped", 0, 0, 1},
    {"MJ_SYNC_ROUNDS", nullptr, 0, 64, 1},   {"MJ_SYNC_CHUNK", nullptr, 256, 65536, 4}, {"MJ_SYNC_WARM", nullptr, 0, 65536, 1},
    {"MJ_PROG_BANDS", nullptr, 0, 1, 1},     {"MJ_PROG_ROWS", nullptr, 1, 4096, 1},     {"MJ_PROG_FAST", nullptr, 0, 1, 1},
    {"MJ_LANES_WAVES", nullptr, 1, 16, 1},   {"MJ_LANES_PER_WAVE", nullptr, 1, 64, 1},  {"MJ_LANES_RING", "64|128", 0, 0, 1},
    {"MJ_STAGE2_CHUNK", nullptr, 1, 4096, 1}, {"MJ_PROG_SPLIT", nullptr, 0, 2, 1},      {"MJ_PROG_PARTS", nullptr, 1, kProgSub, 1},
    {"MJ_FUSED", nullptr, 0, 1, 1},          {"MJ_FUSED_CONSUMERS", nullptr, 0, 8, 1},
};
constexpr int kNumOptions = (int)(sizeof(kOptionRules) / sizeof(kOptionRules[0]));
struct OptionTable {
    std::mutex mu;
    struct Entry { std::string value; bool set = false; };
    Entry e[kNumOptions];
};
OptionTable g_options;
bool option_value_ok(const OptionRule &r, const char *v) {
    if (r.words) {
        const size_t n = strlen(v);
        for (const char *w = r.words; *w;) {
            const char *bar = strchr(w, '|');
            const size_t len = bar ? (size_t)(bar - w) : strlen(w);
            if (len == n && !strncmp(w, v, n)) return true;
            w += len + (bar ? 1 : 0);
        }
        return false;
    }
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    return end != v && *end == 0 && x >= r.lo && x <= r.hi && x % r.step == 0;
}
}  // namespace
// The value is COPIED under the lock (another thread may set the option while this one parses it) into a small per-thread
// ring: the pointer stays good for this thread's next seven opt() calls — every caller consumes it on the spot.
const char *opt(const char *name) {
    thread_local std::string ring[8];
    thread_local unsigned turn = 0;
    {
        std::lock_guard<std::mutex> lk(g_options.mu);
        for (int i = 0; i < kNumOptions; ++i)
            if (!strcmp(kOptionRules[i].name, name)) {
                if (!g_options.e[i].set) break;
                std::string &slot = ring[turn++ & 7u];
                slot = g_options.e[i].value;
                return slot.c_str();
            }
    }
#ifdef MJ_DIAGNOSTIC
    return getenv(name);
#else
    return nullptr;
#endif
}
int get_opt(const char *name, char *out, int cap) {
    std::lock_guard<std::mutex> lk(g_options.mu);
    for (int i = 0; i < kNumOptions; ++i)
        if (!strcmp(kOptionRules[i].name, name)) {
            if (cap > 0) { strncpy(out, g_options.e[i].set ? g_options.e[i].value.c_str() : "", (size_t)cap - 1); out[cap - 1] = 0; }
            return MJ_OK;
        }
    return MJ_ERR_INVALID;
}
int set_opt(const char *name, const char *value) {
    std::lock_guard<std::mutex> lk(g_options.mu);
    for (int i = 0; i < kNumOptions; ++i)
        if (!strcmp(kOptionRules[i].name, name)) {
            const bool set = value != nullptr && value[0] != 0;
            if (set && !option_value_ok(kOptionRules[i], value)) return MJ_ERR_INVALID;     // a sweep must not time the default under another label
            g_options.e[i].set = set;
            g_options.e[i].value = set ? value : "";
            return MJ_OK;
        }
    return MJ_ERR_INVALID;
}
__constant__ uint8_t c_nat[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__global__ void k_permute_blocks(const int16_t *__restrict__ src, int16_t *__restrict__ dst, int64_t n_blocks,
                                 int to_natural, int tr) {
    const int lane = threadIdx.x & 63;
    const int n0 = c_nat[lane], pos = tr ? ((n0 & 7) << 3 | n0 >> 3) : n0;   // store position of zig-zag index `lane`
    for (int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < n_blocks;
         b += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        if (to_natural) dst[b * 64 + pos] = src[b * 64 + lane];
        else dst[b * 64 + lane] = src[b * 64 + pos];
    }
}
// Small clears inside an execute (statuses, counters, the synchronisation form's records) are KERNELS, not hipMemsetAsync:
// a re-executed plan replays a captured graph, and a memset node of a size that is no multiple of 16 bytes (1021 statuses)
// was seen to write the byte value of an unrelated hipMemset issued between two replays (ROCm 7.0 runtime, MI355X; found
// with the test hook that poisons the coefficient store: tests/test_gpu_parity.py::_decode_plan).  A kernel node carries
// its value in its own arguments.
__global__ void k_fill_words(uint32_t *__restrict__ p, uint32_t value, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}
hipError_t launch_fill_words(hipStream_t stream, void *p, uint32_t value, int64_t n_words) {
    if (n_words <= 0) return hipSuccess;
    const int64_t want = (n_words + 255) / 256;
    hipLaunchKernelGGL(k_fill_words, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, stream, static_cast<uint32_t *>(p), value, n_words);
    return hipGetLastError();
}
hipError_t launch_permute_blocks(hipStream_t stream, const int16_t *src, int16_t *dst, int64_t n_blocks, int to_natural,
                                 int transposed) {
    if (n_blocks == 0) return hipSuccess;
    int64_t want = (n_blocks + 3) / 4;
    unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_permute_blocks, dim3(blocks), dim3(256), 0, stream, src, dst, n_blocks, to_natural, transposed);
    return hipGetLastError();
}
}  // namespace mj

extern "C" {

int mj_version(void) { return MJ_VERSION; }

const char *mj_last_error(const mj_context *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mj_create(int device_id, mj_context **out) {
    if (!out) return fail(nullptr, MJ_ERR_INVALID, "mj_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, MJ_ERR_HIP, "mj_create: no HIP device available (%s); libmijpeg has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n) return fail(nullptr, MJ_ERR_INVALID, "mj_create: device %d of %d", device_id, n);
    mj_context *ctx = new mj_context();
    ctx->device = device_id;
    MJ_HIP(nullptr, hipSetDevice(device_id));
    MJ_HIP(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    MJ_HIP(nullptr, hipStreamCreateWithFlags(&ctx->setup_stream, hipStreamNonBlocking));
    MJ_HIP(nullptr, hipHostMalloc((void **)&ctx->h_word, 64, hipHostMallocDefault));
    std::vector<double> tt(64 * 64);
    build_idct_tt(tt.data());
    MJ_HIP(nullptr, hipMalloc((void **)&ctx->d_idct_tt, tt.size() * sizeof(double)));
    MJ_HIP(nullptr, hipMemcpy(ctx->d_idct_tt, tt.data(), tt.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->no_graph = getenv("MJ_NO_GRAPH") != nullptr;
    MJ_HIP(nullptr, hipMalloc((void **)&ctx->d_dump, mj::kStage2DumpBytes));
    MJ_HIP(nullptr, hipMemset(ctx->d_dump, 0, mj::kStage2DumpBytes));
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
        ctx->cache.limit_bytes = total_b / 4;
        if (const char *e = getenv("MJ_CACHE_MB")) ctx->cache.limit_bytes = (size_t)atoll(e) << 20;
    }
    *out = ctx;
    return MJ_OK;
}

void mj_destroy(mj_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->setup_stream) { (void)hipStreamSynchronize(ctx->setup_stream); (void)hipStreamDestroy(ctx->setup_stream); }
    if (getenv("MJ_CACHE_STATS"))
        fprintf(stderr, "[mijpeg] device buffer cache: %llu hits, %llu misses (hipMalloc), %llu evictions, %.1f MB cached at the end\n",
                (unsigned long long)ctx->cache.n_hit, (unsigned long long)ctx->cache.n_miss, (unsigned long long)ctx->cache.n_evict, ctx->cache.cached_bytes / 1048576.0);
    ctx->cache.trim(0);
    for (auto &a : ctx->free_arenas) (void)hipHostFree(a.base);
    if (ctx->h_word) (void)hipHostFree(ctx->h_word);
    if (ctx->d_idct_tt) (void)hipFree(ctx->d_idct_tt);
    if (ctx->d_dump) (void)hipFree(ctx->d_dump);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mj_context_wait_event(mj_context *ctx, void *hip_event) {
    if (!ctx || !hip_event) return MJ_ERR_INVALID;
    MJ_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return MJ_OK;
}

/* Host-side table export for tests: the IDCT table exactly as the library builds it ([u*8+v][x*8+y]). */
void mj_host_idct_table(double *tt) { build_idct_tt(tt); }

void mj_plan_destroy(mj_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    // the buffers go back to the context for the next plan: nothing of this plan may still be running on them (other
    // plans' work on the same streams is none of its business: a serving loop destroys batch k while batch k+1 runs)
    if (p->ready) { (void)hipEventSynchronize(p->ready); (void)hipEventDestroy(p->ready); }
    if (p->done) { if (p->done_valid) (void)hipEventSynchronize(p->done); (void)hipEventDestroy(p->done); }
    if (p->arena.base) {
        if (p->ctx->free_arenas.size() < 4) p->ctx->free_arenas.push_back(p->arena);
        else (void)hipHostFree(p->arena.base);
    }
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    void *ptrs[] = {p->d_blob_owned, p->d_segs, p->d_images, p->d_huff, p->d_lut11, p->d_lut13, p->d_lut12, p->d_by_length, p->d_wg_tabs_lanes, p->d_wg_tabs_count, p->d_stream, p->d_seg_bits, p->d_jobs, p->d_lut11u, p->d_chunks, p->d_stateA, p->d_stateB, p->d_couts, p->d_vsegs, p->d_changed, p->d_pieces, p->d_piece_kept, p->d_pscans, p->d_psegs, p->d_pstates, p->d_psubs, p->d_prog_dsegs, p->d_lut11p, p->d_qt, p->d_mcu_prefix, p->d_job_prefix, p->d_tmp_coef, p->d_coef,
                    p->d_rgb, p->d_rgb_tmp, p->d_planes, p->d_idct, p->d_status};
    for (void *q : ptrs)
        if (q) p->ctx->cache.put(q);
    delete p;
}

int mj_set_option(const char *name, const char *value) {
    if (!name) return MJ_ERR_INVALID;
    return mj::set_opt(name, value);
}

int mj_get_option(const char *name, char *value_out, int32_t cap) {
    if (!name || (cap > 0 && !value_out)) return MJ_ERR_INVALID;
    return mj::get_opt(name, value_out, cap);
}

int mj_plan_create(mj_context *ctx, const mj_batch *b, mj_plan **out) {
    if (!ctx) return MJ_ERR_INVALID;
    if (!b || !out) return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: NULL argument");
    *out = nullptr;
    if (b->n_images <= 0 || !b->images) return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: empty batch");
    if (b->layout < MJ_LAYOUT_XMAJOR || b->layout > MJ_LAYOUT_PLANAR_ROWMAJOR)
        return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: unknown layout %d", b->layout);
    if (b->n_qt <= 0 || !b->qt) return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: no quantisation tables");
    MJ_HIP(ctx, hipSetDevice(ctx->device));

    mj_plan *p = new mj_plan();
    p->ctx = ctx;
    p->n_images = b->n_images;
    p->layout = b->layout;
    p->flags = b->flags;
    for (int i = 0; i < b->n_images; ++i) { int h_, v_; if (!sampling_class(b->images[i], h_, v_)) p->generic = true; }
    // (the generic stage 2, like the exact-order one, writes either orientation itself: no transposed store)
    p->transposed = (b->layout & 1) == MJ_LAYOUT_ROWMAJOR && !(p->flags & MJ_FLAG_EXACT_ONLY) && !p->generic;
    struct Guard { mj_plan *p; mj_context *c; ~Guard() { c->cur = nullptr; if (p) mj_plan_destroy(p); } } guard{p, ctx};
    if (!ctx->free_arenas.empty()) { p->arena = ctx->free_arenas.back(); ctx->free_arenas.pop_back(); }
    else if (hipHostMalloc((void **)&p->arena.base, (size_t)8 << 20, hipHostMallocDefault) == hipSuccess) p->arena.cap = (size_t)8 << 20;
    else { (void)hipGetLastError(); p->arena = mj_context::Arena{}; }
    p->arena.used = 0;
    ctx->cur = p->arena.base ? &p->arena : nullptr;

    const bool have_entropy = b->blob_mem != MJ_MEM_NONE && b->blob != nullptr;
    const bool prog = have_entropy && b->n_scans > 0;
    p->progressive = prog;
    if (prog && !b->scans) return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: n_scans > 0 without scans");
    if (have_entropy && (b->n_huff <= 0 || !b->huff || !b->seg_begin || !b->seg_end))
        return fail(ctx, MJ_ERR_INVALID, "mj_plan_create: entropy data without Huffman tables / segment offsets");

    std::vector<mj::DevImage> &imgs = p->h_images;
    std::vector<mj::DevSegment> segs;
    std::vector<mj::DevScanJob> jobs;      // MJ_FLAG_GPU_SEGMENT: one marker-scan job per image
    std::vector<int64_t> mcu_prefix(b->n_images + 1, 0);
    imgs.resize(b->n_images);
    int64_t blk = 0, mcu = 0, rgb = 0, pix = 0, ent = 0;
    p->uniform = true;
    p->lut_slots = 1;
    for (int i = 0; i < b->n_images; ++i) {
        const mj_image_desc &d = b->images[i];
        mj::DevImage &im = imgs[i];
        memset(&im, 0, sizeof(im));
        int hmax, vmax;
        if (d.width <= 0 || d.height <= 0 || d.width > 65535 || d.height > 65535)
            return fail(ctx, MJ_ERR_INVALID, "image %d: bad dimensions %dx%d", i, d.width, d.height);
        const bool common = sampling_class(d, hmax, vmax);
        if (!common && !generic_sampling(d, hmax, vmax))
            return fail(ctx, MJ_ERR_UNSUPPORTED,
                        "image %d: sampling layout not supported by the MI355X path (ncomp=%d, Y %dx%d, Cb %dx%d, Cr %dx%d); "
                        "supported: one component, or three with factors 1..4 and at most %d blocks per MCU", i, d.ncomp, d.hs[0], d.vs[0],
                        d.hs[1], d.vs[1], d.hs[2], d.vs[2], mj::kMaxBlocksPerMcu);
        if (!common && prog) {
            // the reference's final pass (:1319-1362) resizes every 8x8 block of a component to the full MCU shape and stores it
            // into ratio x ratio blocks: that only fits when the component is 1x1 — or is not resized at all
            for (int c = 0; c < 3; ++c)
                if (!((d.hs[c] == 1 && d.vs[c] == 1) || (d.hs[c] == hmax && d.vs[c] == vmax)))
                    return fail(ctx, MJ_ERR_UNSUPPORTED, "image %d: scan-by-scan files need every component at 1x1 or at the full resolution "
                                "(the reference's final pass cannot place the blocks of a %dx%d component under %dx%d: ValueError)", i,
                                d.hs[c], d.vs[c], hmax, vmax);
        }
        if (i == 0) { p->hmax = hmax; p->vmax = vmax; p->ncomp = d.ncomp; }
        else if (hmax != p->hmax || vmax != p->vmax || d.ncomp != p->ncomp || common == p->generic ||
                 (p->generic && (memcmp(d.hs, b->images[0].hs, sizeof(d.hs)) || memcmp(d.vs, b->images[0].vs, sizeof(d.vs)))))
            return fail(ctx, MJ_ERR_UNSUPPORTED, "image %d: a plan holds one sampling layout; split the batch by layout", i);
        const int mw = d.ncomp == 1 ? 8 : 8 * hmax, mh = d.ncomp == 1 ? 8 : 8 * vmax;
        if (d.mcu_count_h != (d.width + mw - 1) / mw || d.mcu_count_v != (d.height + mh - 1) / mh)
            return fail(ctx, MJ_ERR_INVALID, "image %d: MCU counts %dx%d do not match %dx%d with %dx%d MCUs", i,
                        d.mcu_count_h, d.mcu_count_v, d.width, d.height, mw, mh);
        im.width = d.width; im.height = d.height; im.ncomp = d.ncomp;
        p->max_pixels = std::max(p->max_pixels, (int64_t)d.width * d.height);
        im.hmax = hmax; im.vmax = vmax;
        im.blocks_per_mcu = d.ncomp == 1 ? 1 : hmax * vmax + 2;
        im.generic = common ? 0 : 1;
        if (!common) im.blocks_per_mcu = d.hs[0] * d.vs[0] + d.hs[1] * d.vs[1] + d.hs[2] * d.vs[2];
        im.mcu_count_h = d.mcu_count_h; im.mcu_count_v = d.mcu_count_v;
        im.restart_interval = d.restart_interval;
        const int64_t mcus = (int64_t)d.mcu_count_h * d.mcu_count_v;
        // per-block component / table slots, decode order (jpeg_decoder.py:774, :805)
        int nb = 0;
        for (int c = 0; c < d.ncomp; ++c) {
            if (d.qt_sel[c] < 0 || d.qt_sel[c] >= b->n_qt) return fail(ctx, MJ_ERR_INVALID, "image %d: qt_sel out of range", i);
            im.qt_index[c] = d.qt_sel[c];
            int dslot = 0, aslot = 0;
            if (have_entropy && !prog) {
                if (d.dc_sel[c] < 0 || d.dc_sel[c] >= b->n_huff || d.ac_sel[c] < 0 || d.ac_sel[c] >= b->n_huff)
                    return fail(ctx, MJ_ERR_INVALID, "image %d: Huffman table selector out of range", i);
                auto slot_of = [&](int t) {
                    for (int s = 0; s < im.n_tabs; ++s) if (im.tab_index[s] == t) return s;
                    im.tab_index[im.n_tabs] = t;
                    return im.n_tabs++;
                };
                dslot = slot_of(d.dc_sel[c]);
                aslot = slot_of(d.ac_sel[c]);
            }
            const int rep = d.ncomp == 1 ? 1 : (common ? (c == 0 ? hmax * vmax : 1) : d.hs[c] * d.vs[c]);
            im.comp_h[c] = (uint8_t)(d.ncomp == 1 ? 1 : (common ? (c == 0 ? hmax : 1) : d.hs[c]));
            im.comp_v[c] = (uint8_t)(d.ncomp == 1 ? 1 : (common ? (c == 0 ? vmax : 1) : d.vs[c]));
            im.comp_first[c] = (uint8_t)nb;
            for (int r = 0; r < rep; ++r, ++nb) {
                im.blk_comp[nb] = (uint8_t)c; im.blk_dc_slot[nb] = (uint8_t)dslot; im.blk_ac_slot[nb] = (uint8_t)aslot;
            }
        }
        if (im.n_tabs > p->lut_slots) p->lut_slots = im.n_tabs;
        im.block_off = blk; im.mcu_off = mcu; im.rgb_off = rgb; im.pix_off = pix;
        mcu_prefix[i] = mcu;
        if (i > 0 && (d.width != b->images[0].width || d.height != b->images[0].height)) p->uniform = false;
        if (have_entropy && !prog) {
            const int64_t want = d.restart_interval > 0 ? (mcus + d.restart_interval - 1) / d.restart_interval : 1;
            const bool gpu_seg = (b->flags & MJ_FLAG_GPU_SEGMENT) != 0;
            if (d.n_segments != (gpu_seg ? 1 : want))
                return fail(ctx, MJ_ERR_INVALID, "image %d: %d restart segments given, %lld expected (restart interval %d, %lld MCUs)",
                            i, d.n_segments, (long long)(gpu_seg ? 1 : want), d.restart_interval, (long long)mcus);
            if (d.first_segment < 0 || d.first_segment + d.n_segments > b->n_segments)
                return fail(ctx, MJ_ERR_INVALID, "image %d: segment range outside seg_begin/seg_end", i);
            if (gpu_seg) {      // one byte range per image; stage 0 finds the markers and fills begin/len (destuff.hip)
                const int64_t sb = b->seg_begin[d.first_segment], se = b->seg_end[d.first_segment];
                if (sb < 0 || se < sb || se > b->blob_len || se - sb > 0x7fff0000)
                    return fail(ctx, MJ_ERR_INVALID, "image %d: bad byte range [%lld, %lld)", i, (long long)sb, (long long)se);
                mj::DevScanJob jb{};
                jb.begin = sb; jb.end = se; jb.first_seg = (int64_t)segs.size(); jb.n_seg = (int32_t)want; jb.image = i;
                jobs.push_back(jb);
                ent += se - sb;
            }
            for (int s = 0; s < (int)want; ++s) {
                mj::DevSegment g{};
                if (gpu_seg) {
                    g.begin = b->seg_begin[d.first_segment]; g.len = 0;
                } else {
                    const int64_t sb = b->seg_begin[d.first_segment + s], se = b->seg_end[d.first_segment + s];
                    if (sb < 0 || se < sb || se > b->blob_len || se - sb > 0x7fff0000)
                        return fail(ctx, MJ_ERR_INVALID, "image %d segment %d: bad byte range [%lld, %lld)", i, s, (long long)sb, (long long)se);
                    g.begin = sb; g.len = (int32_t)(se - sb);
                    ent += se - sb;
                }
                g.image = i;
                g.mcu0 = d.restart_interval > 0 ? s * d.restart_interval : 0;
                g.n_mcu = (int32_t)(d.restart_interval > 0 ? std::min<int64_t>(d.restart_interval, mcus - g.mcu0) : mcus);
                g.last = s == (int)want - 1;
                segs.push_back(g);
            }
        }
        blk += mcus * im.blocks_per_mcu;
        mcu += mcus;
        rgb += (int64_t)d.width * d.height * d.ncomp;
        pix += (int64_t)d.width * d.height;
    }
    mcu_prefix[b->n_images] = mcu;
    std::vector<mj::DevProgScan> pscans;
    std::vector<mj::DevProgSeg> psegs;
    if (prog) {
        // (what decides the walks' form comes first: the dependency levels below depend on it)
        p->prog_fast = false;
        for (int k = 0; k < b->n_scans; ++k)       // every scan of a progressive frame (sequential scans — non-interleaved baseline files — stay with progressive.hip)
            p->prog_fast = p->prog_fast || !(b->scans[k].ss == 0 && b->scans[k].se == 63);
        if (const char *e = mj::opt("MJ_PROG_FAST")) p->prog_fast = p->prog_fast && atoi(e) != 0;
        if (p->hmax == 3 || p->vmax == 3) p->prog_fast = false;      // the stream walks step through a component's blocks with shifts
        int max_rows = 1;
        for (int i = 0; i < b->n_images; ++i) max_rows = std::max(max_rows, (int)imgs[i].mcu_count_v);
        p->prog_banded = true;
        if (const char *e = mj::opt("MJ_PROG_BANDS")) p->prog_banded = atoi(e) != 0;
        p->prog_rows_per_band = p->prog_banded ? 1 : max_rows;      // (one frame MCU row per band: 1-2 % faster than two up to 1024 files, equal above)
        if (const char *e = mj::opt("MJ_PROG_ROWS")) { const int v = atoi(e); if (v >= 1 && p->prog_banded) p->prog_rows_per_band = v; }
        // Split scans (progressive_fast.hip): a refining AC scan is one serial chain — a batch lasts as long as its longest scan's
        // walk — and more than half of a block's walk is placing what the symbols say, which needs no order once the bit position
        // of the block is known.  A scout follows the positions alone; a few walks per band (MJ_PROG_PARTS, 4), one launch
        // behind, place.  Worth it where a band's walk is long: from 1 KiB of entropy-coded bytes per band on (MJ_PROG_SPLIT: 0 never,
        // 2 every refining AC scan).
        int split_mode = 1;
        if (const char *e = mj::opt("MJ_PROG_SPLIT")) split_mode = atoi(e);
        if (const char *e = mj::opt("MJ_PROG_PARTS")) p->prog_parts = std::min(std::max(atoi(e), 1), mj::kProgSub);
        if (!p->prog_fast || !p->prog_banded) split_mode = 0;
        std::vector<char> split_of(b->n_scans, 0);
        if (split_mode) {
            // ... and while the chip has wave slots for it: past that the added work — a split scan is walked one and a half
            // times — costs more than the shorter chain gains.
            const int64_t n_bands = std::max(1, (max_rows + p->prog_rows_per_band - 1) / p->prog_rows_per_band);
            std::vector<std::pair<int64_t, int>> cand;        // (bytes, scan)
            for (int k = 0; k < b->n_scans; ++k) {
                const mj_scan_desc &sd = b->scans[k];
                if (sd.ss == 0 || sd.ah == 0 || sd.n_comp != 1) continue;
                if (sd.first_segment < 0 || sd.n_segments < 1 || sd.first_segment + sd.n_segments > b->n_segments) continue;   // (refused below)
                int64_t bytes = 0;
                for (int g = 0; g < sd.n_segments; ++g) bytes += b->seg_end[sd.first_segment + g] - b->seg_begin[sd.first_segment + g];
                if (split_mode >= 2 || bytes / n_bands >= 1024) cand.push_back({-bytes, k});
            }
            // All of them if their scouts and parts find a wave slot at once, else none.  (libjpeg's script, 1080p: 512 files
            // 56.2 ms split / 73.2 not, 768: 69.3 / 76.7, 1024: 92.1 / 78.6, 1536: 135 / 94.  Only each image's largest scan where
            // those fit: 1024 files 79.5, 1536 116.  As many images as fit: worse than either, 99.7 at 1024.)
            const int64_t slots = (int64_t)mj::device_cus() * 32;
            int64_t need_all = 0;
            for (auto &c : cand) need_all += (int64_t)b->scans[c.second].n_segments * (1 + p->prog_parts);
            if (split_mode >= 2 || need_all <= slots)
                for (auto &c : cand) split_of[c.second] = 1;
        }
        auto want_split = [&](int k) { return split_of[k] != 0; };
        std::vector<int> ordinal_of(b->n_scans, 0);
        std::vector<int> seen(b->n_images, 0);
        int n_ord = 0;
        for (int k = 0; k < b->n_scans; ++k) {
            const mj_scan_desc &sd = b->scans[k];
            if (sd.image < 0 || sd.image >= b->n_images) return fail(ctx, MJ_ERR_INVALID, "scan %d: image index out of range", k);
            if (k > 0 && sd.image < b->scans[k - 1].image) return fail(ctx, MJ_ERR_INVALID, "scans must be grouped by image, in file order");
            // Dependency level instead of file ordinal: a scan must wait only for earlier scans of the same image that
            // touch the same coefficients (same component, overlapping spectral band).  libjpeg's 10-scan script has
            // 4 levels: DC | the four first AC scans | the refinements of what is complete | the last luma refinement.
            {
                int lvl = 0;
                for (int j = k - 1; j >= 0 && b->scans[j].image == sd.image; --j) {
                    const mj_scan_desc &pj = b->scans[j];
                    bool comp_overlap = false;
                    for (int a1 = 0; a1 < sd.n_comp && a1 < 3; ++a1)
                        for (int a2 = 0; a2 < pj.n_comp && a2 < 3; ++a2) comp_overlap |= sd.comp[a1] == pj.comp[a2];
                    // (a split scan's parts run one launch behind its scout: what follows it waits for them)
                    if (comp_overlap && sd.ss <= pj.se && pj.ss <= sd.se) lvl = std::max(lvl, ordinal_of[j] + 1 + (want_split(j) ? 1 : 0));
                }
                ordinal_of[k] = lvl;
            }
            (void)seen;
            n_ord = std::max(n_ord, ordinal_of[k] + 1 + (want_split(k) ? 1 : 0));
            const mj_image_desc &d = b->images[sd.image];
            const mj::DevImage &im = imgs[sd.image];
            mj::DevProgScan ps{};
            ps.image = sd.image; ps.n_comp = sd.n_comp;
            if (sd.n_comp < 1 || sd.n_comp > d.ncomp) return fail(ctx, MJ_ERR_INVALID, "scan %d: %d components", k, sd.n_comp);
            // ss = 0, se = 63, ah = al = 0: a sequential (baseline) scan of one component — non-interleaved baseline files
            const bool sequential = sd.ss == 0 && sd.se == 63 && sd.ah == 0 && sd.al == 0;
            if (sd.ss < 0 || sd.se > 63 || sd.se < sd.ss || sd.al < 0 || sd.al > 13 || (sd.ss == 0 && sd.se != 0 && !sequential))
                return fail(ctx, MJ_ERR_INVALID, "scan %d: bad spectral selection / successive approximation", k);
            if ((sd.ss > 0 || sequential) && sd.n_comp != 1)
                return fail(ctx, sequential ? MJ_ERR_UNSUPPORTED : MJ_ERR_INVALID, "scan %d: an AC or sequential scan has one component here", k);
            for (int i = 0; i < sd.n_comp; ++i) {
                if (sd.comp[i] < 0 || sd.comp[i] >= d.ncomp) return fail(ctx, MJ_ERR_INVALID, "scan %d: component out of range", k);
                ps.comp[i] = sd.comp[i];
                const bool need_dc = sd.ss == 0 && sd.ah == 0, need_ac = sd.se > 0;
                if ((need_dc && (sd.dc_sel[i] < 0 || sd.dc_sel[i] >= b->n_huff)) || (need_ac && (sd.ac_sel[i] < 0 || sd.ac_sel[i] >= b->n_huff)))
                    return fail(ctx, MJ_ERR_INVALID, "scan %d: Huffman table selector out of range", k);
                ps.dc_tab[i] = need_dc ? sd.dc_sel[i] : 0;
                ps.ac_tab[i] = need_ac ? sd.ac_sel[i] : 0;
            }
            // geometry the kernel relies on
            int want_h, want_v;
            if (sd.n_comp > 1) {
                // (an interleaved DC scan may cover a subset of the components: its MCUs are still the frame's, :591-594, :610-611)
                want_h = im.mcu_count_h; want_v = im.mcu_count_v;
            } else {
                const int c = sd.comp[0];
                const int h = im.comp_h[c], v = im.comp_v[c];
                if (sd.ss == 0 && (h > 1 || v > 1))
                    return fail(ctx, MJ_ERR_UNSUPPORTED, "scan %d: single-component DC scan of a component with sampling > 1 (the reference steps "
                                "its blocks by the component's MCU size, :993-994, and runs off its array: IndexError)", k);
                const int cw = (d.width * h + im.hmax - 1) / im.hmax, ch = (d.height * v + im.vmax - 1) / im.vmax;   // ceil(W / ratio)
                want_h = (cw + 7) / 8; want_v = (ch + 7) / 8;
                if (d.ncomp == 1) { want_h = (d.width + 7) / 8; want_v = (d.height + 7) / 8; }
            }
            if (sd.mcu_count_h != want_h || sd.mcu_count_v != want_v)
                return fail(ctx, MJ_ERR_INVALID, "scan %d: MCU counts %dx%d, expected %dx%d", k, sd.mcu_count_h, sd.mcu_count_v, want_h, want_v);
            ps.ss = sd.ss; ps.se = sd.se; ps.ah = sd.ah; ps.al = sd.al;
            ps.level = ordinal_of[k];
            ps.split = want_split(k) ? 1 : 0;
            ps.mcu_count_h = sd.mcu_count_h; ps.mcu_count_v = sd.mcu_count_v;
            pscans.push_back(ps);
        }
        // segments grouped by ordinal
        // ... and, inside a level, by the kernel that walks them: DC first scans, AC first scans, AC refining scans, the rest
        auto kind_of = [&](const mj_scan_desc &sd) {
            const bool sequential = sd.ss == 0 && sd.se == 63;
            if (sequential) return 3;
            if (sd.ss == 0) return 0;           // DC scans, first and refining (round 4: the refinement is walked by progressive_fast.hip too)
            return sd.ah == 0 ? 1 : 2;
        };
        p->ordinal_seg_off.assign(n_ord + 1, 0);
        p->ordinal_kind_off.assign((size_t)n_ord * 4, 0);
        for (int o = 0; o < n_ord; ++o) {
            p->ordinal_seg_off[o] = (int64_t)psegs.size();
          for (int kind = 0; kind < 4; ++kind) {
            p->ordinal_kind_off[(size_t)o * 4 + kind] = (int64_t)psegs.size();
            for (int k = 0; k < b->n_scans; ++k) {
                if (ordinal_of[k] != o || kind_of(b->scans[k]) != kind) continue;
                const mj_scan_desc &sd = b->scans[k];
                const int64_t mcus = (int64_t)sd.mcu_count_h * sd.mcu_count_v;
                const int64_t want = sd.restart_interval > 0 ? (mcus + sd.restart_interval - 1) / sd.restart_interval : 1;
                if (sd.n_segments != want || sd.first_segment < 0 || sd.first_segment + sd.n_segments > b->n_segments)
                    return fail(ctx, MJ_ERR_INVALID, "scan %d: %d restart segments given, %lld expected", k, sd.n_segments, (long long)want);
                for (int sgi = 0; sgi < sd.n_segments; ++sgi) {
                    const int64_t sb = b->seg_begin[sd.first_segment + sgi], se = b->seg_end[sd.first_segment + sgi];
                    if (sb < 0 || se < sb || se > b->blob_len) return fail(ctx, MJ_ERR_INVALID, "scan %d segment %d: bad byte range", k, sgi);
                    mj::DevProgSeg g{};
                    g.begin = sb; g.len = (int32_t)(se - sb); g.scan = k;
                    g.mcu0 = sd.restart_interval > 0 ? sgi * sd.restart_interval : 0;
                    g.n_mcu = (int32_t)(sd.restart_interval > 0 ? std::min<int64_t>(sd.restart_interval, mcus - g.mcu0) : mcus);
                    g.last = sgi == sd.n_segments - 1;
                    psegs.push_back(g);
                    ent += se - sb;
                }
            }
          }
        }
        p->ordinal_seg_off[n_ord] = (int64_t)psegs.size();
        {   // Band pipelining (see progressive_fast.hip): one frame MCU row per band (round 4; two before), launches = bands + levels - 1.  It
            // shortens the critical path from the sum of the levels' longest scans to about the longest scan — a refining scan
            // follows one band behind what it refines — and keeps all of an image's scans on the chip at once: faster than one
            // launch per dependency level at every batch size measured (profiles/r02d_progressive_sweep.txt: 16 x 1080p
            // 146 -> 80.5 ms, 1024: 184 -> 93 ms, 8192: 728 -> 510 ms with the ordering and the loops of progressive_fast.hip).
            // MJ_PROG_BANDS=0 keeps one launch per level.  (MJ_PROG_BANDS, MJ_PROG_ROWS, MJ_PROG_FAST, MJ_SYNC_ROUNDS,
            // MJ_SYNC_CHUNK, MJ_HUFFMAN, MJ_SEG_ORDER and the MJ_LANES_* variables are hooks of the test-suite and of
            // tools/stage_probe.py: read once, at plan creation or launch; mj_plan_stage1_form() reports the form in effect.)
            if (p->prog_banded) {
                // every launch of the pipeline covers all segments, and more workgroups than the chip holds at once: the long
                // walks go first (a launch lasts as long as its slowest wave; started last, the final luma refinement — half
                // of a file's bytes — would begin when the short scans' waves leave)
                // (behind them the segments of the scans progressive.hip walks — DC refinement, sequential scans — so that its
                // launches cover only those)
                auto rest = [&](const mj::DevProgSeg &g) { return kind_of(b->scans[g.scan]) == 3; };
                // (in front of them all the split scans' segments: the kernel finds their parts by position)
                auto split = [&](const mj::DevProgSeg &g) { return pscans[g.scan].split != 0; };
                std::stable_sort(psegs.begin(), psegs.end(), [&](const mj::DevProgSeg &x, const mj::DevProgSeg &y) {
                    if (rest(x) != rest(y)) return rest(y);
                    if (split(x) != split(y)) return split(x);
                    return x.len > y.len;
                });
                p->n_split = 0;
                while (p->n_split < (int64_t)psegs.size() && split(psegs[p->n_split])) ++p->n_split;
                p->prog_rest_off = 0;
                while (p->prog_rest_off < (int64_t)psegs.size() && !rest(psegs[p->prog_rest_off])) ++p->prog_rest_off;
            }
            const int n_bands = (max_rows + p->prog_rows_per_band - 1) / p->prog_rows_per_band;
            p->prog_steps = n_bands + n_ord - 1;
        }
    }
    p->mcus_per_image = (int32_t)(mcu / b->n_images);
    p->info.total_blocks = blk; p->info.total_mcus = mcu; p->info.total_pixels = pix;
    p->info.rgb_bytes = rgb; p->info.entropy_bytes = ent;
    p->n_segs = (int64_t)segs.size();

    int rc;
    if ((rc = upload(ctx, &p->d_images, imgs.data(), imgs.size())) != MJ_OK) return rc;
    if ((rc = upload(ctx, &p->d_mcu_prefix, mcu_prefix.data(), mcu_prefix.size())) != MJ_OK) return rc;
    {
        std::vector<uint16_t> qn((size_t)b->n_qt * 64);
        for (int t = 0; t < b->n_qt; ++t)
            for (int z = 0; z < 64; ++z) {
                const int n = kNatOfZz[z];
                qn[(size_t)t * 64 + (p->transposed ? ((n & 7) << 3 | n >> 3) : n)] = b->qt[(size_t)t * 64 + z];
            }
        if ((rc = upload(ctx, &p->d_qt, qn.data(), qn.size())) != MJ_OK) return rc;
        // the fast stage 2 hands its work out in JOBS (reconstruct_fast.hip): up to `chunk_strips` vertically consecutive strips
        // (a strip = fast_tile_mcus() MCUs) of one MCU column — a whole column where that is at most 24 strips (1080p: 17),
        // else equal pieces of one.  Jobs are numbered image by image; the kernel's ticket counter is the (zero) word behind
        // the prefix.
        const int tm = p->generic ? 1 : mj::fast_tile_mcus(p->hmax, p->vmax, p->ncomp, p->transposed);
        int max_spc = 1;
        for (int i = 0; i < b->n_images; ++i) {
            const int rows = p->transposed ? imgs[i].mcu_count_h : imgs[i].mcu_count_v;
            max_spc = std::max(max_spc, (rows + tm - 1) / tm);
        }
        const int pieces_max = (max_spc + 23) / 24;
        p->chunk_strips = (max_spc + pieces_max - 1) / pieces_max;
        if (const char *e = mj::opt("MJ_STAGE2_CHUNK")) { const int v = atoi(e); if (v >= 1 && v <= 4096) p->chunk_strips = v; }
        std::vector<int64_t> tp(b->n_images + 1, 0);
        for (int i = 0; i < b->n_images; ++i) {
            // strips run down the MCU columns of the image the kernel sees (the transposed one for row-major plans)
            const int cols = p->transposed ? imgs[i].mcu_count_v : imgs[i].mcu_count_h;
            const int rows = p->transposed ? imgs[i].mcu_count_h : imgs[i].mcu_count_v;
            const int spc = (rows + tm - 1) / tm;
            tp[i + 1] = tp[i] + (int64_t)cols * ((spc + p->chunk_strips - 1) / p->chunk_strips);
        }
        p->total_jobs = tp[b->n_images];
        {   // a ticket should be worth ~400 blocks of IDCT work (a 1080p 4:2:0 column: 17 strips x 24 blocks): consecutive jobs per ticket
            const int blocks_per_strip = p->generic ? 1 : tm * (p->ncomp == 1 ? 1 : p->hmax * p->vmax + 2);
            const int per_job = std::max(1, blocks_per_strip * std::min(p->chunk_strips, max_spc));
            p->jobs_per_ticket = std::max(1, (400 + per_job / 2) / per_job);
        }
        p->jobs_per_image = (int32_t)(tp[1] - tp[0]);
        tp.insert(tp.end(), 5, 0);         // the ticket counter, a spare word, the three level counters of mj_plan_idct_levels
        if ((rc = upload(ctx, &p->d_job_prefix, tp.data(), tp.size())) != MJ_OK) return rc;
    }
    if (have_entropy) {
        std::vector<mj::DevHuff> hh(b->n_huff);
        for (int t = 0; t < b->n_huff; ++t) build_dev_huff(b->huff[t], hh[t]);
        if ((rc = upload(ctx, &p->d_huff, hh.data(), hh.size())) != MJ_OK) return rc;
        p->n_huff = b->n_huff;
        std::vector<int> role(b->n_huff, 0);
        bool both_roles = false, dc_fits = true;
        {   // 11-bit LUTs for the lane-parallel kernel
            const int LB = mj::kLaneLutBits, LS = 1 << LB;
            std::vector<uint16_t> l11((size_t)b->n_huff * LS, 0);
            // how each table is used: bit 0 = as a DC table, bit 1 = as an AC table (the two LUT formats differ)
            for (const mj::DevImage &im : imgs)
                for (int k2 = 0; k2 < im.blocks_per_mcu && k2 < mj::kMaxBlocksPerMcu; ++k2) {
                    role[im.tab_index[im.blk_dc_slot[k2]]] |= 1;
                    role[im.tab_index[im.blk_ac_slot[k2]]] |= 2;
                }
            for (int t = 0; t < b->n_huff; ++t) both_roles = both_roles || role[t] == 3;
            for (int t = 0; t < b->n_huff; ++t) {
                int code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) {
                    code <<= 1;
                    for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                        if (l <= LB && code < (1 << l)) {
                            const int shift = LB - l, hv = b->huff[t].vals[k];
                            // AC tables: length, zero run and size ready for use; end of block = a run of 64 (huffman_lanes.hip)
                            const uint16_t entry = (role[t] & 2) ? (uint16_t)((l << 11) | ((hv == 0 ? 64 : hv >> 4) << 4) | (hv & 15))
                                                                 : (uint16_t)((l << 8) | hv);
                            for (int f = 0; f < (1 << shift); ++f) {
                                uint16_t &e = l11[(size_t)t * LS + ((code << shift) | f)];
                                if (e == 0) e = entry;
                            }
                        }
                    }
                }
            }
            if ((rc = upload(ctx, &p->d_lut11, l11.data(), l11.size())) != MJ_OK) return rc;
            // the fast variant of the lane form (huffman_lanes13.hip): 13-bit AC tables whose entries are finished symbols
            // — bits consumed, step of the write position, EXTENDed coefficient (jpeg_decoder.py:834-866, :1636-1646) —
            // wherever code + value bits fit the index; every table must have one role and the lot must fit LDS
            {
                int n_ac = 0, n_dc = 0;
                uint64_t ac_pk = 0, dc_pk = 0, dct_pk = 0;
                bool ok13 = b->n_huff <= 8 && !both_roles && !prog;
                for (int t = 0; t < b->n_huff && ok13; ++t) {
                    if (role[t] == 2) ac_pk |= (uint64_t)n_ac++ << (8 * t);
                    else if (role[t] == 1) { dc_pk |= (uint64_t)n_dc << (8 * t); dct_pk |= (uint64_t)t << (8 * n_dc); ++n_dc; }
                }
                const char *f13 = mj::opt("MJ_HUFFMAN");
                if (f13 && !strcmp(f13, "lanes11")) ok13 = false;
                if (ok13 && mj::lanes13_fits(n_ac, n_dc)) {
                    std::vector<uint32_t> l13;
                    int slot_bytes = 0;
                    if (!build_resolved_tables(b, role, ac_pk, n_ac, 13, mj::kLanes13SlotBytes, l13, slot_bytes)) goto no_lanes13;
                    if ((rc = upload(ctx, &p->d_lut13, l13.data(), l13.size())) != MJ_OK) return rc;
                    p->n_ac13 = n_ac; p->n_dc13 = n_dc;
                    p->ac_slot_pk = ac_pk; p->dc_slot_pk = dc_pk; p->dc_tab_pk = dct_pk;
                    // the same tables with a 12-bit main level, as small as the batch's codes allow: what a fused launch keeps in
                    // LDS beside its reconstruction wavefronts' strips (fused.hip)
                    std::vector<uint32_t> l12;
                    if (build_resolved_tables(b, role, ac_pk, n_ac, 12, 0, l12, p->lut12_slot_bytes))
                        if ((rc = upload(ctx, &p->d_lut12, l12.data(), l12.size())) != MJ_OK) return rc;
                }
            no_lanes13:;
            }
            {   // huffman_sync.hip wants every table in the unified format (DC tables: run 0, size = the symbol)
                std::vector<uint16_t> lu = l11;
                for (int t = 0; t < b->n_huff; ++t) {
                    if (role[t] & 2) continue;
                    std::fill(lu.begin() + (size_t)t * LS, lu.begin() + (size_t)(t + 1) * LS, (uint16_t)0);
                    int code = 0, k = 0;
                    for (int l = 1; l <= 16; ++l) {
                        code <<= 1;
                        for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                            if (l <= LB && code < (1 << l)) {
                                const int shift = LB - l, hv = b->huff[t].vals[k];
                                if (hv > 15) dc_fits = false;          // a DC size above 15 has no place in the format
                                for (int f = 0; f < (1 << shift); ++f) {
                                    uint16_t &e = lu[(size_t)t * LS + ((code << shift) | f)];
                                    if (e == 0) e = (uint16_t)((l << 11) | (hv & 15));
                                }
                            }
                        }
                    }
                }
                if ((rc = upload(ctx, &p->d_lut11u, lu.data(), lu.size())) != MJ_OK) return rc;
            }
        }
        // one segment per lane pays off once there are enough segments to fill the chip that way
        const char *force = mj::opt("MJ_HUFFMAN");
        // (a table serving as DC and as AC table at once, or a stream beyond 32-bit offsets, stays with the wave form)
        // stage 0 places segment i's stream at dword (begin_i >> 2) + i: that needs the segments (or, with the GPU
        // marker scan, the images' byte ranges) in ascending, non-overlapping blob order — what any packer produces
        bool ordered = true;
        if (jobs.empty()) {
            for (size_t i = 1; i < segs.size() && ordered; ++i) ordered = segs[i].begin >= segs[i - 1].begin + segs[i - 1].len;
        } else {
            for (size_t i = 1; i < jobs.size() && ordered; ++i) ordered = jobs[i].begin >= jobs[i - 1].end;
        }
        // More tables than LDS holds (every file with its own optimised tables): a workgroup's segments belong to one or
        // two images, so it loads just their tables — if every workgroup of the launch gets by with 8, or else 16, of them
        // (16 LUTs = 64 KiB leave room for two workgroups per CU instead of four: slower, but not the wave form).
        const bool many_tabs = b->n_huff > mj::kMaxLaneTables;
        auto wg_lists = [&](const std::vector<int32_t> &unit_image, int64_t units_per_wg, int cap, std::vector<int32_t> &lists) -> bool {
            const int64_t n_wg = ((int64_t)unit_image.size() + units_per_wg - 1) / units_per_wg;
            lists.assign((size_t)n_wg * mj::kMaxWgTables, -1);
            for (int64_t g = 0; g < n_wg; ++g) {
                int32_t *l = lists.data() + (size_t)g * mj::kMaxWgTables;
                int n = 0, last_img = -1;
                const int64_t u1 = std::min<int64_t>((g + 1) * units_per_wg, (int64_t)unit_image.size());
                for (int64_t u = g * units_per_wg; u < u1; ++u) {
                    const int img = unit_image[(size_t)u];
                    if (img == last_img) continue;
                    last_img = img;
                    for (int k2 = 0; k2 < imgs[img].n_tabs; ++k2) {
                        const int t = imgs[img].tab_index[k2];
                        bool seen = false;
                        for (int j = 0; j < n; ++j) seen = seen || l[j] == t;
                        if (seen) continue;
                        if (n == cap) return false;
                        l[n++] = t;
                    }
                }
            }
            return true;
        };
        std::vector<int32_t> seg_image, chunk_image, wl_lanes, wl_count;
        bool many_ok_dri = true, many_ok_sync = true;
        if (many_tabs && !prog && !both_roles) {
            seg_image.reserve(segs.size());
            for (const auto &g : segs) seg_image.push_back(g.image);
            many_ok_dri = false;
            for (int cap = 8; cap <= mj::kMaxWgTables && !many_ok_dri; cap *= 2) {
                many_ok_dri = wg_lists(seg_image, 4 * (int64_t)mj::lanes_per_wave((int64_t)segs.size(), cap), cap, wl_lanes);
                p->wg_slots_lanes = cap;
            }
        }
        const bool lanes_ok = ordered && (!many_tabs || many_ok_dri) && !both_roles && !prog && !p->generic &&
                              (uint64_t)b->blob_len + 4 * (uint64_t)segs.size() + 4096 < (1ull << 32);
        p->use_lanes = lanes_ok && (int64_t)segs.size() >= 1024;       // measured crossover with the wave form: ~1000 segments
        if (force && !strcmp(force, "wave")) p->use_lanes = false;
        if (force && !strcmp(force, "lanes") && lanes_ok) p->use_lanes = true;
        // Long segments (no DRI, or a very large restart interval) leave the chip empty at one lane each: they are cut
        // into chunks, the decoder state at the chunk boundaries is found by synchronisation rounds (huffman_sync.hip)
        // and the pieces are decoded by the lane-parallel kernel.  Chosen when segments average >= 32 KiB (a single such
        // image already wins: the alternative is one serial walk per segment); MJ_HUFFMAN=sync forces it, wave / lanes
        // exclude it.  With the GPU marker scan the segment lengths are not known here: possible when every image is one
        // segment (no DRI), whose byte range bounds its length.
        if (const char *e = mj::opt("MJ_SYNC_ROUNDS")) { const int v = atoi(e); if (v >= 0 && v <= 64) p->sync_rounds = v; }
        if (const char *e = mj::opt("MJ_SYNC_CHUNK")) { const int v = atoi(e); if (v >= 256 && v <= 65536 && v % 4 == 0) p->sync_chunk_bytes = v; }
        if (const char *e = mj::opt("MJ_SYNC_WARM")) p->sync_warm_bits = atoi(e) * 8;
        bool one_seg_each = true;
        for (const auto &jb : jobs) one_seg_each = one_seg_each && jb.n_seg == 1;
        if (!jobs.empty() && one_seg_each)
            for (size_t i = 0; i < jobs.size(); ++i) segs[(size_t)jobs[i].first_seg].len = (int32_t)(jobs[i].end - jobs[i].begin);   // upper bound; the scan writes the real one
        int64_t total_len = 0, est_chunks = 0;
        for (const auto &g : segs) total_len += g.len;
        if (!mj::opt("MJ_SYNC_CHUNK")) {
            // Chunk size by the amount of stream: small batches want many short chunks (a single 1080p image: 1.65 ms with
            // 512-byte chunks, 3.0 ms with 2 KiB ones — six wavefronts' worth), big ones fewer long ones (the run-up in
            // front of every chunk and the per-chunk records cost; 1024 images: 19.0 ms at 2 KiB, 20.2 ms at 512 bytes).
            // Measured optimum: the shortest of 512 / 1024 / 2048 bytes that keeps the batch under ~330 000 chunks.
            // ... and a single segment not in more than ~12 000 of them: wrongly guessed entry states are repaired one link
            // of a chain per round, and chains grow with the chunks of a segment (one 24-megapixel image, 10 MB of
            // stream: 5.4 ms with 512-byte chunks, 3.8 ms with 1 KiB).
            int32_t longest = 0;
            for (const auto &g : segs) longest = std::max(longest, g.len);
            p->sync_chunk_bytes = 2048;
            for (int cb : {512, 1024})
                if (total_len / cb <= 330000 && longest / cb <= 12000) { p->sync_chunk_bytes = cb; break; }
        }
        for (const auto &g : segs) est_chunks += std::max(1, (g.len + p->sync_chunk_bytes - 1) / p->sync_chunk_bytes);
        // (one long segment is enough: in a batch that mixes files with and without restart markers, an image without
        // them would otherwise be one lane's — or one wavefront's — serial walk, 220 ms for a 1080p file)
        int32_t max_len = 0;
        for (const auto &g : segs) max_len = std::max(max_len, g.len);
        const bool long_segs = !segs.empty() && (total_len / (int64_t)segs.size() >= 32768 || max_len >= 65536) && est_chunks >= 64;
        // ... and so do small batches of ordinary restart segments: below ~20 000 segments the lane form cannot fill the
        // chip (its time is one segment's serial walk, ~3.3 ms for a 1080p MCU row, however few there are), while chunks
        // can (measured, 1080p with one restart interval per MCU row: 1 image 3.3 -> 1.4 ms, 32 images 3.9 -> 2.1 ms,
        // 128 images 4.9 -> 3.8 ms, break-even at ~300 images = 20 000 segments).  Segments shorter than a few chunks
        // gain nothing from being cut.
        const bool few_segs = !segs.empty() && (int64_t)segs.size() < 20000 && total_len / (int64_t)segs.size() >= 2048 && est_chunks >= 64;
        // (with many tables the synchronisation form needs its own two workgroup shapes to get by with their table lists;
        // its lane launch runs over chunks, so the restart-segment shape checked above does not matter for it)
        const bool sync_shape_ok = ordered && !both_roles && !prog && !p->generic && (uint64_t)b->blob_len + 4 * (uint64_t)segs.size() + 4096 < (1ull << 32);
        bool want_sync = !(b->flags & MJ_FLAG_NO_SYNC) && (many_tabs ? sync_shape_ok : lanes_ok) && (jobs.empty() || one_seg_each) && dc_fits &&
                         ((force && !strcmp(force, "sync")) || (!force && (long_segs || few_segs)));
        if (want_sync && many_tabs) {
            // (shorter chunks = less stream per workgroup = fewer images per workgroup: if the chunk size chosen above
            // leaves some workgroup with too many tables, shorter chunks get a try)
            many_ok_sync = false;
            for (int cb : {p->sync_chunk_bytes, 512, 256}) {
                if (cb > p->sync_chunk_bytes) continue;
                chunk_image.clear();
                for (size_t i = 0; i < segs.size(); ++i)
                    for (int j = 0; j < std::max(1, (segs[i].len + cb - 1) / cb); ++j) chunk_image.push_back(segs[i].image);
                bool ok_count = false, ok_lanes = false;
                for (int cap = 8; cap <= mj::kMaxWgTables && !ok_count; cap *= 2) {
                    ok_count = wg_lists(chunk_image, 256, cap, wl_count);
                    p->wg_slots_count = cap;
                }
                for (int cap = 8; cap <= mj::kMaxWgTables && !ok_lanes; cap *= 2) {
                    ok_lanes = wg_lists(chunk_image, 4 * (int64_t)mj::lanes_per_wave((int64_t)chunk_image.size(), cap), cap, wl_lanes);
                    p->wg_slots_lanes = cap;
                }
                if (ok_count && ok_lanes) { many_ok_sync = true; p->sync_chunk_bytes = cb; break; }
            }
            if (!many_ok_sync) want_sync = false;
        }
        if (want_sync) p->use_lanes = true;
        if (p->use_lanes && many_tabs) {
            if ((rc = upload(ctx, &p->d_wg_tabs_lanes, wl_lanes.data(), wl_lanes.size())) != MJ_OK) return rc;
            if (want_sync && (rc = upload(ctx, &p->d_wg_tabs_count, wl_count.data(), wl_count.size())) != MJ_OK) return rc;
        }
        if (p->use_lanes) {
            // stage 0 output: segment i's kept bytes start at dword (begin_i >> 2) + i, so regions never overlap
            const size_t sbytes = ((size_t)b->blob_len / 4 + segs.size() + 256) * 4;
            MJ_HIP(ctx, ctx->cache.get((void **)&p->d_stream, sbytes));
            MJ_HIP(ctx, hipMemsetAsync(p->d_stream, 0, sbytes, ctx->setup_stream));
            MJ_HIP(ctx, ctx->cache.get((void **)&p->d_seg_bits, (segs.size() + 1) * sizeof(int32_t)));
            if (want_sync) {
                const int cb = p->sync_chunk_bytes;
                std::vector<mj::DevChunk> ck;
                for (size_t i = 0; i < segs.size(); ++i)
                    for (int j = 0; j < std::max(1, (segs[i].len + cb - 1) / cb); ++j) ck.push_back(mj::DevChunk{(int32_t)i, j});
                p->n_chunks = (int64_t)ck.size();
                if ((rc = upload(ctx, &p->d_chunks, ck.data(), ck.size())) != MJ_OK) return rc;
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_stateA, ck.size() * 8 + 16));
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_stateB, ck.size() * 8 + 16));
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_couts, ck.size() * sizeof(mj::DevChunkOut) + 16));
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_vsegs, ck.size() * sizeof(mj::DevVSeg) + 16));
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_changed, (size_t)(p->sync_rounds + 8) * sizeof(int32_t)));   // [0]: round 0's, [r]: repair round r's count of changed exit states
                // stage 0 of long segments runs piece by piece (16 KiB of source bytes per wavefront)
                std::vector<mj::DevPiece> pcs;
                for (size_t i = 0; i < segs.size(); ++i) {
                    const int32_t first = (int32_t)pcs.size();
                    for (int off = 0; off == 0 || off < segs[i].len; off += 16384)
                        pcs.push_back(mj::DevPiece{(int32_t)i, first, off, std::min(16384, std::max(0, segs[i].len - off))});
                }
                p->n_pieces = (int64_t)pcs.size();
                if ((rc = upload(ctx, &p->d_pieces, pcs.data(), pcs.size())) != MJ_OK) return rc;
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_piece_kept, pcs.size() * sizeof(int32_t) + 16));
                p->use_sync = true;
            }
        }
        if (p->use_lanes && !p->use_sync && p->d_lut13 && jobs.empty() && segs.size() > 1) {
            // the lane form deals restart segments out by length (huffman_lanes13.hip); MJ_SEG_ORDER = blob | binned | striped (tests, measurements)
            const char *e = mj::opt("MJ_SEG_ORDER");
            // (measured, 1024 x 1080p: files of mixed content 7.5 ms in blob order, 7.9 binned, 6.65 striped; files of one kind
            // 4.01 / 4.13 — so segments of similar length stay in blob order)
            int64_t sum_len = 0;
            int32_t top_len = 0;
            for (const auto &g : segs) { sum_len += g.len; top_len = std::max(top_len, g.len); }
            const bool spread = (int64_t)top_len * 4 * (int64_t)segs.size() > 5 * sum_len;             // longest > 1.25 x mean
            p->seg_order_mode = (e && !strcmp(e, "blob")) ? 0 : ((e && !strcmp(e, "binned")) ? 1 : ((e && !strcmp(e, "striped")) || spread ? 2 : 0));
            if (p->seg_order_mode) {
                std::vector<int32_t> ord(segs.size());
                for (size_t i = 0; i < segs.size(); ++i) ord[i] = (int32_t)i;
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return segs[x].len > segs[y].len; });
                if ((rc = upload(ctx, &p->d_by_length, ord.data(), ord.size())) != MJ_OK) return rc;
            }
        }
        {   // One launch for both stages (fused.hip) where the batch allows it: the resolved-table lane form in blob order on
            // a uniform batch of 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 colour files whose restart interval is ONE MCU ROW (a producer
            // wave that is through MCU m has then finished column m of all its rows, which is the consumers' unit of work),
            // x-major pixels, no seam outputs, a stage-2 job = a whole MCU column, and LDS left for at least one consumer
            // wavefront beside the producers.  MJ_FUSED=0 keeps the two launches; MJ_FUSED_CONSUMERS bounds the consumers.
            int want_cons = 8;
            bool allow = true;
            if (const char *e = mj::opt("MJ_FUSED")) allow = atoi(e) != 0;
            if (const char *e = mj::opt("MJ_FUSED_CONSUMERS")) want_cons = atoi(e);
            const mj::DevImage &i0 = imgs[0];
            // (not for restart segments of very different lengths, which the lane launch deals out by length: a fused launch
            // walks them in blob order — whole images per workgroup — and its longest wave then sets the pace of everything;
            // measured on bench.py's mixed content: 11.3 ms fused against 10.6 as two launches)
            const bool shape_ok = p->use_lanes && !p->use_sync && p->d_lut13 && p->seg_order_mode == 0 && p->uniform && !p->generic && !prog &&
                                  p->ncomp == 3 && (p->hmax == 1 || p->hmax == 2) && (p->vmax == 1 || p->vmax == 2) && !p->transposed &&
                                  p->layout == MJ_LAYOUT_XMAJOR && !(p->flags & (MJ_FLAG_EXACT_ONLY | MJ_FLAG_KEEP_PLANES | MJ_FLAG_KEEP_IDCT)) &&
                                  i0.restart_interval == i0.mcu_count_h && p->jobs_per_image == i0.mcu_count_h &&
                                  (int64_t)segs.size() == (int64_t)b->n_images * i0.mcu_count_v;
            if (allow && want_cons > 0 && shape_ok && p->d_lut12) {
                p->fused = mj::fused_shape(p->n_ac13, p->n_dc13, p->lut12_slot_bytes, p->hmax, p->vmax, b->n_images, i0.mcu_count_v, want_cons);
                p->fused_spi = i0.mcu_count_v;
                p->use_fused = p->fused.ok;
            }
        }
        if ((rc = upload(ctx, &p->d_segs, segs.data(), segs.size())) != MJ_OK) return rc;
        if (!jobs.empty()) {
            if (prog) return fail(ctx, MJ_ERR_INVALID, "MJ_FLAG_GPU_SEGMENT is for baseline batches");
            if ((rc = upload(ctx, &p->d_jobs, jobs.data(), jobs.size())) != MJ_OK) return rc;
            p->n_jobs = (int)jobs.size();
        }
        if (prog) {
            if ((rc = upload(ctx, &p->d_pscans, pscans.data(), pscans.size())) != MJ_OK) return rc;
            p->n_psegs = (int64_t)psegs.size();
            MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pstates, psegs.size() * sizeof(mj::DevProgState) + 16));
            // DC/AC first scans and AC refining scans walk the stage-0 stream (progressive_fast.hip)
            // (a plan none of whose scans they take — non-interleaved baseline files, DC refinement only — needs neither the
            // stage-0 stream nor its pass per execute)
            if (p->prog_fast) {
                // stage 0 for every segment of the progressive scans, 16 KiB of source bytes per wavefront.  Stage 0 puts
                // segment number n at dword (begin >> 2) + n of the stream buffer, which keeps the segments apart only if
                // they are numbered in blob order — psegs is ordered by dependency level, so each one records its number
                std::vector<int32_t> order(psegs.size());
                for (size_t i = 0; i < psegs.size(); ++i) order[i] = (int32_t)i;
                std::sort(order.begin(), order.end(), [&](int32_t a2, int32_t b2) { return psegs[a2].begin < psegs[b2].begin; });
                std::vector<mj::DevSegment> ds(psegs.size());
                std::vector<mj::DevPiece> pcs;
                for (size_t n = 0; n < psegs.size(); ++n) {
                    mj::DevProgSeg &g = psegs[order[n]];
                    g.stream_slot = (int32_t)n;
                    ds[n] = mj::DevSegment{g.begin, g.len, pscans[g.scan].image, g.mcu0, g.n_mcu, g.last, 0};
                    const int32_t first = (int32_t)pcs.size();
                    for (int off = 0; off == 0 || off < g.len; off += 16384)
                        pcs.push_back(mj::DevPiece{(int32_t)n, first, off, std::min(16384, std::max(0, g.len - off))});
                }
                if ((rc = upload(ctx, &p->d_prog_dsegs, ds.data(), ds.size())) != MJ_OK) return rc;
                p->n_pieces = (int64_t)pcs.size();
                if ((rc = upload(ctx, &p->d_pieces, pcs.data(), pcs.size())) != MJ_OK) return rc;
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_piece_kept, pcs.size() * sizeof(int32_t) + 16));
                const size_t sbytes = ((size_t)b->blob_len / 4 + psegs.size() + 256) * 4;
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_stream, sbytes));
                MJ_HIP(ctx, hipMemsetAsync(p->d_stream, 0, sbytes, ctx->setup_stream));
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_seg_bits, (psegs.size() + 1) * sizeof(int32_t)));
                const int LS = 1 << mj::kProgLutBits;
                std::vector<uint16_t> lp((size_t)b->n_huff * LS, 0);
                for (int t = 0; t < b->n_huff; ++t) {
                    int code = 0, k = 0;
                    for (int l = 1; l <= 16; ++l) {
                        code <<= 1;
                        for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                            if (l > mj::kProgLutBits || code >= (1 << l)) continue;
                            const int shift = mj::kProgLutBits - l;
                            for (int f = 0; f < (1 << shift); ++f) {
                                uint16_t &e = lp[(size_t)t * LS + ((code << shift) | f)];
                                if (e == 0) e = (uint16_t)((l << 8) | b->huff[t].vals[k]);      // the shortest key wins
                            }
                        }
                    }
                }
                if ((rc = upload(ctx, &p->d_lut11p, lp.data(), lp.size())) != MJ_OK) return rc;
            }
            if ((rc = upload(ctx, &p->d_psegs, psegs.data(), psegs.size())) != MJ_OK) return rc;
            if (p->n_split)     // by segment (the first n_split of them), two sets: even and odd bands
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_psubs, (size_t)p->n_split * 2 * mj::kProgSub * sizeof(mj::DevProgSub)));
        }
        if (b->blob_mem == MJ_MEM_HOST) {
            if ((rc = upload(ctx, &p->d_blob_owned, b->blob, (size_t)b->blob_len, 1024)) != MJ_OK) return rc;
            p->d_blob = p->d_blob_owned;
        } else {
            if (((uintptr_t)b->blob & 3) != 0) return fail(ctx, MJ_ERR_INVALID, "device blob must be 4-byte aligned");
            // the bit readers fetch up to 127 dwords past a segment's aligned start (wave_bits.h) and one dword ahead per
            // lane: a caller-owned blob must be that much longer than its last segment (uploads get the slack here)
            int64_t last_end = 0;
            for (int64_t i = 0; i < b->n_segments; ++i) last_end = b->seg_end[i] > last_end ? b->seg_end[i] : last_end;
            if (!(b->flags & MJ_FLAG_GPU_SEGMENT) && last_end + 512 > b->blob_len)
                return fail(ctx, MJ_ERR_INVALID, "device blob: blob_len must include 512 readable bytes behind the last segment");
            if (!jobs.empty() && ((uintptr_t)b->blob & 15) != 0)
                return fail(ctx, MJ_ERR_INVALID, "MJ_FLAG_GPU_SEGMENT: device blob must be 16-byte aligned");
            p->d_blob = b->blob;
            // MJ_FLAG_GPU_SEGMENT promises 16 readable bytes behind blob_len, which is all the marker scan and stage 0 need —
            // but a plan that ends up in the wave form (small batches, generic sampling layouts, tables in both roles) reads the
            // blob itself, up to 508 bytes behind a segment's aligned start: such a plan works on its own padded copy
            // (copied at every execute, on the execute's stream: the caller's bytes need not be there yet when the plan is made)
            if ((b->flags & MJ_FLAG_GPU_SEGMENT) && !p->use_lanes && last_end + 512 > b->blob_len) {
                MJ_HIP(ctx, ctx->cache.get((void **)&p->d_blob_owned, (size_t)b->blob_len + 1024 + 16));
                MJ_HIP(ctx, hipMemsetAsync(p->d_blob_owned + b->blob_len, 0, 1024, ctx->setup_stream));
                p->blob_src = b->blob; p->blob_src_len = b->blob_len;
                p->d_blob = p->d_blob_owned;
            }
        }
    }
    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_coef, (size_t)blk * 64 * sizeof(int16_t) + 16));
    // (the resolved-table lane form stores every block of every MCU of every segment it is given, zeros included: no need to
    // clear 6 GB per plan first — 1.5 ms of a 1024-image plan's creation.  Only where the host listed the segments, though:
    // virtual segments of an image that did not settle, or the segments of a file whose marker count is off, do not cover
    // their image, and what a recycled buffer held before must not show through in a failed image's pixels)
    if (!(p->d_lut13 && p->use_lanes && !p->use_sync && jobs.empty()))
        MJ_HIP(ctx, hipMemsetAsync(p->d_coef, 0, (size_t)blk * 64 * sizeof(int16_t), ctx->setup_stream));
    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_status, (size_t)b->n_images * sizeof(int32_t)));
    MJ_HIP(ctx, hipMemsetAsync(p->d_status, 0, (size_t)b->n_images * sizeof(int32_t), ctx->setup_stream));
    if (b->flags & MJ_FLAG_KEEP_PLANES) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_planes, (size_t)rgb * sizeof(int16_t) + 16));
    if (b->flags & MJ_FLAG_KEEP_IDCT) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_idct, (size_t)blk * 64 * sizeof(int16_t) + 16));
    // The clears above run on the setup stream, which neither the context stream nor a caller's stream waits for: the
    // plan's first use waits (on the host) for this event, or the tail of the 6 GB clear could land after the first
    // blocks the first execute writes.  Not waiting here lets a serving loop create the next batch's plan while this
    // context's stream is still busy with the current batch.
    MJ_HIP(ctx, hipEventCreateWithFlags(&p->ready, hipEventDisableTiming));
    MJ_HIP(ctx, hipEventRecord(p->ready, ctx->setup_stream));
    MJ_HIP(ctx, hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    guard.p = nullptr;
    *out = p;
    return MJ_OK;
}

int mj_plan_stage1_form(const mj_plan *p) {
    if (!p) return MJ_ERR_INVALID;
    if (p->progressive) return MJ_FORM_SCANS;
    const int base = p->use_sync ? MJ_FORM_SYNC : (p->use_lanes ? MJ_FORM_LANES : MJ_FORM_WAVE);
    return base | (p->d_wg_tabs_lanes ? MJ_FORM_WG_TABLES : 0) | (p->use_lanes && p->d_lut13 ? MJ_FORM_RESOLVED : 0) | (p->use_fused ? MJ_FORM_FUSED : 0);
}

int mj_plan_get_info(const mj_plan *p, mj_plan_info *info) {
    if (!p || !info) return MJ_ERR_INVALID;
    *info = p->info;
    return MJ_OK;
}

int mj_plan_image_offsets(const mj_plan *p, int32_t image, int64_t *block_off, int64_t *rgb_off) {
    if (!p || image < 0 || image >= p->n_images) return MJ_ERR_INVALID;
    if (block_off) *block_off = p->h_images[image].block_off;
    if (rgb_off) *rgb_off = p->h_images[image].rgb_off;
    return MJ_OK;
}

// The plan's buffers are usable once the creation-time uploads and clears are done (see mj_plan_create): the host waits
// for them here — except in the plan's first execute, whose stream waits instead (the host goes on).
static int plan_ready(mj_plan *p, hipStream_t first_use = nullptr) {
    if (p->ready && !p->ready_done) {
        if (first_use && !p->ready_stream) {
            MJ_HIP(p->ctx, hipStreamWaitEvent(first_use, p->ready, 0));
            p->ready_stream = first_use;
            return MJ_OK;
        }
        if (first_use && first_use == p->ready_stream) return MJ_OK;       // same stream: ordered behind the wait above
        MJ_HIP(p->ctx, hipEventSynchronize(p->ready));
        p->ready_done = true;
    }
    return MJ_OK;
}

static int mark_done(mj_plan *p, hipStream_t s) {
    MJ_HIP(p->ctx, hipEventRecord(p->done, s));
    p->done_valid = true;
    return MJ_OK;
}

static int stage1_impl(mj_plan *p, void *stream);
static int stage2_impl(mj_plan *p, void *stream, uint8_t *rgb_device);

int mj_plan_execute_stage1(mj_plan *p, void *stream) {
    if (!p) return MJ_ERR_INVALID;
    const int rc = stage1_impl(p, stream);
    return rc != MJ_OK ? rc : mark_done(p, stream ? (hipStream_t)stream : p->ctx->stream);
}

int mj_plan_execute_stage2(mj_plan *p, void *stream, uint8_t *rgb_device) {
    if (!p) return MJ_ERR_INVALID;
    const int rc = stage2_impl(p, stream, rgb_device);
    return rc != MJ_OK ? rc : mark_done(p, stream ? (hipStream_t)stream : p->ctx->stream);
}

static int stage1_impl(mj_plan *p, void *stream) {
    mj_context *ctx = p->ctx;
    if (int rc = plan_ready(p, stream ? (hipStream_t)stream : ctx->stream)) return rc;
    if (!p->d_blob) return fail(ctx, MJ_ERR_INVALID, "plan has no entropy-coded data (stage 1 unavailable)");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MJ_HIP(ctx, mj::launch_fill_words(s, p->d_status, 0u, p->n_images));
    if (p->progressive) {
        // scans accumulate into the coefficient store (:1029, :1225): start from zeros, then one launch per scan ordinal
        MJ_HIP(ctx, hipMemsetAsync(p->d_coef, 0, (size_t)p->info.total_blocks * 128, s));
        // The scans are pipelined over bands of MCU rows: in launch number `step` a scan of dependency level L does band
        // step - L, so a refining scan follows one band behind what it refines instead of waiting for the whole scan.
        MJ_HIP(ctx, hipMemsetAsync(p->d_pstates, 0xFF, (size_t)p->n_psegs * sizeof(mj::DevProgState), s));
        if (p->n_split) MJ_HIP(ctx, hipMemsetAsync(p->d_psubs, 0xFF, (size_t)p->n_split * 2 * mj::kProgSub * sizeof(mj::DevProgSub), s));
        const bool fast = p->prog_fast;
        const int spec = (p->flags & MJ_FLAG_SPEC_REFINE) ? 1 : 0, tr = p->transposed ? 1 : 0;
        if (fast)       // stage 0 for every segment: what progressive_fast.hip's walks read
            MJ_HIP(ctx, mj::launch_destuff_pieces(s, p->d_blob, p->d_prog_dsegs, p->d_pieces, p->n_pieces, p->d_piece_kept, p->d_stream, p->d_seg_bits));
        if (p->prog_banded) {
            for (int step = 0; step < p->prog_steps; ++step) {
                if (fast)
                    MJ_HIP(ctx, mj::launch_progressive_fast(s, p->d_stream, p->d_seg_bits, p->d_psegs, (int)p->prog_rest_off, p->d_pscans, p->d_images,
                                                            p->d_huff, p->d_lut11p, p->d_coef, p->d_status, spec, tr, p->d_pstates, step,
                                                            p->prog_rows_per_band, (int)p->n_split, p->d_psubs, p->prog_parts));
                const int64_t r0 = fast ? p->prog_rest_off : 0;
                MJ_HIP(ctx, mj::launch_progressive_scan(s, p->d_blob, p->d_psegs + r0, (int)(p->n_psegs - r0), p->d_pscans, p->d_images, p->d_huff,
                                                        p->d_coef, p->d_status, spec | (fast ? 2 : 0), tr, p->d_pstates + r0, step,
                                                        p->prog_rows_per_band));
            }
        } else {        // one launch per dependency level over that level's segments (one band = the whole image)
            for (size_t o = 0; o + 1 < p->ordinal_seg_off.size(); ++o) {
                const int64_t s0 = p->ordinal_seg_off[o], s1 = p->ordinal_seg_off[o + 1];
                if (!fast) {
                    MJ_HIP(ctx, mj::launch_progressive_scan(s, p->d_blob, p->d_psegs + s0, (int)(s1 - s0), p->d_pscans, p->d_images, p->d_huff,
                                                            p->d_coef, p->d_status, spec, tr, p->d_pstates + s0, (int)o, 0));
                    continue;
                }
                const int64_t r0 = p->ordinal_kind_off[o * 4 + 3];      // DC first | AC first | AC refining | the rest
                MJ_HIP(ctx, mj::launch_progressive_fast(s, p->d_stream, p->d_seg_bits, p->d_psegs + s0, (int)(r0 - s0), p->d_pscans, p->d_images,
                                                        p->d_huff, p->d_lut11p, p->d_coef, p->d_status, spec, tr, p->d_pstates + s0, 0, 0));
                MJ_HIP(ctx, mj::launch_progressive_scan(s, p->d_blob, p->d_psegs + r0, (int)(s1 - r0), p->d_pscans, p->d_images, p->d_huff,
                                                        p->d_coef, p->d_status, spec | 2, tr, p->d_pstates + r0, (int)o, -1));
            }
        }
        return MJ_OK;
    }
    if (p->blob_src) MJ_HIP(ctx, hipMemcpyAsync(p->d_blob_owned, p->blob_src, (size_t)p->blob_src_len, hipMemcpyDeviceToDevice, s));
    if (p->n_jobs)      // restart markers and the end of each scan, found on the GPU
        MJ_HIP(ctx, mj::launch_scan_markers(s, p->d_blob, p->d_jobs, p->n_jobs, p->d_segs, p->d_status));
    if (p->use_lanes) {
        if (p->d_pieces)
            MJ_HIP(ctx, mj::launch_destuff_pieces(s, p->d_blob, p->d_segs, p->d_pieces, p->n_pieces, p->d_piece_kept, p->d_stream, p->d_seg_bits));
        else
            MJ_HIP(ctx, mj::launch_destuff(s, p->d_blob, p->d_segs, p->n_segs, p->d_stream, p->d_seg_bits));
        if (p->use_sync) {
            // round 0 guesses, round 1.. start every chunk from its predecessor's exit state until no exit state changes
            // (typically the second true-state round changes nothing), then the pieces are decoded like restart segments
            const int cbits = p->sync_chunk_bytes * 8;
            MJ_HIP(ctx, mj::launch_fill_words(s, p->d_couts, 0xFFFFFFFFu, p->n_chunks * (int64_t)(sizeof(mj::DevChunkOut) / 4)));
            MJ_HIP(ctx, mj::launch_sync_count(s, p->d_stream, p->d_seg_bits, p->d_segs, p->d_images, p->d_huff, p->d_lut11u, p->n_huff,
                                              p->d_chunks, p->n_chunks, cbits, nullptr, p->d_stateA, p->d_couts, p->d_changed, p->d_wg_tabs_count, p->wg_slots_count, nullptr, p->sync_warm_bits));
            uint64_t *in = p->d_stateA, *out = p->d_stateB;
            // repair rounds: a fixed number, queued without looking (a chain of wrongly guessed entry states gets one link
            // shorter per round; after round 0's run-up nearly every guess is right and the second repair round changes
            // nothing).  Whether they sufficed is decided on the device: k_build_vsegs marks the images whose chunk states
            // had not settled (MJ_ST_UNCONVERGED) and the caller decodes those again with MJ_FLAG_NO_SYNC.  No host
            // round trip: the execute is asynchronous and can be captured into a graph like every other form.
            MJ_HIP(ctx, mj::launch_fill_words(s, p->d_changed, 0u, p->sync_rounds + 8));
            for (int round = 1; round <= p->sync_rounds; ++round) {
                MJ_HIP(ctx, mj::launch_sync_count(s, p->d_stream, p->d_seg_bits, p->d_segs, p->d_images, p->d_huff, p->d_lut11u,
                                                  p->n_huff, p->d_chunks, p->n_chunks, cbits, in, out, p->d_couts, p->d_changed + round, p->d_wg_tabs_count, p->wg_slots_count,
                                                  round >= 2 ? p->d_changed + round - 1 : nullptr, p->sync_warm_bits));
                std::swap(in, out);
            }
            MJ_HIP(ctx, mj::launch_build_vsegs(s, p->d_chunks, p->n_chunks, p->d_couts, p->d_segs, p->d_seg_bits, p->d_images, p->d_vsegs,
                                               in, p->d_status));
            if (p->d_lut13)
                MJ_HIP(ctx, mj::launch_huffman_lanes13(s, p->d_stream, p->d_seg_bits, p->d_segs, p->n_chunks, p->d_images, p->d_huff, p->d_lut11,
                                                       p->d_lut13, p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk,
                                                       p->d_coef, p->d_status, p->transposed ? 1 : 0, p->d_vsegs));
            else
                MJ_HIP(ctx, mj::launch_huffman_lanes(s, p->d_stream, p->d_seg_bits, p->d_segs, p->n_chunks, p->d_images, p->d_huff, p->d_lut11,
                                                     p->n_huff, p->d_coef, p->d_status, p->transposed ? 1 : 0, p->d_vsegs, p->d_wg_tabs_lanes, p->wg_slots_lanes));
            return MJ_OK;
        }
        if (p->d_lut13)
            MJ_HIP(ctx, mj::launch_huffman_lanes13(s, p->d_stream, p->d_seg_bits, p->d_segs, p->n_segs, p->d_images, p->d_huff, p->d_lut11,
                                                   p->d_lut13, p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk,
                                                   p->d_coef, p->d_status, p->transposed ? 1 : 0, nullptr, p->d_by_length, p->seg_order_mode));
        else
            MJ_HIP(ctx, mj::launch_huffman_lanes(s, p->d_stream, p->d_seg_bits, p->d_segs, p->n_segs, p->d_images, p->d_huff, p->d_lut11,
                                                 p->n_huff, p->d_coef, p->d_status, p->transposed ? 1 : 0, nullptr, p->d_wg_tabs_lanes, p->wg_slots_lanes));
    } else
        MJ_HIP(ctx, mj::launch_huffman(s, p->d_blob, p->d_segs, p->n_segs, p->d_images, p->d_huff, p->d_coef,
                                       p->d_status, p->lut_slots, p->transposed ? 1 : 0));
    return MJ_OK;
}

// what every stage-2 launch of the plan is told (the output buffer resolved, allocated on first use)
static int recon_args(mj_plan *p, uint8_t *rgb_device, mj::ReconArgs &a) {
    mj_context *ctx = p->ctx;
    if (!rgb_device) {
        if (!p->d_rgb) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_rgb, (size_t)p->info.rgb_bytes + 16));
        rgb_device = p->d_rgb;
    }
    p->last_rgb = rgb_device;
    a = mj::ReconArgs{};
    a.images = p->d_images; a.n_images = p->n_images; a.mcu_prefix = p->d_mcu_prefix;
    a.total_mcus = p->info.total_mcus; a.coef = p->d_coef; a.qt = p->d_qt; a.idct_tt = ctx->d_idct_tt;
    a.up_taps = nullptr; a.rgb = rgb_device; a.dump = ctx->d_dump; a.planes = p->d_planes; a.idct_out = p->d_idct;
    a.layout = p->layout & 1; a.exact_only = (p->flags & MJ_FLAG_EXACT_ONLY) ? 1 : 0;
#ifdef MJ_DIAGNOSTIC      // phase ablations of the diagnostic build (make DIAG=1); the product never looks at the environment here
    a.debug = getenv("MJ_DEBUG_STAGE2") ? atoi(getenv("MJ_DEBUG_STAGE2")) : 0;
    a.debug_mask = getenv("MJ_DEBUG_MASK") ? atoi(getenv("MJ_DEBUG_MASK")) : 0;
#else
    a.debug = 0;
#endif
    a.uniform_geometry = p->uniform ? 1 : 0; a.mcus_per_image = p->mcus_per_image;
    a.work_counter = reinterpret_cast<uint32_t *>(p->d_job_prefix + p->n_images + 1); a.chunk_strips = p->chunk_strips;
    a.jobs_per_ticket = p->jobs_per_ticket;
    a.level_counts = reinterpret_cast<unsigned long long *>(p->d_job_prefix + p->n_images + 3);
    return MJ_OK;
}

static int stage2_impl(mj_plan *p, void *stream, uint8_t *rgb_device) {
    if (int rc = plan_ready(p, stream ? (hipStream_t)stream : p->ctx->stream)) return rc;
    mj_context *ctx = p->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    mj::ReconArgs a{};
    if (int rc = recon_args(p, rgb_device, a)) return rc;
    rgb_device = p->last_rgb;
    // planar layouts: the kernels write the interleaved image of the same orientation into a plan-owned buffer and a copy
    // kernel separates the components (one extra pass over the pixels; the interleaved layouts are the fast ones)
    const bool planar = p->layout >= MJ_LAYOUT_PLANAR_XMAJOR && p->ncomp == 3;
    if (planar) {
        if (!p->d_rgb_tmp) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_rgb_tmp, (size_t)p->info.rgb_bytes + 16));
        a.rgb = p->d_rgb_tmp;
    }
    if (a.planes || a.idct_out) MJ_HIP(ctx, mj::launch_fill_words(s, a.level_counts, 0u, 6));     // mj_plan_idct_levels
    // the launch's ticket counter starts from zero whatever an earlier launch left behind (one that was aborted never drew its
    // last ticket).  The kernel still resets it itself at its end: the word is per PLAN, so a plan's executes must not overlap
    // (mijpeg.h) — two plans, or one plan's executes one after the other on any streams, are fine.
    if (!p->generic && !a.exact_only) MJ_HIP(ctx, mj::launch_fill_words(s, a.work_counter, 0u, 1));
    if (p->generic) {
        MJ_HIP(ctx, mj::launch_reconstruct_generic(s, a));
    } else if (a.exact_only) {
        MJ_HIP(ctx, mj::launch_reconstruct(s, a, p->hmax, p->vmax, p->ncomp));
    } else {
        // row-major plans: the same kernel on the transposed problem, whose x-major output IS the row-major image
        MJ_HIP(ctx, mj::launch_reconstruct_fast(s, a, p->hmax, p->vmax, p->ncomp, p->transposed, p->d_job_prefix,
                                               p->total_jobs, p->jobs_per_image));
    }
    if (planar) MJ_HIP(ctx, mj::launch_planes_from_interleaved(s, p->d_images, p->n_images, p->max_pixels, p->d_rgb_tmp, rgb_device));
    return MJ_OK;
}

// Both stages in one launch (fused.hip), for the plans that can (use_fused): the marker scan and stage 0 as in stage1_impl,
// then producers and consumers side by side.
static int fused_impl(mj_plan *p, void *stream, uint8_t *rgb_device) {
    mj_context *ctx = p->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (int rc = plan_ready(p, s)) return rc;
    mj::ReconArgs a{};
    if (int rc = recon_args(p, rgb_device, a)) return rc;
    MJ_HIP(ctx, mj::launch_fill_words(s, p->d_status, 0u, p->n_images));
    if (p->n_jobs) MJ_HIP(ctx, mj::launch_scan_markers(s, p->d_blob, p->d_jobs, p->n_jobs, p->d_segs, p->d_status));
    MJ_HIP(ctx, mj::launch_destuff(s, p->d_blob, p->d_segs, p->n_segs, p->d_stream, p->d_seg_bits));
#ifdef MJ_DIAGNOSTIC
    const bool dbg_fused = getenv("MJ_DEBUG_FUSED") != nullptr;
    if (dbg_fused) { (void)hipStreamSynchronize(s); mj::dbg_fused_clear(ctx->d_dump); }
#endif
    MJ_HIP(ctx, mj::launch_fused(s, p->fused, p->d_stream, p->d_seg_bits, p->d_segs, p->n_segs, p->d_images, p->d_huff, p->d_lut11, p->d_lut12,
                                 p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk, p->d_coef, p->d_status, a, p->hmax, p->vmax,
                                 p->fused_spi, p->d_job_prefix, p->total_jobs, p->jobs_per_image));
#ifdef MJ_DIAGNOSTIC
    if (dbg_fused) {
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[diag fused] shape: %d images per workgroup, %d producers x %d lanes, %d consumers beside them\n", p->fused.ipw, p->fused.n_prod, p->fused.lpw, p->fused.n_cons);
        mj::dbg_fused_report(ctx->d_dump, (p->n_images + p->fused.ipw - 1) / p->fused.ipw);
    }
#endif
    return MJ_OK;
}

// one execute's launches: fused where the plan can, else stage 1 then stage 2
static int execute_impl(mj_plan *p, void *stream, uint8_t *rgb_device) {
    if (p->use_fused && p->d_blob) return fused_impl(p, stream, rgb_device);
    int rc = stage1_impl(p, stream);
    if (rc == MJ_OK) rc = stage2_impl(p, stream, rgb_device);
    return rc;
}

int mj_plan_execute(mj_plan *p, void *stream, uint8_t *rgb_device) {
    if (!p) return MJ_ERR_INVALID;
    if (int rc = plan_ready(p, stream ? (hipStream_t)stream : p->ctx->stream)) return rc;
    mj_context *ctx = p->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // Re-executions of a plan with the same stream and output buffer replay a captured graph of its launches (one
    // submission instead of a memset and three or four kernel launches).  The first execute runs plainly (it also
    // sizes grids and allocates the plan-owned output).  Progressive plans (dozens of launches, latency bound) launch plainly.
    const bool graphable = !p->progressive && p->d_blob && p->executed_once && (rgb_device || p->d_rgb) &&
                           !ctx->no_graph;
    if (graphable && p->graph_exec && p->graph_stream == s && p->graph_rgb == (rgb_device ? rgb_device : p->d_rgb)) {
        p->last_rgb = p->graph_rgb;
        p->last_was_graph = true;
        MJ_HIP(ctx, hipGraphLaunch(p->graph_exec, s));
        return mark_done(p, s);
    }
    // capture only what is evidently a loop: the previous execute used this very stream and buffer (a caller that
    // alternates output buffers keeps launching plainly instead of re-capturing every time)
    uint8_t *want_rgb = rgb_device ? rgb_device : p->d_rgb;
    if (graphable && s != nullptr && p->prev_stream == s && p->prev_rgb == want_rgb) {
        if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = execute_impl(p, s, rgb_device);
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (rc == MJ_OK && ce == hipSuccess && g && hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                (void)hipGraphDestroy(g);
                p->graph_stream = s;
                p->graph_rgb = p->last_rgb;
                p->last_was_graph = true;
                MJ_HIP(ctx, hipGraphLaunch(p->graph_exec, s));
                return mark_done(p, s);
            }
            if (g) (void)hipGraphDestroy(g);
            p->graph_exec = nullptr;
            (void)hipGetLastError();
        }
    }
    const int rc = execute_impl(p, stream, rgb_device);
    p->executed_once = rc == MJ_OK;
    p->last_was_graph = false;
    p->prev_stream = s;
    p->prev_rgb = p->last_rgb;
    return rc != MJ_OK ? rc : mark_done(p, s);
}

int mj_plan_sync(mj_plan *p) {
    if (!p) return MJ_ERR_INVALID;
    // this plan's latest execute, on whichever stream it went; not the rest of that stream (in a serving loop the next
    // batch's kernels are queued behind it)
    if (p->done_valid) MJ_HIP(p->ctx, hipEventSynchronize(p->done));
    return MJ_OK;
}

int mj_plan_device_buffers(mj_plan *p, int16_t **coef, uint8_t **rgb, int16_t **planes, int16_t **idct) {
    if (!p) return MJ_ERR_INVALID;
    if (int rc = plan_ready(p)) return rc;
    if (coef) *coef = p->d_coef;
    if (rgb) *rgb = p->last_rgb ? p->last_rgb : p->d_rgb;
    if (planes) *planes = p->d_planes;
    if (idct) *idct = p->d_idct;
    return MJ_OK;
}

int mj_plan_read(mj_plan *p, uint8_t *rgb_host, int16_t *coef_host, int16_t *planes_host, int16_t *idct_host,
                 int32_t *status_host) {
    if (!p) return MJ_ERR_INVALID;
    mj_context *ctx = p->ctx;
    if (p->done_valid) MJ_HIP(ctx, hipEventSynchronize(p->done));
    if (rgb_host) {
        if (!p->last_rgb) return fail(ctx, MJ_ERR_INVALID, "mj_plan_read: nothing executed yet");
        MJ_HIP(ctx, hipMemcpy(rgb_host, p->last_rgb, (size_t)p->info.rgb_bytes, hipMemcpyDeviceToHost));
    }
    if (coef_host) {   // the :869 seam is in zig-zag order; the device keeps blocks in natural order
        if (!p->d_tmp_coef) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_tmp_coef, (size_t)p->info.total_blocks * 128 + 16));
        MJ_HIP(ctx, mj::launch_permute_blocks(ctx->stream, p->d_coef, p->d_tmp_coef, p->info.total_blocks, 0, p->transposed ? 1 : 0));
        MJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MJ_HIP(ctx, hipMemcpy(coef_host, p->d_tmp_coef, (size_t)p->info.total_blocks * 128, hipMemcpyDeviceToHost));
    }
    if (planes_host) {
        if (!p->d_planes) return fail(ctx, MJ_ERR_INVALID, "mj_plan_read: plan was created without MJ_FLAG_KEEP_PLANES");
        MJ_HIP(ctx, hipMemcpy(planes_host, p->d_planes, (size_t)p->info.rgb_bytes * 2, hipMemcpyDeviceToHost));
    }
    if (idct_host) {
        if (!p->d_idct) return fail(ctx, MJ_ERR_INVALID, "mj_plan_read: plan was created without MJ_FLAG_KEEP_IDCT");
        MJ_HIP(ctx, hipMemcpy(idct_host, p->d_idct, (size_t)p->info.total_blocks * 128, hipMemcpyDeviceToHost));
    }
    if (status_host) MJ_HIP(ctx, hipMemcpy(status_host, p->d_status, (size_t)p->n_images * 4, hipMemcpyDeviceToHost));
    return MJ_OK;
}

int mj_plan_write_coef(mj_plan *p, const int16_t *coef, int32_t mem) {
    if (!p || !coef) return MJ_ERR_INVALID;
    if (int rc = plan_ready(p)) return rc;
    mj_context *ctx = p->ctx;
    const int16_t *src = coef;
    if (mem != MJ_MEM_DEVICE) {
        if (!p->d_tmp_coef) MJ_HIP(ctx, ctx->cache.get((void **)&p->d_tmp_coef, (size_t)p->info.total_blocks * 128 + 16));
        MJ_HIP(ctx, hipMemcpy(p->d_tmp_coef, coef, (size_t)p->info.total_blocks * 128, hipMemcpyHostToDevice));
        src = p->d_tmp_coef;
    }
    MJ_HIP(ctx, hipDeviceSynchronize());
    MJ_HIP(ctx, mj::launch_permute_blocks(ctx->stream, src, p->d_coef, p->info.total_blocks, 1, p->transposed ? 1 : 0));   // zig-zag -> store order
    MJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MJ_OK;
}

int mj_plan_fill_coef(mj_plan *p, int byte_value) {
    if (!p) return MJ_ERR_INVALID;
    if (int rc = plan_ready(p)) return rc;
    mj_context *ctx = p->ctx;
    if (p->done_valid) MJ_HIP(ctx, hipEventSynchronize(p->done));
    if (getenv("MJ_DEBUG_FILL")) fprintf(stderr, "[fill] coef %p + %zu = %p, status %p, n_images %d\n", (void *)p->d_coef, (size_t)p->info.total_blocks * 128,
                                         (void *)((char *)p->d_coef + (size_t)p->info.total_blocks * 128), (void *)p->d_status, p->n_images);
    MJ_HIP(ctx, hipMemset(p->d_coef, byte_value & 0xFF, (size_t)p->info.total_blocks * 128));
    MJ_HIP(ctx, hipDeviceSynchronize());
    return MJ_OK;
}

int mj_decode_baseline_batch(mj_context *ctx, const mj_batch *batch, uint8_t *rgb_out, int16_t *coef_out,
                             int32_t *status_out) {
    mj_plan *p = nullptr;
    int rc = mj_plan_create(ctx, batch, &p);
    if (rc != MJ_OK) return rc;
    rc = mj_plan_execute(p, nullptr, nullptr);
    if (rc == MJ_OK) rc = mj_plan_sync(p);
    if (rc == MJ_OK) rc = mj_plan_read(p, rgb_out, coef_out, nullptr, nullptr, status_out);
    mj_plan_destroy(p);
    return rc;
}

int mj_idct_batch(mj_context *ctx, const mj_batch *batch, const int16_t *coef, uint8_t *rgb_out) {
    if (!batch || !coef) return fail(ctx, MJ_ERR_INVALID, "mj_idct_batch: NULL argument");
    mj_batch b = *batch;
    b.blob = nullptr; b.blob_mem = MJ_MEM_NONE;
    mj_plan *p = nullptr;
    int rc = mj_plan_create(ctx, &b, &p);
    if (rc != MJ_OK) return rc;
    rc = mj_plan_write_coef(p, coef, MJ_MEM_HOST);
    if (rc == MJ_OK) rc = mj_plan_execute_stage2(p, nullptr, nullptr);
    if (rc == MJ_OK) rc = mj_plan_sync(p);
    if (rc == MJ_OK) rc = mj_plan_read(p, rgb_out, nullptr, nullptr, nullptr, nullptr);
    mj_plan_destroy(p);
    return rc;
}

int mj_plan_idct_levels(mj_plan *p, uint64_t counts[3]) {
    if (!p || !counts) return MJ_ERR_INVALID;
    mj_context *ctx = p->ctx;
    if (p->done_valid) MJ_HIP(ctx, hipEventSynchronize(p->done));
    MJ_HIP(ctx, hipMemcpy(counts, p->d_job_prefix + p->n_images + 3, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MJ_OK;
}

int mj_plan_time_execute(mj_plan *p, int iters, uint8_t *rgb_device, float *front_ms, float *main_ms) {
    if (!p || iters <= 0) return MJ_ERR_INVALID;
    mj_context *ctx = p->ctx;
    hipStream_t s = ctx->stream;
    if (!p->d_blob) return fail(ctx, MJ_ERR_INVALID, "plan has no entropy-coded data");
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    MJ_HIP(ctx, hipEventCreate(&ev.a));
    MJ_HIP(ctx, hipEventCreate(&ev.b));
    if (int rc = plan_ready(p, s)) return rc;
    float ms = 0.f;
    if (front_ms) *front_ms = 0.f;
    if (main_ms) *main_ms = 0.f;
    if (!p->use_fused) {                       // the two launches: front = stage 1 (with stage 0), main = stage 2
        return mj_plan_time_stages(p, iters, rgb_device, front_ms, main_ms);
    }
    mj::ReconArgs a{};
    if (int rc = recon_args(p, rgb_device, a)) return rc;
    auto front = [&]() -> int {
        MJ_HIP(ctx, mj::launch_fill_words(s, p->d_status, 0u, p->n_images));
        if (p->n_jobs) MJ_HIP(ctx, mj::launch_scan_markers(s, p->d_blob, p->d_jobs, p->n_jobs, p->d_segs, p->d_status));
        MJ_HIP(ctx, mj::launch_destuff(s, p->d_blob, p->d_segs, p->n_segs, p->d_stream, p->d_seg_bits));
        return MJ_OK;
    };
    auto fused = [&]() -> int {
        MJ_HIP(ctx, mj::launch_fused(s, p->fused, p->d_stream, p->d_seg_bits, p->d_segs, p->n_segs, p->d_images, p->d_huff, p->d_lut11, p->d_lut12,
                                     p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk, p->d_coef, p->d_status, a, p->hmax, p->vmax,
                                     p->fused_spi, p->d_job_prefix, p->total_jobs, p->jobs_per_image));
        return MJ_OK;
    };
    int rc = front();
    if (rc == MJ_OK) rc = fused();
    if (rc != MJ_OK) return rc;
    MJ_HIP(ctx, hipEventRecord(ev.a, s));
    for (int i = 0; i < iters && rc == MJ_OK; ++i) rc = front();
    MJ_HIP(ctx, hipEventRecord(ev.b, s));
    MJ_HIP(ctx, hipEventSynchronize(ev.b));
    MJ_HIP(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
    if (front_ms) *front_ms = ms / iters;
    if (rc != MJ_OK) return rc;
    MJ_HIP(ctx, hipEventRecord(ev.a, s));
    for (int i = 0; i < iters && rc == MJ_OK; ++i) rc = fused();
    MJ_HIP(ctx, hipEventRecord(ev.b, s));
    MJ_HIP(ctx, hipEventSynchronize(ev.b));
    MJ_HIP(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
    if (main_ms) *main_ms = ms / iters;
    return rc != MJ_OK ? rc : mark_done(p, s);
}

int mj_plan_time_stages(mj_plan *p, int iters, uint8_t *rgb_device, float *stage1_ms, float *stage2_ms) {
    if (!p || iters <= 0) return MJ_ERR_INVALID;
    mj_context *ctx = p->ctx;
    hipStream_t s = ctx->stream;
    struct Events {           // destroyed on every way out, the error returns of MJ_HIP included
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    MJ_HIP(ctx, hipEventCreate(&ev.a));
    MJ_HIP(ctx, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    int rc = MJ_OK;
    float ms = 0.f;
    if (stage1_ms) {
        *stage1_ms = 0.f;
        if (p->d_blob) {
            if ((rc = stage1_impl(p, s)) != MJ_OK) return rc;   // warm
            MJ_HIP(ctx, hipEventRecord(e0, s));
            for (int i = 0; i < iters && rc == MJ_OK; ++i) rc = stage1_impl(p, s);
            MJ_HIP(ctx, hipEventRecord(e1, s));
            MJ_HIP(ctx, hipEventSynchronize(e1));
            MJ_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            *stage1_ms = ms / iters;
#ifdef MJ_DIAGNOSTIC
            mj::dbg_lanes_report();
            mj::dbg_prog_report();
            if (getenv("MJ_DEBUG_STAGE1_WAVES")) mj::dbg_lanes13_waves_report();
            if (getenv("MJ_DEBUG_PROG_STEP")) mj::dbg_prog_waves_report();
#endif
        }
    }
    if (stage2_ms && rc == MJ_OK) {
        if ((rc = stage2_impl(p, s, rgb_device)) != MJ_OK) return rc;   // warm
        MJ_HIP(ctx, hipEventRecord(e0, s));
        for (int i = 0; i < iters && rc == MJ_OK; ++i) rc = stage2_impl(p, s, rgb_device);
        MJ_HIP(ctx, hipEventRecord(e1, s));
        MJ_HIP(ctx, hipEventSynchronize(e1));
        MJ_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        *stage2_ms = ms / iters;
#ifdef MJ_DIAGNOSTIC      // the wait probe of the diagnostic build (MJ_DEBUG_STAGE2=8/9): sums the kernel left in the dump buffer
        if (getenv("MJ_DEBUG_STAGE2") && atoi(getenv("MJ_DEBUG_STAGE2")) >= 8) {
            unsigned long long h[3] = {0, 0, 0};
            (void)hipMemcpy(h, ctx->d_dump + 393216 * 8, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "[mijpeg diag] wait cycles %.4g of kernel cycles %.4g per wave (%llu waves) = %.1f %%\n", (double)h[0] / (double)(h[2] ? h[2] : 1),
                    (double)h[1] / (double)(h[2] ? h[2] : 1), h[2], 100.0 * (double)h[0] / (double)(h[1] ? h[1] : 1));
            unsigned long long ph[6];
            (void)hipMemcpy(ph, ctx->d_dump + (393216 + 8) * 8, sizeof(ph), hipMemcpyDeviceToHost);
            double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)ph[i];
            if (tot > 0) fprintf(stderr, "[mijpeg diag] phase shares: rounds %.1f %%, level3+next fetch %.1f %%, pixels %.1f %%, staging+stores %.1f %%, slow paths %.1f %%, loop head %.1f %%\n",
                                 100 * ph[0] / tot, 100 * ph[1] / tot, 100 * ph[2] / tot, 100 * ph[3] / tot, 100 * ph[4] / tot, 100 * ph[5] / tot);
            (void)hipMemset(ctx->d_dump + 393216 * 8, 0, 128);
            if (atoi(getenv("MJ_DEBUG_STAGE2")) == 11) {      // start / end of every wave of the last launch (1024 workgroups x 4)
                std::vector<unsigned long long> t(1024 * 4 * 4);
                (void)hipMemcpy(t.data(), ctx->d_dump + (2u << 20), t.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull;
                for (size_t i = 0; i < t.size(); i += 4) if (t[i] && t[i] < t0) t0 = t[i];
                std::vector<double> st, en;
                FILE *f = getenv("MJ_DEBUG_WAVES_CSV") ? fopen(getenv("MJ_DEBUG_WAVES_CSV"), "w") : nullptr;
                if (f) fprintf(f, "block,wave,start_us,end_us,hw_id,xcc_id\n");
                for (size_t i = 0; i < t.size(); i += 4) if (t[i]) {
                    st.push_back((double)(t[i] - t0) * 0.01); en.push_back((double)(t[i + 1] - t0) * 0.01);
                    if (f) fprintf(f, "%zu,%zu,%.2f,%.2f,%llu,%llu\n", i / 16, (i / 4) % 4, st.back(), en.back(), t[i + 2] & 0xFFFFFFFFull, t[i + 2] >> 32);
                }
                if (f) fclose(f);
                std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
                auto q = [](const std::vector<double> &v, double f) { return v.empty() ? 0.0 : v[(size_t)(f * (v.size() - 1))]; };
                fprintf(stderr, "[mijpeg diag] %zu waves; start us: max %.1f; end us: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f\n", st.size(),
                        q(st, 1), q(en, 0), q(en, 0.1), q(en, 0.5), q(en, 0.9), q(en, 0.99), q(en, 1));
                (void)hipMemset(ctx->d_dump + (2u << 20), 0, t.size() * 8);
            }
        }
#endif
    }
    return rc;
}

}  // extern "C"
