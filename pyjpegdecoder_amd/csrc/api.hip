// C ABI of libmijpeg.so: context, plan (upload once / execute many), one-shot helpers.
// Host-side only; the kernels are in huffman.hip and reconstruct.hip.
#include <math.h>

#include <mutex>

#include "plan.h"

std::string g_create_err;

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
    MJ_HIP(nullptr, hipEventCreateWithFlags(&ctx->fused_done, hipEventDisableTiming));
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
    if (ctx->fused_done) (void)hipEventDestroy(ctx->fused_done);
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
    void *ptrs[] = {p->d_blob_owned, p->d_segs, p->d_images, p->d_huff, p->d_lut11, p->d_lut13, p->d_lut12, p->d_by_length, p->d_holder, p->d_xwords, p->d_wg_tabs_lanes, p->d_wg_tabs_count, p->d_stream, p->d_seg_bits, p->d_jobs, p->d_lut11u, p->d_acsegs, p->d_pc_chunks, p->d_pc_tabs, p->d_pc_exit, p->d_pc_outs, p->d_pc_items, p->d_pc_owner, p->d_pc_vsegs, p->d_lutc, p->d_sync_items, p->d_seg_chunk0, p->d_chunks, p->d_stateA, p->d_stateB, p->d_couts, p->d_vsegs, p->d_changed, p->d_pieces, p->d_piece_kept, p->d_pscans, p->d_psegs, p->d_pstates, p->d_psubs, p->d_prog_dsegs, p->d_lut11p, p->d_qt, p->d_mcu_prefix, p->d_job_prefix, p->d_tmp_coef, p->d_coef,
                    p->d_rgb, p->d_rgb_tmp, p->d_planes, p->d_idct, p->d_status};
    for (void *q : ptrs)
        if (q) p->ctx->cache.put(q);
    delete p;
}

int mj_plan_stage1_form(const mj_plan *p) {
    if (!p) return MJ_ERR_INVALID;
    if (p->progressive) return MJ_FORM_SCANS | (p->prog_chunks ? MJ_FORM_COUNT_RESOLVED : 0);
    const int base = p->use_sync ? MJ_FORM_SYNC : (p->use_lanes ? MJ_FORM_LANES : MJ_FORM_WAVE);
    return base | (p->d_wg_tabs_lanes ? MJ_FORM_WG_TABLES : 0) | (p->use_lanes && p->d_lut13 ? MJ_FORM_RESOLVED : 0) | (p->use_fused ? MJ_FORM_FUSED : 0) |
           (p->use_sync && p->d_lutc ? MJ_FORM_COUNT_RESOLVED : 0);
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
        if (p->prog_chunks)     // the first AC scans, chunk by chunk, before the band pipeline starts
            MJ_HIP(ctx, mj::launch_progressive_chunks(s, p->d_stream, p->d_seg_bits, p->d_acsegs, p->n_acsegs, p->d_pc_tabs, p->d_lut11p, p->d_pc_chunks, p->n_pc_chunks,
                                                      p->pc_chunk_bytes * 8, p->d_pc_exit, p->d_pc_outs, p->d_pc_items, p->d_pc_owner + p->n_pc_chunks, p->d_pc_owner,
                                                      p->d_pc_vsegs, p->d_images, p->d_coef, p->d_status, tr, p->sync_rounds));
        if (p->prog_banded) {
            for (int step = 0; step < p->prog_steps; ++step) {
                if (fast)
                    MJ_HIP(ctx, mj::launch_progressive_fast(s, p->d_stream, p->d_seg_bits, p->d_psegs, (int)p->prog_rest_off, p->d_pscans, p->d_images,
                                                            p->d_huff, p->d_lut11p, p->d_coef, p->d_status, spec, tr, p->d_pstates, step,
                                                            p->prog_rows_per_band, (int)p->n_split, p->d_psubs, p->prog_parts));
                const int64_t r0 = fast ? p->prog_rest_off : 0;
                MJ_HIP(ctx, mj::launch_progressive_scan(s, p->d_blob, p->d_psegs + r0, (int)(p->n_psegs_wave - r0), p->d_pscans, p->d_images, p->d_huff,
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
            uint64_t *in = p->d_stateA, *out = p->d_stateB;
            if (p->d_lutc) {
                // resolved tables: one walk over every chunk, the list of wrong guesses, their repair — three launches
                MJ_HIP(ctx, mj::launch_count(s, p->d_stream, p->d_seg_bits, p->d_segs, p->d_images, p->d_lutc, p->lutc_tab_bytes, p->n_huff, p->lutc_bits,
                                             p->d_chunks, p->n_chunks, cbits, p->sync_warm_bits, p->d_stateA, p->d_couts, p->d_sync_items, p->d_changed, p->sync_rounds,
                                             reinterpret_cast<int32_t *>(p->d_stateB)));
            } else {
                MJ_HIP(ctx, mj::launch_fill_words(s, p->d_couts, 0xFFFFFFFFu, p->n_chunks * (int64_t)(sizeof(mj::DevChunkOut) / 4)));
                MJ_HIP(ctx, mj::launch_sync_count(s, p->d_stream, p->d_seg_bits, p->d_segs, p->d_images, p->d_huff, p->d_lut11u, p->n_huff,
                                                  p->d_chunks, p->n_chunks, cbits, nullptr, p->d_stateA, p->d_couts, p->d_changed, p->d_wg_tabs_count, p->wg_slots_count, nullptr, p->sync_warm_bits));
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
            }
            MJ_HIP(ctx, mj::launch_build_vsegs(s, p->d_chunks, p->d_seg_chunk0, p->n_segs, p->d_couts, p->d_segs, p->d_seg_bits, p->d_images, p->d_vsegs,
                                               in, cbits, p->d_status));
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
                                 p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk, p->lutf_off, p->lutf_bits, p->d_coef, p->d_status, a, p->hmax, p->vmax,
                                 p->transposed, p->fused_spi, p->h_images[0].restart_interval, p->h_images[0].mcu_count_h, p->h_images[0].mcu_count_v, p->d_job_prefix, p->total_jobs, p->jobs_per_image,
                                 p->d_by_length, p->d_holder, p->d_xwords));
#ifdef MJ_DIAGNOSTIC
    if (dbg_fused) {
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[diag fused] shape: %d workgroups, %d images each (0 = segments dealt out by length), %d producers x %d lanes, %d consumers beside them\n",
                p->fused.n_wg, p->fused.ipw, p->fused.n_prod, p->fused.lpw, p->fused.n_cons);
        mj::dbg_fused_report(ctx->d_dump, p->fused.n_wg);
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

static int execute_launches(mj_plan *p, void *stream, uint8_t *rgb_device);

int mj_plan_execute(mj_plan *p, void *stream, uint8_t *rgb_device) {
    if (!p) return MJ_ERR_INVALID;
    mj_context *ctx = p->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const bool fused = p->use_fused && p->d_blob;
    // fused launches of one context take turns (plan.h): outside any graph — the wait and the record bracket whatever this
    // execute submits, replayed or not
    if (fused && ctx->fused_stream && ctx->fused_stream != s) MJ_HIP(ctx, hipStreamWaitEvent(s, ctx->fused_done, 0));
    const int rc = execute_launches(p, stream, rgb_device);
    if (fused && rc == MJ_OK) {
        MJ_HIP(ctx, hipEventRecord(ctx->fused_done, s));
        ctx->fused_stream = s;
    }
    return rc;
}

static int execute_launches(mj_plan *p, void *stream, uint8_t *rgb_device) {
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
                                     p->n_ac13, p->n_dc13, p->ac_slot_pk, p->dc_slot_pk, p->dc_tab_pk, p->lutf_off, p->lutf_bits, p->d_coef, p->d_status, a, p->hmax, p->vmax,
                                     p->transposed, p->fused_spi, p->h_images[0].restart_interval, p->h_images[0].mcu_count_h, p->h_images[0].mcu_count_v, p->d_job_prefix, p->total_jobs, p->jobs_per_image,
                                 p->d_by_length, p->d_holder, p->d_xwords));
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

int mj_context_launch_clock(mj_context *ctx, float *shader_mhz, float *launch_ms) {
    if (!ctx) return MJ_ERR_INVALID;
    unsigned long long w[4] = {0, 0, 0, 0};
    MJ_HIP(ctx, hipSetDevice(ctx->device));
    MJ_HIP(ctx, hipMemcpy(w, ctx->d_dump + mj::kDumpClockWords, sizeof(w), hipMemcpyDeviceToHost));
    const bool have = w[3] > w[1] && w[2] > w[0];
    if (shader_mhz) *shader_mhz = have ? (float)((double)(w[2] - w[0]) / (double)(w[3] - w[1]) * 100.0) : 0.f;
    if (launch_ms) *launch_ms = have ? (float)((double)(w[3] - w[1]) * 1e-5) : 0.f;
    return MJ_OK;
}

int mj_device_copy_rate(mj_context *ctx, int64_t bytes, int iters, float *ms_per_copy) {
    if (!ctx || bytes < 16 || iters <= 0 || !ms_per_copy) return MJ_ERR_INVALID;
    MJ_HIP(ctx, hipSetDevice(ctx->device));
    bytes &= ~(int64_t)15;
    struct Bufs {
        mj_context *c; void *a = nullptr, *b = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Bufs() { if (a) c->cache.put(a); if (b) c->cache.put(b); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } h{ctx};
    MJ_HIP(ctx, ctx->cache.get(&h.a, (size_t)bytes));
    MJ_HIP(ctx, ctx->cache.get(&h.b, (size_t)bytes));
    MJ_HIP(ctx, hipEventCreate(&h.e0));
    MJ_HIP(ctx, hipEventCreate(&h.e1));
    hipStream_t s = ctx->stream;
    MJ_HIP(ctx, mj::launch_fill_words(s, h.a, 0x01020304u, bytes / 4));
    float best = 0.f;
    for (int v = 0; v < mj::copy16_variants(); ++v) {          // every launch shape: the best one is the ceiling (util_kernels.hip)
        for (int i = 0; i < 3; ++i) MJ_HIP(ctx, mj::launch_copy16(s, h.a, h.b, bytes, v));      // warm
        MJ_HIP(ctx, hipEventRecord(h.e0, s));
        for (int i = 0; i < iters; ++i) MJ_HIP(ctx, mj::launch_copy16(s, h.a, h.b, bytes, v));
        MJ_HIP(ctx, hipEventRecord(h.e1, s));
        MJ_HIP(ctx, hipEventSynchronize(h.e1));
        float t = 0.f;
        MJ_HIP(ctx, hipEventElapsedTime(&t, h.e0, h.e1));
        if (best == 0.f || t < best) best = t;
    }
    float ms = best;
    *ms_per_copy = ms / iters;
    return MJ_OK;
}

// Where a fused plan's coefficient store lies relative to the output buffer changes the fused launch's time by up to 9 %, in two
// classes (config 3: ~5.8 or ~6.3 ms per step), stable for the life of the two allocations and invisible in their virtual
// addresses (tools/placement_probe.py, profiles/r06_placement.txt: the same plan into ten output blocks — three slow, seven fast;
// a second plan's store beside the same blocks — all fast; fresh processes land in either class).  What the round-5 review read as
// "the headline depends on the board" was mostly this.  The store is the plan's own, so the plan can try a few: each candidate
// a fresh hipMalloc, a handful of timed executes (HIP events), the fastest stays.  Worth it for a plan that is executed many
// times into one output buffer (a decode service's slot, a benchmark's step); a one-shot decode should not bother.
int mj_plan_tune_placement(mj_plan *p, void *stream, uint8_t *rgb_device, int32_t candidates, float *ms_out, int32_t *chosen, float *best_out) {
    if (!p || candidates < 1 || candidates > 16) return MJ_ERR_INVALID;
    if (best_out) *best_out = 0.f;
    mj_context *ctx = p->ctx;
    if (chosen) *chosen = 0;
    if (ms_out) for (int i = 0; i < candidates; ++i) ms_out[i] = 0.f;
    if (!(p->use_fused && p->d_blob)) return MJ_OK;          // (only the fused launch has the two streams of traffic that collide)
    MJ_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (int rc = plan_ready(p)) return rc;
    if (p->done_valid) MJ_HIP(ctx, hipEventSynchronize(p->done));
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    MJ_HIP(ctx, hipEventCreate(&ev.a));
    MJ_HIP(ctx, hipEventCreate(&ev.b));
    auto timed = [&](float &ms) -> int {
        int rc = MJ_OK;
        for (int i = 0; i < 3 && rc == MJ_OK; ++i) rc = execute_impl(p, s, rgb_device);
        if (rc != MJ_OK) return rc;
        MJ_HIP(ctx, hipEventRecord(ev.a, s));
        for (int i = 0; i < 5 && rc == MJ_OK; ++i) rc = execute_impl(p, s, rgb_device);
        MJ_HIP(ctx, hipEventRecord(ev.b, s));
        MJ_HIP(ctx, hipEventSynchronize(ev.b));
        MJ_HIP(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
        ms /= 5.f;
        return rc;
    };
    // Two of the plan's buffers carry the launch's traffic: the coefficient store (written, read back) and the stage-0 stream (read
    // by the walk).  Each is tried in turn against everything else as it stands: the plan's own first, then `candidates - 1` others.
    // Candidates come through the context's buffer cache — a block it holds from an earlier plan is as good a candidate as a fresh
    // one, and the winner is then a block the cache knows: when this plan goes, the next plan of its size gets it back (most
    // recently released first), which is how a queue of plans into one output slot keeps what was tuned for that slot.  The losers
    // go back to the device: kept, they would be the next plan's buffers.
    float best_ms = 0.f;
    if (int rc = timed(best_ms)) return rc;
    if (ms_out) ms_out[0] = best_ms;
    auto try_buffers = [&](void **slot, size_t bytes, bool clear, int which) -> int {
        void *best = *slot;
        for (int c = 1; c < candidates; ++c) {
            void *cand = nullptr;
            if (ctx->cache.get(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }     // (no room for another: what we have stands)
            if (clear) MJ_HIP(ctx, hipMemsetAsync(cand, 0, bytes, s));
            *slot = cand;
            float ms = 0.f;
            const int rc = timed(ms);
            if (ms_out && which == 0) ms_out[c] = ms;
            if (rc == MJ_OK && ms < best_ms * 0.99f) {       // (the classes are 5-9 % apart: one per cent is noise)
                ctx->cache.drop(best);
                best = cand; best_ms = ms;
                if (chosen && which == 0) *chosen = c;
            } else {
                ctx->cache.drop(cand);
            }
            *slot = best;
            if (rc != MJ_OK) return rc;
        }
        return MJ_OK;
    };
    if (int rc = try_buffers(reinterpret_cast<void **>(&p->d_coef), (size_t)p->info.total_blocks * 64 * sizeof(int16_t) + 16, false, 0)) return rc;
    if (p->stream_bytes)
        if (int rc = try_buffers(reinterpret_cast<void **>(&p->d_stream), p->stream_bytes, true, 1)) return rc;
    if (best_out) *best_out = best_ms;
    // whatever graph was captured holds the old store's address
    if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
    p->graph_stream = nullptr; p->prev_stream = nullptr; p->last_was_graph = false;
    return mark_done(p, s);
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
