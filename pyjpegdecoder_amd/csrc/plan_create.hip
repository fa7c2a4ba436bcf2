// mj_plan_create: validate a batch description, choose the forms its stages take (form_select.h), build the tables and
// descriptors and upload them.  Host-side only.
#include <math.h>

#include "plan.h"

namespace {

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



}  // namespace

extern "C" {

int mj_debug_fused_applies(int32_t layout, int32_t ncomp, int32_t hmax, int32_t vmax, int32_t mcus_per_row, int32_t mcu_rows,
                           int32_t restart_interval, int32_t n_images, uint32_t traits, uint32_t flags, int32_t *mode_out) {
    if (!mode_out || mcus_per_row < 1 || mcu_rows < 1 || n_images < 1) return MJ_ERR_INVALID;
    mj::FusedInputs f;
    f.lanes_resolved = !(traits & 1u); f.seg_order_mode = (traits & 2u) ? 2 : ((traits & 4u) ? 1 : 0);
    f.uniform = !(traits & 8u); f.generic = (traits & 16u) != 0; f.progressive = (traits & 32u) != 0; f.same_interval = !(traits & 64u);
    f.layout = layout; f.transposed = (layout & 1) != 0; f.ncomp = ncomp; f.hmax = hmax; f.vmax = vmax;
    f.flags = flags; f.seam_or_exact_flags = MJ_FLAG_EXACT_ONLY | MJ_FLAG_KEEP_PLANES | MJ_FLAG_KEEP_IDCT;
    f.restart_interval = restart_interval; f.mcu_count_h = mcus_per_row; f.mcu_count_v = mcu_rows;
    f.jobs_per_image = mcus_per_row;        // (x-major plans number their jobs column by column, whole or in equal pieces)
    f.n_images = n_images; f.n_segs = (int64_t)n_images * mj::fused_segments_per_image(f);
    *mode_out = mj::fused_applies(f);
    return MJ_OK;
}

int mj_debug_stage1_form(const int32_t *seg_len, int64_t n_segs, uint64_t blob_len, int32_t n_huff, uint32_t traits, const char *force,
                         int32_t forced_chunk, int32_t out[4]) {
    if (!out || (n_segs > 0 && !seg_len) || n_segs < 0) return MJ_ERR_INVALID;
    mj::FormInputs in;
    in.seg_len = seg_len; in.n_segs = n_segs; in.blob_len = blob_len; in.n_huff = n_huff;
    in.both_roles = traits & 1; in.ordered = !(traits & 2); in.progressive = traits & 4; in.generic = traits & 8;
    in.gpu_segment = traits & 16; in.one_seg_each = traits & 32; in.dc_fits = !(traits & 64); in.no_sync = traits & 128;
    in.wg_lists_ok = !(traits & 256);
    in.force = (force && force[0]) ? force : nullptr;
    in.forced_chunk = forced_chunk;
    const mj::FormChoice c = mj::choose_stage1_form(in);
    out[0] = in.progressive ? MJ_FORM_SCANS : ((c.want_sync ? MJ_FORM_SYNC : (c.use_lanes ? MJ_FORM_LANES : MJ_FORM_WAVE)) | (c.many_tabs && (c.want_sync || c.use_lanes) ? MJ_FORM_WG_TABLES : 0));
    out[1] = c.sync_chunk_bytes;
    out[2] = (int32_t)std::min<int64_t>(c.est_chunks, 0x7fffffff);
    out[3] = mj::spread_lengths(seg_len, n_segs) ? 1 : 0;
    return MJ_OK;
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
    mj::ProgScans prog_scans;      // progressive batches: plan_progressive.hip
    int rc0;
    if (prog && (rc0 = mj::plan_progressive_scans(ctx, b, p, prog_scans, ent)) != MJ_OK) return rc0;
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
        for (int t = 0; t < b->n_huff; ++t) mj::build_dev_huff(b->huff[t], hh[t]);
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
                    const int ab13[4] = {13, 13, 13, 13};
                    int off13[4] = {0, 0, 0, 0}, total13 = 0;
                    if (!mj::build_resolved_tables(b, role, ac_pk, n_ac, ab13, mj::kLanes13SlotBytes, l13, off13, total13)) goto no_lanes13;
                    if ((rc = upload(ctx, &p->d_lut13, l13.data(), l13.size())) != MJ_OK) return rc;
                    p->n_ac13 = n_ac; p->n_dc13 = n_dc;
                    p->ac_slot_pk = ac_pk; p->dc_slot_pk = dc_pk; p->dc_tab_pk = dct_pk;
                    // (a fused launch keeps smaller copies in LDS beside its reconstruction wavefronts' strips: built when the plan
                    // turns out to be one, below)
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
        // which form stage 1 takes: the rule is form_select.h's (choose_stage1_form), here are its inputs
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
        if (const char *e = mj::opt("MJ_SYNC_ROUNDS")) { const int v = atoi(e); if (v >= 0 && v <= 64) p->sync_rounds = v; }
        if (const char *e = mj::opt("MJ_SYNC_WARM")) p->sync_warm_bits = atoi(e) * 8;
        bool one_seg_each = true;
        for (const auto &jb : jobs) one_seg_each = one_seg_each && jb.n_seg == 1;
        if (!jobs.empty() && one_seg_each)
            for (size_t i = 0; i < jobs.size(); ++i) segs[(size_t)jobs[i].first_seg].len = (int32_t)(jobs[i].end - jobs[i].begin);   // upper bound; the scan writes the real one
        std::vector<int32_t> seg_len(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) seg_len[i] = segs[i].len;
        mj::FormInputs fin;
        fin.seg_len = seg_len.data(); fin.n_segs = (int64_t)segs.size(); fin.blob_len = (uint64_t)b->blob_len; fin.n_huff = b->n_huff;
        fin.both_roles = both_roles; fin.ordered = ordered; fin.progressive = prog; fin.generic = p->generic;
        fin.gpu_segment = !jobs.empty(); fin.one_seg_each = one_seg_each; fin.dc_fits = dc_fits;
        fin.no_sync = (b->flags & MJ_FLAG_NO_SYNC) != 0; fin.wg_lists_ok = many_ok_dri;
        const char *force = mj::opt("MJ_HUFFMAN");
        fin.force = force;
        if (const char *e = mj::opt("MJ_SYNC_CHUNK")) { const int v = atoi(e); if (v >= 256 && v <= 65536 && v % 4 == 0) fin.forced_chunk = v; }
        const mj::FormChoice fc = mj::choose_stage1_form(fin);
        const bool lanes_ok = fc.lanes_ok;
        (void)lanes_ok;
        p->use_lanes = fc.use_lanes;
        p->sync_chunk_bytes = fc.sync_chunk_bytes;
        bool want_sync = fc.want_sync;
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
            p->stream_bytes = sbytes;
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
                {   // first chunk of every restart segment: k_build_vsegs runs one workgroup per segment
                    std::vector<int32_t> c0(segs.size() + 1, 0);
                    for (size_t i = 0; i < segs.size(); ++i) c0[i + 1] = c0[i] + std::max(1, (segs[i].len + cb - 1) / cb);
                    if ((rc = upload(ctx, &p->d_seg_chunk0, c0.data(), c0.size())) != MJ_OK) return rc;
                }
                // the counting walks on resolved tables (huffman_sync.hip: k_count) where the batch is of the everyday kind: at most
                // 8 tables, one role each, MCUs of at most 8 blocks; MJ_SYNC_COUNT = classic | resolved (tests, measurements)
                {
                    const char *e = mj::opt("MJ_SYNC_COUNT");
                    bool ok = !(e && !strcmp(e, "classic")) && !many_tabs && !both_roles && b->n_huff <= 8;
                    for (const mj::DevImage &im : imgs) ok = ok && im.blocks_per_mcu <= 8 && im.ncomp <= 3;
                    int wb = 12;
                    if (const char *w = mj::opt("MJ_SYNC_BITS")) wb = atoi(w);
                    for (; ok && wb >= 10; --wb) {
                        std::vector<uint32_t> lc;
                        int tb = 0;
                        if (!mj::build_count_tables(b, role, wb, lc, tb)) { ok = false; break; }
                        if ((size_t)tb * (size_t)b->n_huff > 150 * 1024) continue;             // a narrower index fits
                        if ((rc = upload(ctx, &p->d_lutc, lc.data(), lc.size())) != MJ_OK) return rc;
                        p->lutc_tab_bytes = tb; p->lutc_bits = wb;
                        MJ_HIP(ctx, ctx->cache.get((void **)&p->d_sync_items, ck.size() * 16 + 16));
                        break;
                    }
                }
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
        std::vector<int32_t> by_length_order;      // restart segments, longest first (seg_order_mode != 0)
        if (p->use_lanes && !p->use_sync && p->d_lut13 && jobs.empty() && segs.size() > 1) {
            // the lane form deals restart segments out by length (huffman_lanes13.hip); MJ_SEG_ORDER = blob | binned | striped (tests, measurements)
            const char *e = mj::opt("MJ_SEG_ORDER");
            // (measured, 1024 x 1080p: files of mixed content 7.5 ms in blob order, 7.9 binned, 6.65 striped; files of one kind
            // 4.01 / 4.13 — so segments of similar length stay in blob order)
            const bool spread = mj::spread_lengths(seg_len.data(), (int64_t)seg_len.size());
            p->seg_order_mode = (e && !strcmp(e, "blob")) ? 0 : ((e && !strcmp(e, "binned")) ? 1 : ((e && !strcmp(e, "striped")) || spread ? 2 : 0));
            if (p->seg_order_mode) {
                std::vector<int32_t> &ord = by_length_order;
                ord.resize(segs.size());
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
            int want_cons = 8, want_cons_x = 6;
            bool allow = true;
            if (const char *e = mj::opt("MJ_FUSED")) allow = atoi(e) != 0;
            if (const char *e = mj::opt("MJ_FUSED_CONSUMERS")) want_cons = want_cons_x = atoi(e);
            int luma13 = -1;                                     // MJ_FUSED_LUMA13: 0 / 1 overrides which form keeps component 0's table at 13 bits
            if (const char *e = mj::opt("MJ_FUSED_LUMA13")) luma13 = atoi(e);
            const mj::DevImage &i0 = imgs[0];
            mj::FusedInputs fi;
            fi.lanes_resolved = p->use_lanes && !p->use_sync && p->d_lut13 && p->n_ac13 <= 4;
            fi.seg_order_mode = p->seg_order_mode; fi.uniform = p->uniform; fi.generic = p->generic; fi.progressive = prog;
            fi.transposed = p->transposed; fi.ncomp = p->ncomp; fi.hmax = p->hmax; fi.vmax = p->vmax; fi.layout = p->layout;
            fi.flags = p->flags; fi.seam_or_exact_flags = MJ_FLAG_EXACT_ONLY | MJ_FLAG_KEEP_PLANES | MJ_FLAG_KEEP_IDCT;
            fi.restart_interval = i0.restart_interval; fi.mcu_count_h = i0.mcu_count_h; fi.mcu_count_v = i0.mcu_count_v;
            fi.jobs_per_image = p->jobs_per_image; fi.n_segs = (int64_t)segs.size(); fi.n_images = b->n_images;
            for (const mj::DevImage &im : imgs) fi.same_interval = fi.same_interval && im.restart_interval == i0.restart_interval;
            const int fused_spi = (int)mj::fused_segments_per_image(fi);
            // Restart segments of very different lengths (dealt out by length, seg_order_mode 2: files of mixed content) in
            // blob order — whole images per workgroup — would let the longest wave set the pace of everything (bench.py's mixed
            // content: 11.3 ms fused that way against 10.6 as two launches): they keep their order, and the fused launch's
            // consumers take their jobs from ONE pool, handed over across workgroups (mode 2).
            const int mode = mj::fused_applies(fi);
            // A fused launch's AC tables: a 12-bit main level (half the LDS of the stage-1 kernel's 13 bits) and second-level tables
            // sized to the batch's codes.  With the segments dealt out by length (mode 2) the table of component 0 keeps 13 bits
            // where four consumers still fit beside it: the long segments of such batches are the ones with large coefficients,
            // whose symbols a 12-bit table finishes least often, and the launch lasts as long as their walk.
            // (0 = built and uploaded, 1 = such tables cannot be built — no fused launch then —, negative = an API error)
            int acb = 12, want_prod = 0;                         // MJ_FUSED_ACBITS / MJ_FUSED_PRODUCERS: the experiments of profiles/r06_fused_balance.txt
            if (const char *e = mj::opt("MJ_FUSED_ACBITS")) acb = atoi(e);
            if (const char *e = mj::opt("MJ_FUSED_PRODUCERS")) want_prod = atoi(e);
            auto fused_tables = [&](bool luma13) -> int {
                int ab[4] = {acb, acb, acb, acb};
                if (luma13) ab[(p->ac_slot_pk >> (8 * imgs[0].tab_index[imgs[0].blk_ac_slot[0]])) & 0xFF] = 13;
                std::vector<uint32_t> lf;
                if (!mj::build_resolved_tables(b, role, p->ac_slot_pk, p->n_ac13, ab, 0, lf, p->lutf_off, p->lutf_total)) return 1;
                for (int sl = 0; sl < 4; ++sl) p->lutf_bits[sl] = ab[sl];
                if (p->d_lut12) { ctx->cache.put(p->d_lut12); p->d_lut12 = nullptr; }
                return upload(ctx, &p->d_lut12, lf.data(), lf.size());
            };
            if (allow && want_cons > 0 && mode == 1) {
                if ((rc = fused_tables(luma13 == 1)) < 0) return rc;
                if (rc == 0) {
                    p->fused = mj::fused_shape(mj::device_cus(), p->lutf_total, p->n_dc13, p->hmax, p->vmax, p->transposed, b->n_images, fused_spi, want_cons, want_prod);
                    p->fused_spi = fused_spi;
                    p->use_fused = p->fused.ok;
                }
            } else if (allow && want_cons_x > 0 && mode == 2 && p->d_by_length) {
                if ((rc = fused_tables(luma13 != 0)) < 0) return rc;
                if (rc == 0) p->fused = mj::fused_shape_x(mj::device_cus(), p->lutf_total, p->n_dc13, p->hmax, p->vmax, p->transposed, (int64_t)segs.size(), want_cons_x);
                if (rc == 1 || !p->fused.ok || p->fused.n_cons < std::min(want_cons_x, 4)) {       // (no room for them beside a 13-bit table: 12 bits all round)
                    if ((rc = fused_tables(false)) < 0) return rc;
                    p->fused = mj::FusedShape{};
                    if (rc == 0) p->fused = mj::fused_shape_x(mj::device_cus(), p->lutf_total, p->n_dc13, p->hmax, p->vmax, p->transposed, (int64_t)segs.size(), want_cons_x);
                }
                // (six consumers — all that fit beside a 13-bit table — not eight: with segments of very different lengths the launch
                // lasts as long as its longest wave's walk, and every consumer beside it slows that walk.  bench.py's mixed content,
                // ms per step: 12-bit tables all round 2 consumers 10.2, 4: 8.9, 6: 9.1, 8: 10.5; component 0's table at 13 bits
                // 4: 8.1-8.4, 5: 7.8-7.9, 6: 7.6-7.8; the two launches 10.4)
                p->fused_spi = fused_spi;
                if (p->fused.ok) {
                    // which progress word a segment's wave reports to: the walk deals rank r of the sorted list to wave r mod waves
                    const int64_t n_waves = (int64_t)p->fused.n_wg * p->fused.n_prod;
                    std::vector<int32_t> holder(segs.size());
                    for (size_t r = 0; r < by_length_order.size(); ++r) holder[(size_t)by_length_order[r]] = (int32_t)((int64_t)r % n_waves);
                    if ((rc = upload(ctx, &p->d_holder, holder.data(), holder.size())) != MJ_OK) return rc;
                    // (the ticket counters, the progress words, and room for every job on the list of jobs given up)
                    const int64_t jobs_cap = (int64_t)b->n_images * std::max<int64_t>(p->jobs_per_image, (int64_t)i0.mcu_count_v * i0.mcu_count_h);
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_xwords, (size_t)(32 + n_waves + jobs_cap) * sizeof(uint32_t)));
                    p->use_fused = true;
                }
            }
        }
        if ((rc = upload(ctx, &p->d_segs, segs.data(), segs.size())) != MJ_OK) return rc;
        if (!jobs.empty()) {
            if (prog) return fail(ctx, MJ_ERR_INVALID, "MJ_FLAG_GPU_SEGMENT is for baseline batches");
            if ((rc = upload(ctx, &p->d_jobs, jobs.data(), jobs.size())) != MJ_OK) return rc;
            p->n_jobs = (int)jobs.size();
        }
        if (prog && (rc = mj::plan_progressive_upload(ctx, b, p, prog_scans)) != MJ_OK) return rc;
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

}  // extern "C"
