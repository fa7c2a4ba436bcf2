// mj_plan_create: validate a batch description, choose the forms its stages take (form_select.h), build the tables and
// descriptors and upload them.  Host-side only.
#include <math.h>

#include "plan.h"

namespace {

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


// Resolved AC tables (huffman_lanes13.hip's entry format) for every table of the batch used as an AC table, table (LDS slot) s with
// ab_of_slot[s] index bits: a main level of 2^AB entries — the FINISHED symbol wherever code + value bits fit the index
// (jpeg_decoder.py:834-866 and bin_twos_complement :1636-1646 evaluated here), else what the arithmetic step needs — and second-level
// tables of 2^(16 - AB) entries for the prefixes of longer codes.  fixed_slot_bytes: the stride of a table in `out` (the stage-1
// kernel's), or 0: back to back, each as small as its codes allow (a fused launch's).  slot_off / total_bytes: where each lies.
// false: does not fit.
bool build_resolved_tables(const mj_batch *b, const std::vector<int> &role, uint64_t ac_pk, int n_ac, const int ab_of_slot[4], int fixed_slot_bytes,
                           std::vector<uint32_t> &out, int slot_off[4], int &total_bytes) {
    if (n_ac > 4) return false;
    int subs_of_slot[4] = {1, 1, 1, 1}, words_of_slot[4] = {0, 0, 0, 0};
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 2) continue;
        const int slot = (int)((ac_pk >> (8 * t)) & 0xFF), AB = ab_of_slot[slot], AS = 1 << AB, SUB = 1 << (16 - AB);
        if (fixed_slot_bytes) {
            subs_of_slot[slot] = (fixed_slot_bytes / 4 - AS) / SUB;
            words_of_slot[slot] = fixed_slot_bytes / 4;
            continue;
        }
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
        subs_of_slot[slot] = n;
        words_of_slot[slot] = ((AS + n * SUB) * 4 + 15) / 16 * 4;
    }
    int at = 0;
    for (int sl = 0; sl < n_ac; ++sl) {
        if ((size_t)words_of_slot[sl] * 4 > 65535u) return false;        // (second-level tables are addressed by a 16-bit byte offset)
        slot_off[sl] = at * 4;
        at += words_of_slot[sl];
    }
    total_bytes = at * 4;
    out.assign((size_t)at, 0xFFFFFFFFu);
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 2) continue;
        const int slot = (int)((ac_pk >> (8 * t)) & 0xFF), AB = ab_of_slot[slot], AS = 1 << AB, SUB = 1 << (16 - AB);
        const int SLOT = words_of_slot[slot], max_sub = subs_of_slot[slot];
        uint32_t *tab = out.data() + slot_off[slot] / 4;
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
                } else if (hv == 0 ? l <= AB : l + size <= AB) {
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

// Every table of a batch of at most 8 as the counting walks of the synchronisation form want it (huffman_sync.hip: k_count): all
// with W index bits, `tab_bytes` apart.  A 32-bit entry: bits consumed — code AND value — (0..5) | run + 1, 128 = end of block
// (8..15) | DC tables: the EXTENDed difference (16..30; jpeg_decoder.py:818-820, bin_twos_complement :1636-1646) — finished
// wherever the code fits the index (AC tables: counting does not look at AC values) or code + value bits do (DC tables).  Bit 31
// = not finished: bit 30 set = a code longer than the index, (0..15) the byte offset of the second-level table (2^(16 - W)
// entries for the bits behind the index) for its prefix; else the open form, which second-level tables hold throughout: code
// length (0..4; 0 = no such code) | run + 1 / 128 (8..15) | size (16..19).  false: does not fit (or a DC size above 15).
bool build_count_tables(const mj_batch *b, const std::vector<int> &role, int W, std::vector<uint32_t> &out, int &tab_bytes) {
    if (b->n_huff > 8) return false;
    const int AS = 1 << W, SUB = 1 << (16 - W);
    auto walk_codes = [&](const mj_huff_spec &spec, auto &&f) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            code <<= 1;
            for (int i = 0; i < spec.bits[l - 1] && k < 256; ++i, ++k, ++code)
                if (code < (1 << l)) f(l, code, (int)spec.vals[k]);
        }
    };
    int max_words = AS + SUB;
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 1 && role[t] != 2) return false;
        std::vector<char> seen((size_t)AS, 0);
        int n = 1;
        walk_codes(b->huff[t], [&](int l, int code, int) {
            if (l > W && !seen[(size_t)(code >> (l - W))]) { seen[(size_t)(code >> (l - W))] = 1; ++n; }
        });
        max_words = std::max(max_words, AS + n * SUB);
    }
    tab_bytes = (max_words * 4 + 15) / 16 * 16;
    if (tab_bytes > 65535) return false;
    const int TW = tab_bytes / 4;
    out.assign((size_t)b->n_huff * TW, 0xFFFFFFFFu);
    for (int t = 0; t < b->n_huff; ++t) {
        uint32_t *tab = out.data() + (size_t)t * TW;
        const bool is_dc = role[t] == 1;
        int n_sub = 1;                                            // table 0 = "no such code"
        for (int i = 0; i < SUB; ++i) tab[AS + i] = 0x80000000u;
        bool ok = true;
        auto put = [&](uint32_t *base, uint32_t first, uint32_t count, uint32_t entry) {     // the shortest code wins (first fit)
            for (uint32_t f = 0; f < count; ++f)
                if (base[first + f] == 0xFFFFFFFFu) base[first + f] = entry;
        };
        walk_codes(b->huff[t], [&](int l, int code, int hv) {
            const int run = is_dc ? 0 : hv >> 4, size = is_dc ? hv : (hv & 15);
            if (size > 15) { ok = false; return; }
            const bool eob = !is_dc && hv == 0;
            const uint32_t adv = eob ? 128u : (uint32_t)(run + 1);
            const uint32_t open_entry = 0x80000000u | ((uint32_t)size << 16) | (adv << 8) | (uint32_t)l;
            if (l > W) {
                uint32_t &m = tab[(uint32_t)code >> (l - W)];
                if (m == 0xFFFFFFFFu) {                           // first long code under this prefix: a new table
                    for (int j = 0; j < SUB; ++j) tab[AS + n_sub * SUB + j] = 0xFFFFFFFFu;
                    m = 0xC0000000u | (uint32_t)((AS + n_sub * SUB) * 4);
                    ++n_sub;
                }
                if ((m & 0xC0000000u) != 0xC0000000u) return;     // a shorter code owns the prefix (over-subscribed table)
                put(tab + (m & 0xFFFFu) / 4, ((uint32_t)code << (16 - l)) & (uint32_t)(SUB - 1), 1u << (16 - l), open_entry);
            } else if (!is_dc) {
                put(tab, (uint32_t)code << (W - l), 1u << (W - l), (adv << 8) | (uint32_t)(l + size));
            } else if (l + size <= W) {
                const int rest = W - l - size;
                for (uint32_t vb = 0; vb < (1u << size); ++vb) {
                    const int val = size == 0 ? 0 : ((vb >> (size - 1)) ? (int)vb : (int)vb - ((1 << size) - 1));
                    put(tab, (((uint32_t)code << size) | vb) << rest, 1u << rest, (((uint32_t)val & 0x7FFFu) << 16) | (adv << 8) | (uint32_t)(l + size));
                }
            } else {
                put(tab, (uint32_t)code << (W - l), 1u << (W - l), open_entry);
            }
        });
        if (!ok) return false;
        for (int i = 0; i < AS; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x80000000u;       // no such code
        for (int i = AS; i < TW; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x80000000u;
    }
    return true;
}

}  // namespace

extern "C" {

int mj_debug_count_tables(const mj_huff_spec *huff, int32_t n_huff, const int32_t *roles, int32_t wbits, uint32_t *out, int64_t cap_words,
                          int32_t *tab_bytes) {
    if (!huff || !roles || !tab_bytes || n_huff < 1 || n_huff > 8 || wbits < 10 || wbits > 13) return MJ_ERR_INVALID;
    mj_batch b{};
    b.n_huff = n_huff; b.huff = huff;
    std::vector<int> role(roles, roles + n_huff);
    std::vector<uint32_t> t;
    int tb = 0;
    if (!build_count_tables(&b, role, wbits, t, tb)) return MJ_ERR_UNSUPPORTED;
    *tab_bytes = tb;
    if (out) {
        if ((int64_t)t.size() > cap_words) return MJ_ERR_INVALID;
        memcpy(out, t.data(), t.size() * 4);
    }
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
        // The first AC scans of very large batches are cut into self-synchronising chunks and walked one chunk per LANE before the
        // band pipeline starts (progressive_chunks.hip): they are a third of the wavefront walks' work, which the chip runs out of
        // instruction issue for — from ~1 800 files on; below that a batch lasts as long as one image's chain through its last
        // refinement, and the pass in front of the pipeline (4.5 ms per 1 024 files) only adds to it.  1080p, libjpeg's script, ms per
        // batch without / with: 1 024 files 66.9 / 71.8, 1 536: 79.6 / 84.0, 2 048: 104.2 / 93.5, 3 072: 154.6 / 141.0, 4 096: 204.9 /
        // 184.1 (profiles/r05_progressive_chunks.txt).  MJ_PROG_CHUNKS: 0 never, 1 from 2 048 images on (the default), 2 always (tests).
        {
            int mode = 1;
            if (const char *e = mj::opt("MJ_PROG_CHUNKS")) mode = atoi(e);
            p->prog_chunks = p->prog_fast && p->prog_banded && !(b->flags & MJ_FLAG_NO_SYNC) && (mode >= 2 || (mode == 1 && b->n_images >= 2048));
            if (const char *e = mj::opt("MJ_PROG_CHUNK")) p->pc_chunk_bytes = atoi(e);
        }
        auto chunked = [&](int k) { return p->prog_chunks && b->scans[k].ss > 0 && b->scans[k].ah == 0 && b->scans[k].n_comp == 1; };
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
            // All of them while their scouts and parts leave a fifth of the chip's wave slots free; past that only each image's
            // LARGEST refining scan — the last luma refinement, the one a batch waits for — with two parts per band, up to ~1 100
            // images; past that none.  (libjpeg's script, 1080p, ms per batch, round 5: all split with four parts / largest only with
            // two / none: 384 files 47.4 / 47.8 / 61.3, 512: 48.2 / 48.0 / 61.9, 640: 53.2 / 53.0 / 63.1, 768: 61.3 / 54.0 / 64.8,
            // 896: 65.8 / 61.6 / 66.0, 1 024: 81.5 / 66.2 / 66.7, 1 536: 117 / 99.7 / 79.6.  As many images as fit: worse than
            // either, 99.7 at 1 024.)  MJ_PROG_SPLIT: 0 none, 1 this rule, 2 all, 3 the largest of each image.
            const int64_t slots = (int64_t)mj::device_cus() * 32;
            int64_t need_all = 0;
            for (auto &c : cand) need_all += (int64_t)b->scans[c.second].n_segments * (1 + p->prog_parts);
            const bool parts_given = mj::opt("MJ_PROG_PARTS") != nullptr;
            const bool all = split_mode == 2 || (split_mode == 1 && need_all <= slots * 4 / 5);
            const bool largest = split_mode == 3 || (split_mode == 1 && !all && (int64_t)b->n_images * 3 <= slots * 2 / 5);
            if (all) {
                for (auto &c : cand) split_of[c.second] = 1;
            } else if (largest) {
                std::vector<int64_t> best(b->n_images, 0);
                std::vector<int> which(b->n_images, -1);
                for (auto &c : cand) { const int im = b->scans[c.second].image; if (-c.first > best[im]) { best[im] = -c.first; which[im] = c.second; } }
                for (int i = 0; i < b->n_images; ++i) if (which[i] >= 0) split_of[which[i]] = 1;
                if (!parts_given) p->prog_parts = 2;
            }
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
                    if (chunked(j)) continue;                 // (complete before the band pipeline starts)
                    bool comp_overlap = false;
                    for (int a1 = 0; a1 < sd.n_comp && a1 < 3; ++a1)
                        for (int a2 = 0; a2 < pj.n_comp && a2 < 3; ++a2) comp_overlap |= sd.comp[a1] == pj.comp[a2];
                    // (a split scan's parts run one launch behind its scout: what follows it waits for them)
                    if (comp_overlap && sd.ss <= pj.se && pj.ss <= sd.se) lvl = std::max(lvl, ordinal_of[j] + 1 + (want_split(j) ? 1 : 0));
                }
                ordinal_of[k] = lvl;
            }
            (void)seen;
            if (chunked(k)) ordinal_of[k] = 0;
            else n_ord = std::max(n_ord, ordinal_of[k] + 1 + (want_split(k) ? 1 : 0));
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
                // (and behind everything the segments of the first AC scans that are walked in chunks: no wavefront walk takes them)
                auto in_chunks = [&](const mj::DevProgSeg &g) { return chunked(g.scan); };
                std::stable_sort(psegs.begin(), psegs.end(), [&](const mj::DevProgSeg &x, const mj::DevProgSeg &y) {
                    if (in_chunks(x) != in_chunks(y)) return in_chunks(y);
                    if (in_chunks(x)) return false;           // (among themselves: as they come — by image, scan, restart segment)
                    if (rest(x) != rest(y)) return rest(y);
                    if (split(x) != split(y)) return split(x);
                    return x.len > y.len;
                });
                p->n_split = 0;
                while (p->n_split < (int64_t)psegs.size() && split(psegs[p->n_split])) ++p->n_split;
                p->n_psegs_wave = 0;
                while (p->n_psegs_wave < (int64_t)psegs.size() && !in_chunks(psegs[p->n_psegs_wave])) ++p->n_psegs_wave;
                p->prog_rest_off = 0;
                while (p->prog_rest_off < p->n_psegs_wave && !rest(psegs[p->prog_rest_off])) ++p->prog_rest_off;
            }
            if (!p->prog_banded) p->n_psegs_wave = (int64_t)psegs.size();
            const int n_bands = (max_rows + p->prog_rows_per_band - 1) / p->prog_rows_per_band;
            p->prog_steps = n_bands + std::max(n_ord, 1) - 1;
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
                    const int ab13[4] = {13, 13, 13, 13};
                    int off13[4] = {0, 0, 0, 0}, total13 = 0;
                    if (!build_resolved_tables(b, role, ac_pk, n_ac, ab13, mj::kLanes13SlotBytes, l13, off13, total13)) goto no_lanes13;
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
                        if (!build_count_tables(b, role, wb, lc, tb)) { ok = false; break; }
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
            auto fused_tables = [&](bool luma13) -> int {
                int ab[4] = {12, 12, 12, 12};
                if (luma13) ab[(p->ac_slot_pk >> (8 * imgs[0].tab_index[imgs[0].blk_ac_slot[0]])) & 0xFF] = 13;
                std::vector<uint32_t> lf;
                if (!build_resolved_tables(b, role, p->ac_slot_pk, p->n_ac13, ab, 0, lf, p->lutf_off, p->lutf_total)) return 1;
                for (int sl = 0; sl < 4; ++sl) p->lutf_bits[sl] = ab[sl];
                if (p->d_lut12) { ctx->cache.put(p->d_lut12); p->d_lut12 = nullptr; }
                return upload(ctx, &p->d_lut12, lf.data(), lf.size());
            };
            if (allow && want_cons > 0 && mode == 1) {
                if ((rc = fused_tables(luma13 == 1)) < 0) return rc;
                if (rc == 0) {
                    p->fused = mj::fused_shape(mj::device_cus(), p->lutf_total, p->n_dc13, p->hmax, p->vmax, p->transposed, b->n_images, fused_spi, want_cons);
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
                if (p->prog_chunks && p->n_psegs_wave < (int64_t)psegs.size()) {
                    // the chunked first AC scans (progressive_chunks.hip): their segments, the chunk list — padded to whole wavefronts
                    // per segment, so that a wavefront's lanes share a table —, and per table the 9-bit LUT + canonical code book
                    const int cb = std::min(std::max(p->pc_chunk_bytes, 128), 65536) & ~3;
                    p->pc_chunk_bytes = cb;
                    std::vector<mj::DevAcSeg> as;
                    std::vector<mj::DevChunk> ck;
                    for (size_t i = (size_t)p->n_psegs_wave; i < psegs.size(); ++i) {
                        const mj::DevProgSeg &g = psegs[i];
                        const mj::DevProgScan &ps = pscans[g.scan];
                        mj::DevAcSeg a{};
                        a.image = ps.image; a.comp = ps.comp[0]; a.ss = ps.ss; a.se = ps.se; a.al = ps.al; a.table = ps.ac_tab[0];
                        a.stream_slot = g.stream_slot; a.stream_dw = (int32_t)((g.begin >> 2) + g.stream_slot);
                        a.first_blk = g.mcu0; a.n_blk = g.n_mcu; a.mcu_count_h = ps.mcu_count_h; a.last = g.last;
                        a.chunk0 = (int32_t)ck.size();
                        a.n_chunks = std::max(1, (g.len + cb - 1) / cb);
                        for (int j = 0; j < a.n_chunks; ++j) ck.push_back(mj::DevChunk{(int32_t)as.size(), j});
                        while (ck.size() % 64) ck.push_back(mj::DevChunk{-1, 0});
                        as.push_back(a);
                    }
                    std::vector<uint8_t> tb((size_t)b->n_huff * (1024 + mj::kProgCanonBytes), 0);
                    for (int t = 0; t < b->n_huff; ++t) {
                        uint16_t *l9 = reinterpret_cast<uint16_t *>(tb.data() + (size_t)t * (1024 + mj::kProgCanonBytes));
                        for (int i = 0; i < 512; ++i) {
                            const uint16_t e = lp[(size_t)t * LS + ((size_t)i << (mj::kProgLutBits - 9))];
                            l9[i] = (e >> 8) <= 9 ? e : (uint16_t)0;
                        }
                        uint16_t *lim = l9 + 512;
                        int16_t *base = reinterpret_cast<int16_t *>(lim + 16);
                        uint8_t *vals = reinterpret_cast<uint8_t *>(base + 16);
                        int code = 0, k = 0;
                        for (int l = 1; l <= 16; ++l) {
                            code <<= 1;
                            base[l - 1] = (int16_t)(k - code);
                            const int n = b->huff[t].bits[l - 1];
                            code += n; k += n;
                            lim[l - 1] = (uint16_t)std::min<int64_t>((int64_t)code << (16 - l), 65535);
                        }
                        for (int i = 0; i < 256; ++i) vals[i] = b->huff[t].vals[i];
                    }
                    p->n_acsegs = (int)as.size(); p->n_pc_chunks = (int64_t)ck.size();
                    if ((rc = upload(ctx, &p->d_acsegs, as.data(), as.size())) != MJ_OK) return rc;
                    if ((rc = upload(ctx, &p->d_pc_chunks, ck.data(), ck.size())) != MJ_OK) return rc;
                    if ((rc = upload(ctx, &p->d_pc_tabs, tb.data(), tb.size())) != MJ_OK) return rc;
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pc_exit, ck.size() * 8 + 16));
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pc_outs, ck.size() * sizeof(mj::DevChunkOut) + 16));
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pc_items, ck.size() * 16 + 16));
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pc_owner, ck.size() * 8 + 128));     // (+ the work list's counter behind it, + a lock word per chunk behind that)
                    MJ_HIP(ctx, ctx->cache.get((void **)&p->d_pc_vsegs, ck.size() * sizeof(mj::DevVSeg) + 16));
                } else {
                    p->prog_chunks = false;
                }
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

}  // extern "C"
