// mj_plan_create for progressive (and scan-by-scan baseline) batches: the scans' descriptors, their dependency levels, which
// refining scans are walked as scout + parts, the order of the segments in the band pipeline (plan_progressive_scans), and the
// device side of it — stage-0 pieces, the 11-bit LUTs, the chunked first AC scans' tables (plan_progressive_upload).
// Host side only.  jpeg_decoder.py:908-1366 (progressive_dct_scan), :505-632 (start_of_scan).
#include "plan.h"

namespace mj {

int plan_progressive_scans(mj_context *ctx, const mj_batch *b, mj_plan *p, ProgScans &S, int64_t &ent) {
    std::vector<mj::DevImage> &imgs = p->h_images;
    std::vector<mj::DevProgScan> &pscans = S.pscans;
    std::vector<mj::DevProgSeg> &psegs = S.psegs;
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
    // which refining AC scans are walked as scout + parts: the rule is form_select.h's (choose_prog_split, testable without a GPU)
    std::vector<char> split_of(b->n_scans, 0);
    if (split_mode) {
        const int64_t n_bands = std::max(1, (max_rows + p->prog_rows_per_band - 1) / p->prog_rows_per_band);
        std::vector<int32_t> s_image(b->n_scans), s_segs(b->n_scans);
        std::vector<int64_t> s_bytes(b->n_scans, -1);
        for (int k = 0; k < b->n_scans; ++k) {
            const mj_scan_desc &sd = b->scans[k];
            s_image[k] = sd.image; s_segs[k] = sd.n_segments;
            if (sd.ss == 0 || sd.ah == 0 || sd.n_comp != 1) continue;
            if (sd.image < 0 || sd.image >= b->n_images) continue;                                                            // (refused below)
            if (sd.first_segment < 0 || sd.n_segments < 1 || sd.first_segment + sd.n_segments > b->n_segments) continue;   // (refused below)
            int64_t bytes = 0;
            for (int g = 0; g < sd.n_segments; ++g) bytes += b->seg_end[sd.first_segment + g] - b->seg_begin[sd.first_segment + g];
            s_bytes[k] = std::max<int64_t>(bytes, 0);
        }
        mj::ProgSplitInputs in;
        in.mode = split_mode; in.n_images = b->n_images; in.n_bands = n_bands; in.wave_slots = (int64_t)mj::device_cus() * 32;
        in.parts = p->prog_parts; in.parts_given = mj::opt("MJ_PROG_PARTS") != nullptr;
        in.n_scans = b->n_scans; in.image = s_image.data(); in.n_segments = s_segs.data(); in.bytes = s_bytes.data();
        const mj::ProgSplitChoice ch = mj::choose_prog_split(in);
        split_of = ch.split;
        p->prog_parts = ch.parts;
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
    return MJ_OK;
}

int plan_progressive_upload(mj_context *ctx, const mj_batch *b, mj_plan *p, ProgScans &S) {
    std::vector<mj::DevProgScan> &pscans = S.pscans;
    std::vector<mj::DevProgSeg> &psegs = S.psegs;
    int rc;
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
    return MJ_OK;
}

}  // namespace mj

extern "C" int mj_debug_prog_split(int32_t mode, int32_t n_images, int32_t n_bands, int32_t wave_slots, int32_t parts, int32_t n_scans,
                                   const int32_t *image, const int32_t *n_segments, const int64_t *bytes, uint8_t *split_out, int32_t *parts_out) {
    if (n_scans < 0 || (n_scans > 0 && (!image || !n_segments || !bytes || !split_out)) || !parts_out) return MJ_ERR_INVALID;
    mj::ProgSplitInputs in;
    in.mode = mode; in.n_images = n_images; in.n_bands = n_bands; in.wave_slots = wave_slots > 0 ? wave_slots : 256 * 32;
    in.parts = parts > 0 ? parts : 4; in.parts_given = parts > 0;
    in.n_scans = n_scans; in.image = image; in.n_segments = n_segments; in.bytes = bytes;
    const mj::ProgSplitChoice c = mj::choose_prog_split(in);
    for (int k = 0; k < n_scans; ++k) split_out[k] = (uint8_t)c.split[(size_t)k];
    *parts_out = c.parts;
    return MJ_OK;
}
