// Which form stage 1 takes for a batch — the rule, apart from the plan that applies it (plan_create.hip), so that it can be
// read in one place and tested without a GPU (mj_debug_stage1_form, tests/test_form_select.py).  Host-side, no HIP.
//
//   wave    one restart segment per wavefront (huffman.hip): small batches, tables in both roles, unusual sampling layouts
//   lanes   one per lane (huffman_lanes13.hip / huffman_lanes.hip): from ~1000 segments on
//   sync    long segments cut into self-synchronising chunks (huffman_sync.hip): files without restart markers, and
//           batches of ordinary restart segments too small to fill the chip one per lane
// The thresholds are measurements on MI355X (DESIGN.md §3 has the numbers behind each).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace mj {

constexpr int kFormMaxLaneTables = 8;      // = kMaxLaneTables (mijpeg_internal.h): tables the lane forms hold in LDS at once

struct FormInputs {
    const int32_t *seg_len = nullptr;      // byte lengths of the batch's restart segments (with the GPU marker scan and one
    int64_t n_segs = 0;                    // segment per image: the image's byte range, an upper bound)
    uint64_t blob_len = 0;
    int n_huff = 0;
    bool both_roles = false;               // some table serves as DC and as AC table
    bool ordered = true;                   // segments in ascending, non-overlapping blob order
    bool progressive = false, generic = false;
    bool gpu_segment = false;              // MJ_FLAG_GPU_SEGMENT: lengths unknown until the marker scan has run ...
    bool one_seg_each = false;             // ... unless every image is one segment
    bool dc_fits = true;                   // no DC size above 15 (the unified table format of the counting rounds)
    bool no_sync = false;                  // MJ_FLAG_NO_SYNC
    bool wg_lists_ok = true;               // more tables than LDS holds: the lane launch's per-workgroup table lists fit
    const char *force = nullptr;           // MJ_HUFFMAN: wave | lanes | lanes11 | sync, or null
    int forced_chunk = 0;                  // MJ_SYNC_CHUNK, 0 = choose
};

struct FormChoice {
    bool many_tabs = false;                // more distinct tables than the lane forms hold: per-workgroup lists
    bool lanes_ok = false;                 // the lane forms can take the batch at all
    bool use_lanes = false;                // one segment per lane (before want_sync makes it true as well)
    bool want_sync = false;                // chunks + synchronisation rounds (with many tables: still subject to the chunked launches' table lists)
    int sync_chunk_bytes = 2048;
    int64_t est_chunks = 0;
    bool long_segs = false, few_segs = false;
};

inline FormChoice choose_stage1_form(const FormInputs &in) {
    FormChoice c;
    const char *force = in.force;
    c.many_tabs = in.n_huff > kFormMaxLaneTables;
    // (a table serving as DC and as AC table at once, or a stream beyond 32-bit offsets, stays with the wave form; stage 0
    // places segment i's stream at dword (begin_i >> 2) + i, which needs the segments in blob order)
    const bool fits32 = in.blob_len + 4 * (uint64_t)in.n_segs + 4096 < (1ull << 32);
    c.lanes_ok = in.ordered && (!c.many_tabs || in.wg_lists_ok) && !in.both_roles && !in.progressive && !in.generic && fits32;
    c.use_lanes = c.lanes_ok && in.n_segs >= 1024;       // measured crossover with the wave form: ~1000 segments
    if (force && !strcmp(force, "wave")) c.use_lanes = false;
    if (force && (!strcmp(force, "lanes") || !strcmp(force, "lanes11")) && c.lanes_ok) c.use_lanes = true;
    int64_t total_len = 0;
    int32_t longest = 0;
    for (int64_t i = 0; i < in.n_segs; ++i) { total_len += in.seg_len[i]; longest = std::max(longest, in.seg_len[i]); }
    if (in.forced_chunk) {
        c.sync_chunk_bytes = in.forced_chunk;
    } else {
        // Chunk size by the amount of stream: small batches want many short chunks (a single 1080p image: 0.71 ms with
        // 256-byte chunks, 0.74 with 512, 1.14 with 1 KiB — too few wavefronts), big ones fewer long ones (the run-up in
        // front of every chunk and the per-chunk records cost; 1024 images: 11.5 ms at 2 KiB, 12.6 ms at 1 KiB; 256
        // images: 3.43 / 3.26 / 3.74 at 512 / 1024 / 1536; 64 images: 1.38 / 1.34 / 1.70 at 256 / 512 / 1024; 16: 0.94 / 1.06
        // at 256 / 512).  The shortest of 256 / 512 / 1024 / 2048 bytes that keeps the batch under ~50 000 chunks (256)
        // or ~330 000 (the others)
        // ... and a single segment not in more than ~12 000 of them (one 24-megapixel image, 10 MB of stream, round 4:
        // 5.4 ms with 512-byte chunks, 3.8 ms with 1 KiB).
        c.sync_chunk_bytes = 2048;
        for (int cb : {256, 512, 1024})
            if (total_len / cb <= (cb == 256 ? 50000 : 330000) && longest / cb <= 12000) { c.sync_chunk_bytes = cb; break; }
    }
    for (int64_t i = 0; i < in.n_segs; ++i) c.est_chunks += std::max(1, (in.seg_len[i] + c.sync_chunk_bytes - 1) / c.sync_chunk_bytes);
    // Long segments (no DRI, or a very large restart interval) leave the chip empty at one lane each: chosen when segments
    // average >= 32 KiB — or one is >= 64 KiB: in a batch that mixes files with and without restart markers, an image
    // without them would otherwise be one lane's (or one wavefront's) serial walk, 220 ms for a 1080p file
    c.long_segs = in.n_segs > 0 && (total_len / in.n_segs >= 32768 || longest >= 65536) && c.est_chunks >= 64;
    // ... and so do small batches of ordinary restart segments: below ~20 000 segments the lane form cannot fill the
    // chip (its time is one segment's serial walk, ~3.3 ms for a 1080p MCU row, however few there are), while chunks
    // can (measured, 1080p with one restart interval per MCU row: 1 image 3.3 -> 1.4 ms, 32 images 3.9 -> 2.1 ms,
    // 128 images 4.9 -> 3.8 ms, break-even at ~300 images = 20 000 segments).  Segments shorter than a few chunks
    // gain nothing from being cut.
    c.few_segs = in.n_segs > 0 && in.n_segs < 20000 && total_len / in.n_segs >= 2048 && c.est_chunks >= 64;
    // (with many tables the synchronisation form needs its own two workgroup shapes to get by with their table lists;
    // its lane launch runs over chunks, so the restart-segment shape that lanes_ok looked at does not matter for it)
    const bool sync_shape_ok = in.ordered && !in.both_roles && !in.progressive && !in.generic && fits32;
    // (with the GPU marker scan the segment lengths are not known at plan time: possible when every image is one segment)
    c.want_sync = !in.no_sync && (c.many_tabs ? sync_shape_ok : c.lanes_ok) && (!in.gpu_segment || in.one_seg_each) && in.dc_fits &&
                  ((force && !strcmp(force, "sync")) || (!force && (c.long_segs || c.few_segs)));
    return c;
}

// Restart segments of very different lengths (longest > 1.25 x mean) are dealt out to the lane launch's waves by length
// (huffman_lanes13.hip, MJ_SEG_ORDER): measured, 1024 x 1080p of mixed content 7.5 ms in blob order, 6.65 striped; files of
// one kind 4.01 / 4.13 — so segments of similar length stay in blob order.
inline bool spread_lengths(const int32_t *seg_len, int64_t n_segs) {
    int64_t sum = 0;
    int32_t top = 0;
    for (int64_t i = 0; i < n_segs; ++i) { sum += seg_len[i]; top = std::max(top, seg_len[i]); }
    return (int64_t)top * 4 * n_segs > 5 * sum;
}

// Progressive batches: which refining AC scans are walked as SCOUT + PARTS (progressive_fast.hip: a scout follows the bit
// positions alone, a few walks per band place the coefficients one launch behind — a split scan is walked one and a half times,
// for a chain less than half as long).  Worth it where a band's walk is long (from 1 KiB of entropy-coded bytes per band on) and
// while the chip has wave slots for the extra walks:
//   all candidates while their scouts and parts leave a fifth of the chip's wave slots free; past that only each image's LARGEST
//   refining scan — the last luma refinement, the one a batch waits for — with two parts per band, up to ~1 100 images on
//   MI355X (3 walks per image within two fifths of the slots); past that none.
// (libjpeg's script, 1080p, ms per batch, round 5: all split with four parts / largest only with two / none: 384 files 47.4 /
// 47.8 / 61.3, 512: 48.2 / 48.0 / 61.9, 640: 53.2 / 53.0 / 63.1, 768: 61.3 / 54.0 / 64.8, 896: 65.8 / 61.6 / 66.0, 1 024: 81.5 /
// 66.2 / 66.7, 1 536: 117 / 99.7 / 79.6.  As many images as fit: worse than either, 99.7 at 1 024.  The thresholds were measured on
// that one family of files; tests/test_form_select.py pins what the rule says for it.)
struct ProgSplitInputs {
    int mode = 1;                          // MJ_PROG_SPLIT: 0 never | 1 this rule | 2 every candidate | 3 each image's largest
    int64_t n_images = 0, n_bands = 1;
    int64_t wave_slots = 256 * 32;         // CUs x 32 wavefronts
    int parts = 4;                         // parts per band of a split scan (MJ_PROG_PARTS)
    bool parts_given = false;              // ... set by the caller: the rule does not change it
    int n_scans = 0;
    const int32_t *image = nullptr;        // per scan: its image,
    const int32_t *n_segments = nullptr;   // its restart segments,
    const int64_t *bytes = nullptr;        // its entropy-coded bytes — or < 0: no refining AC scan of one component (never split)
};
struct ProgSplitChoice {
    std::vector<char> split;               // per scan
    int parts = 4;
};
inline ProgSplitChoice choose_prog_split(const ProgSplitInputs &in) {
    ProgSplitChoice c;
    c.split.assign((size_t)std::max(in.n_scans, 0), 0);
    c.parts = in.parts;
    if (in.mode == 0) return c;
    std::vector<int> cand;
    for (int k = 0; k < in.n_scans; ++k)
        if (in.bytes[k] >= 0 && (in.mode >= 2 || in.bytes[k] / std::max<int64_t>(in.n_bands, 1) >= 1024)) cand.push_back(k);
    int64_t need_all = 0;
    for (int k : cand) need_all += (int64_t)in.n_segments[k] * (1 + in.parts);
    const bool all = in.mode == 2 || (in.mode == 1 && need_all <= in.wave_slots * 4 / 5);
    const bool largest = in.mode == 3 || (in.mode == 1 && !all && in.n_images * 3 <= in.wave_slots * 2 / 5);
    if (all) {
        for (int k : cand) c.split[(size_t)k] = 1;
    } else if (largest) {
        std::vector<int64_t> best((size_t)std::max<int64_t>(in.n_images, 0), 0);
        std::vector<int> which(best.size(), -1);
        for (int k : cand) {
            const int im = in.image[k];
            if (im < 0 || (size_t)im >= best.size()) continue;
            if (in.bytes[k] > best[(size_t)im]) { best[(size_t)im] = in.bytes[k]; which[(size_t)im] = k; }
        }
        for (int k : which) if (k >= 0) c.split[(size_t)k] = 1;
        if (!in.parts_given) c.parts = 2;
    }
    return c;
}

// One launch for both stages (fused.hip) — the conditions apart from the LDS budget (fused_shape): the resolved-table lane
// form on a uniform batch of 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 / 4:1:1 colour files that all have the SAME restart interval, interleaved
// pixels, no seam outputs.  A consumer's job is ready when the producer waves that hold its MCUs are past them (fused.hip works
// that out per MCU, so any interval is correct); what the rule keeps out is where that would be late:
//   x-major (a job = an MCU column of an image, or a piece of one where it is long): the interval must divide the MCU row — one row (the benchmark's files),
//     half a row, a third ... —: then column m is the (m mod interval)-th MCU of every segment that holds a piece of it.  With
//     several rows per segment a column would only be complete when every segment is in its LAST row: no overlap left;
//   row-major (a job = a piece of one MCU row): any interval — a piece waits for its own segment(s) only.  Not 4:1:1: the
//     transposed strip of its 8 x 32 MCUs is 13 KB, two consumers fit beside the walk — 9.8 ms fused against 7.1 as two launches.
struct FusedInputs {
    bool lanes_resolved = false;           // lane form (not sync) with the resolved tables, 12-bit copies built
    int seg_order_mode = 0;
    bool uniform = false, generic = false, progressive = false, transposed = false;
    bool same_interval = true;             // every image of the batch has image 0's restart interval
    int ncomp = 3, hmax = 1, vmax = 1, layout = 0;
    uint32_t flags = 0, seam_or_exact_flags = 0;
    int restart_interval = 0, mcu_count_h = 0, mcu_count_v = 0, jobs_per_image = 0;
    int64_t n_segs = 0, n_images = 0;
};
// restart segments per image (0: no restart interval)
inline int64_t fused_segments_per_image(const FusedInputs &f) {
    return f.restart_interval > 0 ? ((int64_t)f.mcu_count_h * f.mcu_count_v + f.restart_interval - 1) / f.restart_interval : 0;
}
// 0 = the two launches; 1 = fused, whole images per workgroup (segments in blob order); 2 = fused with the segments dealt out
// by length (seg_order_mode 2: files of mixed content) — one pool of jobs, hand-off across workgroups.
inline int fused_applies(const FusedInputs &f) {
    // x-major: a stage-2 job is an MCU column or an equal piece of one (the plan's numbering); row-major (the strip worker runs
    // on the transposed image): pieces of an MCU row, fused.hip cuts them itself.
    // Which intervals (measured, 1024 x 1080p, EXPERIMENTS.md round 6): row-major jobs need one row's segments, so any interval
    // overlaps; an x-major column needs every MCU row — with a fraction of a row per segment it is complete early in every walk,
    // with TWO rows per segment in the second half of them (-4..-11 % against the two launches), with three rows in the last
    // third (+3 %: the walk beside consumers is slower than alone) and with an interval unrelated to the row only when nearly
    // every segment is done (+3 %).
    const bool layout_ok = f.transposed ? f.layout == 1 : (f.layout == 0 && f.mcu_count_h > 0 && f.jobs_per_image % f.mcu_count_h == 0);
    const int64_t spi = fused_segments_per_image(f);
    const bool interval_ok = f.same_interval && spi >= 1 && spi <= 512 && f.n_segs == f.n_images * spi &&
                             (f.transposed || f.mcu_count_h % f.restart_interval == 0 || f.restart_interval == 2 * f.mcu_count_h);
    const bool ok = f.lanes_resolved && (f.seg_order_mode == 0 || f.seg_order_mode == 2) && f.uniform && !f.generic && !f.progressive && f.ncomp == 3 &&
                    (((f.hmax == 1 || f.hmax == 2) && (f.vmax == 1 || f.vmax == 2)) || (f.hmax == 4 && f.vmax == 1 && !f.transposed)) && layout_ok &&
                    !(f.flags & f.seam_or_exact_flags) && interval_ok;
    return !ok ? 0 : (f.seg_order_mode == 2 ? 2 : 1);
}

}  // namespace mj
