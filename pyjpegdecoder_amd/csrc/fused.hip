// Stages 1 and 2 in ONE launch for the batch the benchmark is about: baseline files with one restart interval per MCU row
// (jpeg_decoder.py:805-891 decodes, dequantises, transforms and stores block by block in one loop; :894-900 the restart).
//
// Why.  The lane form of stage 1 (huffman_lanes13.hip) lasts as long as ONE restart segment's serial walk — a dependent chain
// of LDS round trips that leaves ~86 % of its CU's instruction-issue slots empty — and stage 2 (reconstruct_fast.hip) is the
// opposite: bound by issue slots and by the memory pipeline, with nothing to wait for.  As two launches they add up
// (2.7 + 3.9 ms per 1024 x 1080p); side by side as two kernels they do not fit a CU (the lane kernel wants its whole LDS, stage 2
// its whole register file: DESIGN.md §5b, tools/overlap_probe.py).  Here one workgroup per CU holds both kinds of wavefront:
//
//   PRODUCERS   `n_prod` wavefronts walk the workgroup's restart segments, one per lane — lanes13_walk.h, the code of the
//               stage-1 kernel, unchanged — and store the coefficient blocks as always.  A workgroup takes WHOLE images
//               (`ipw` of them), so every block a consumer needs comes from its own workgroup: no dependency crosses a
//               workgroup, nothing crosses an XCD's L2, no wavefront ever waits for one outside its CU.
//   CONSUMERS   `n_cons` wavefronts run stage 2's strip worker — reconstruct_fast_strips.h, unchanged — on JOBS of this
//               workgroup's images (a job = one MCU column of one image, top to bottom), column by column behind the
//               producers: with one restart segment per MCU row, a producer wave that is through MCU m has finished column
//               m of all its rows.  A producer publishes its progress in LDS when its stores of an MCU are known to be in
//               L2 (lanes13_walk.h: the hook behind the AC loop; vmcnt is zero there anyway); a consumer reads it before it
//               asks for a job's blocks — same CU, same vector cache, workgroup scope: no cache maintenance.
//   PHASE 2     a producer that is done waits for the others (the tables they read sit where its strip will be), then
//               becomes a consumer; the jobs left are handed out from the same LDS ticket counter.
//
// Why the coefficients still go through memory.  Stage 2 needs whole MCUs (chroma comes last), a CU holds 272 segments in
// lock-step, and 272 MCUs of coefficients are 209 KB — more than the CU's 160 KB of LDS, which the Huffman tables, block rows
// and stream windows already fill to 136 KB.  That remaining LDS (24 KB; a consumer wave needs 8.3 KB) is also what bounds
// the number of consumers beside the producers.  The blocks are written to the plan's coefficient store and read back
// microseconds later by the same CU — through its L2 while the walk lasts.
//
// Applies to: uniform batches of the five common sampling layouts' 4:2:0 / 4:2:2 / 4:4:4 three-component files in x-major
// output whose restart interval is one MCU row, decoded with the resolved 13-bit tables in blob order (api.hip: fused_ok).
// Everything else — and mj_plan_execute_stage1 / _stage2, MJ_FLAG_KEEP_* — takes the two launches as before.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "mijpeg_internal.h"
#include "lanes13_walk.h"
#include "reconstruct_fast_strips.h"

#pragma clang fp contract(off)

namespace mj {

namespace {

constexpr int kFusedLds = 160 * 1024;
constexpr int kFusedCtrl = 384;            // control words at the very top of LDS: ticket, producers done, progress per producer wave (128 B),
constexpr int kFusedBitmap = 128;          // ... and behind them 64 words of claimed-job bits (the newest-first order)
constexpr int kFusedThreads = 1024;        // 16 wavefronts = 4 per SIMD, 128 registers each: 8 producers and up to 8 consumers beside them

struct FusedArgs {
    lanes13::Args L;                       // the stage-1 launch (lpw = lanes per producer wave)
    ReconArgs R;                           // the stage-2 launch
    const int64_t *job_prefix;
    int64_t total_jobs;
    int32_t jobs_per_image;                // x-major: MCU columns (a job is a whole column); row-major: MCU rows x pieces
    int32_t n_prod, n_cons;                // producer wavefronts; consumer wavefronts beside them
    int32_t ipw, n_pass, n_virt;           // images per workgroup and pass; passes per workgroup; virtual workgroups in all
    int32_t spi;                           // restart segments per image
    int32_t ri;                            // MCUs per restart segment (the last one of an image may be shorter)
    int32_t n_images;
    // row-major plans (the strip worker runs on the transposed image: its "columns" are the MCU ROWS): a job is a PIECE of one
    // MCU row — `pieces` per row, each `piece_mcus` MCUs long (the last one shorter) —, ready when the waves that hold its
    // segments are past it
    int32_t pieces, piece_mcus, mcus_per_row, mcu_rows;
    int32_t col_pieces, piece_rows;        // x-major plans: pieces of an MCU column (JobGeo)
    int32_t simd_split;                    // experiment (MJ_FUSED_SIMD_SPLIT): producers on SIMDs 0-1, consumers on SIMDs 2-3
    int32_t newest;                        // consumers take the NEWEST complete column instead of the next one (FusedSource::claim)
    // segments dealt out by length (XWG): any workgroup's consumers may need any wave's blocks — one ticket counter and the
    // producers' progress words in global memory, and for every restart segment the progress word of the wave that walks it
    uint32_t *x_counter;                   // [0] tickets, [1] jobs given up, [2] the clean-up launch's tickets
    uint32_t *x_progress;
    const int32_t *x_holder;
    uint32_t *x_left;                      // the jobs given up
    int32_t x_patience;                    // polls (~2.5 us each) before a consumer gives a job up: 2000, MJ_FUSED_PATIENCE
};

typedef uint32_t __attribute__((address_space(3))) *lds_word;
typedef uint32_t __attribute__((address_space(3))) *lds_word_t;

// The waits of a fused launch are for wavefronts that never wait themselves (a producer walks its segments and ends), so
// they always end; what bounds them only guards the GPU against a defect in this file.  The bound is WALL-CLOCK time (the
// 100 MHz counter), not a number of polls: a walk's length grows with the row (8K-16K pixel rows at high quality: tens of
// milliseconds) and a resident wave may be held up by another process's kernels while the poller goes on counting.
constexpr unsigned long long kFusedGuardTicks = 200000000ull;      // 2 s
__device__ __forceinline__ bool guard_expired(unsigned long long t0) { return __builtin_amdgcn_s_memrealtime() - t0 > kFusedGuardTicks; }

// Which MCUs a job needs, in terms of restart segments.  An image's MCUs are numbered in raster order, g = row * mpr + column;
// restart segment s = g / ri holds MCU g as its (g - s * ri)-th, and the wave that walks s reports how many MCUs of it are
// complete (lock-step: of every segment it holds).  Any restart interval will do — one MCU row per segment (the benchmark's
// files) is the case where "column m of every row" is "MCU m of every segment".
struct JobGeo {
    uint32_t ri, mpr, mcv, spi;
    uint32_t pieces, piece_mcus;            // row-major plans: a job = MCUs [piece * piece_mcus, ...) of one MCU row
    uint32_t col_pieces, piece_rows;        // x-major plans: a job = MCU rows [piece * piece_rows, ...) of one MCU column (one piece
                                            // where a column is at most 24 strips: 1080p 4:2:0 has 17, 4:2:2 34 -> two pieces)
};

// Jobs of this workgroup's images from a ticket counter in LDS.  The workgroup walks its images in passes of `ipp` images (one
// pass where its producers' lanes hold them all); tickets go pass by pass and, within a pass, column by column (ticket t = column
// t / images, image t % images), each gated by the progress of the producer waves that hold the MCUs the job needs.  BY_ROWS
// (row-major plans): piece by piece instead (ticket t = piece t / rows, row t % rows of the pass's MCU rows).
template <bool BY_ROWS>
struct FusedSource {
    static constexpr bool kSingleJobs = true;      // a ticket = one job, and consecutive tickets are not consecutive jobs
    uint32_t ctrl;                         // LDS address of the control words
    uint32_t n_tickets;
    uint32_t v0, n_pass, ipp, n_images;    // this workgroup's first virtual workgroup (= its pass 0), its passes, images per pass
    uint32_t jobs_per_image, lpw;
    JobGeo g;
    int32_t *status;
    int lane;
    // Newest first (round 6; x-major, one pass, one MCU row per segment, whole columns).  In ticket order the consumers fall
    // behind the walk — by its end they are at column ~60 of 120 — and read coefficient blocks that left every cache long ago.
    // Taking the newest complete column instead (and leaving the old ones to phase 2, which reads from memory anyway) they read
    // what the walk wrote tens of microseconds before: while the launch moves ~4.4 TB/s the 256 MiB Infinity Cache turns over in
    // ~60 us.  Which columns are taken is a bitmap in LDS (one bit per image and column), claimed with an atomic OR.
    uint32_t newest, wpi, n_prod_w, pref;  // on / bitmap words per image / producer wavefronts / the image this wave looks at first
    __device__ __forceinline__ uint32_t ctl_word(uint32_t addr) const {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t __attribute__((address_space(3))) *)(uintptr_t)addr);
    }
    __device__ __forceinline__ uint32_t claim() const {        // a job nobody has taken whose column is complete; ~0: none left
        const uint32_t ih = images_in(0);
        for (;;) {
            const bool all_done = ctl_word(ctrl + 4u) >= n_prod_w;       // (looked at BEFORE the progress words)
            for (uint32_t jj = 0; jj < ih; ++jj) {
                const uint32_t j = pref + jj < ih ? pref + jj : pref + jj - ih;
                uint32_t r = g.mpr;                                       // columns [0, r) of image j are complete
                if (!all_done) {
                    const uint32_t w0 = (j * g.spi) / lpw, w1 = (j * g.spi + g.spi - 1u) / lpw;
                    for (uint32_t w = w0; w <= w1; ++w) r = min(r, ctl_word(ctrl + 8u + 4u * w));
                }
                while (r > 0u) {
                    const uint32_t w = (r - 1u) >> 5, hi = r - (w << 5);  // bits [0, hi) of word w, hi = 1..32
                    const uint32_t addr = ctrl + (uint32_t)kFusedBitmap + 4u * (j * wpi + w);
                    const uint32_t avail = ~ctl_word(addr) & (hi >= 32u ? 0xFFFFFFFFu : (1u << hi) - 1u);
                    if (avail == 0u) { r = w << 5; continue; }             // all taken down to this word's first column
                    const uint32_t b = 31u - (uint32_t)__builtin_clz(avail);
                    uint32_t old = 0;
                    if (lane == 0) old = __hip_atomic_fetch_or((lds_word_t)(uintptr_t)addr, 1u << b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
                    if (!(old & (1u << b))) return (v0 * ipp + j) * jobs_per_image + (w << 5) + b;
                }
            }
            if (all_done) return ~0u;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __device__ __forceinline__ uint32_t images_in(uint32_t pass) const {
        const uint32_t lo = (v0 + pass) * ipp;
        return lo >= n_images ? 0u : min(ipp, n_images - lo);
    }
    __device__ __forceinline__ uint32_t draw() const {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((lds_word)(uintptr_t)ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;
    }
    __device__ __forceinline__ uint32_t take(uint32_t t) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)t); }
    __device__ __forceinline__ uint32_t first_job(uint32_t ticket) const {
        if constexpr (!BY_ROWS) { if (newest) return claim(); }     // (one call per ticket: the ticket only counts the claims)
        uint32_t pass = 0, ih = images_in(0);
        for (; pass + 1 < n_pass; ++pass) {             // (tickets of the passes in front of this one's)
            const uint32_t tp = ih * jobs_per_image;
            if (ticket < tp) break;
            ticket -= tp;
            ih = images_in(pass + 1);
        }
        const uint32_t img0 = (v0 + pass) * ipp;
        if constexpr (BY_ROWS) {
            const uint32_t rows = ih * g.mcv, pc = ticket / rows, rr = ticket - pc * rows, j = rr / g.mcv, row = rr - j * g.mcv;
            return (img0 + j) * jobs_per_image + row * g.pieces + pc;      // (job numbering of the strip worker: row by row, piece by piece)
        } else {
            const uint32_t m = ticket / ih, j = ticket - m * ih;
            return (img0 + j) * jobs_per_image + m;
        }
    }
    __device__ __forceinline__ uint32_t end_job(uint32_t ticket, uint32_t) const { return newest ? 0u : first_job(ticket) + 1u; }   // (unused: single jobs)
    // the progress word of the wave that holds segment s of the pass's image j
    __device__ __forceinline__ uint32_t word(uint32_t j, uint32_t s) const {
        return *(volatile uint32_t __attribute__((address_space(3))) *)(uintptr_t)(ctrl + 8u + 4u * ((j * g.spi + s) / lpw));
    }
    // every MCU the job needs complete?
    __device__ __forceinline__ bool ready(uint32_t job) const {
        if (newest) return true;           // (a claimed column is a complete one)
        const uint32_t img = job / jobs_per_image, m = job - img * jobs_per_image, v = img / ipp, j = img - v * ipp;
        const uint32_t base = (v - v0) << kFusedPassShift;
        if constexpr (BY_ROWS) {           // MCUs [a, b) of one MCU row: the segment(s) that hold them
            const uint32_t row = m / g.pieces, pc = m - row * g.pieces, a = pc * g.piece_mcus, b = min(a + g.piece_mcus, g.mpr);
            const uint32_t g0 = row * g.mpr + a, g1 = row * g.mpr + b - 1u;
            bool ok = true;
            for (uint32_t s = g0 / g.ri; s <= g1 / g.ri; ++s) {
                const uint32_t need = min(g1, (s + 1u) * g.ri - 1u) - s * g.ri;
                ok = ok && (uint32_t)__builtin_amdgcn_readfirstlane((int)word(j, s)) > base + need;
            }
            return ok;
        }
        // x-major: the job is piece q of MCU column mc — that column's MCUs of the MCU rows [r_lo, r_hi)
        const uint32_t mc = m / g.col_pieces, q = m - mc * g.col_pieces, r_lo = q * g.piece_rows, r_hi = min(r_lo + g.piece_rows, g.mcv);
        if (g.ri == g.mpr) {               // one MCU row per segment: MCU mc of the segments r_lo .. r_hi - 1 of the image
            const uint32_t w0 = (j * g.spi + r_lo) / lpw, w1 = (j * g.spi + r_hi - 1u) / lpw;
            uint32_t least = 0x7FFFFFFFu;
            for (uint32_t w = w0; w <= w1; ++w) {
                const uint32_t p = *(volatile uint32_t __attribute__((address_space(3))) *)(uintptr_t)(ctrl + 8u + 4u * w);
                least = min(least, p);
            }
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)least) > base + mc;
        }
        bool behind = false;               // one MCU row per lane and turn
        for (uint32_t r0 = r_lo; r0 < r_hi; r0 += 64u) {
            const uint32_t row = r0 + (uint32_t)lane;
            if (row < r_hi) {
                const uint32_t gi = row * g.mpr + mc, s = gi / g.ri;
                behind = behind || word(j, s) <= base + (gi - s * g.ri);
            }
        }
        return __builtin_amdgcn_ballot_w64(behind) == 0;
    }
    // (bounded by kFusedGuardTicks: on expiry the image is marked MJ_ST_INTERNAL — its blocks were not there — and the wave goes on)
    __device__ __forceinline__ bool wait_ready(uint32_t job) const {
        if (ready(job)) return true;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (!ready(job)) {
            __builtin_amdgcn_s_sleep(8);
            if (guard_expired(t0)) {
                if (lane == 0) atomicMax(status + job / jobs_per_image, MJ_ST_INTERNAL);
                break;
            }
        }
        return true;
    }
};

// The same when the restart segments are dealt out by length (files of mixed content: the lane launch's striped order): a
// workgroup's producers then walk segments of images all over the batch, so the jobs are ONE pool for the whole launch — a
// global ticket counter, column by column over all images (BY_ROWS: piece by piece over all rows) — and a job is ready when
// the progress words (global memory) of the waves that hold its MCUs say so.  The hand-off across CUs and XCDs is
// MI355X_MICROARCH.md's: write-through (sc1) coefficient stores, drained (s_waitcnt vmcnt(0)) before the wave's sc1 progress
// store; the consumer polls with sc1 loads, then ONE agent-scope acquire (invalidates its CU's vector L1), then plain loads.
// Every workgroup of the launch is resident at once (one per CU, grid = CUs): a producer never waits, so nobody waits for a
// wave that has not started.
template <bool BY_ROWS>
struct FusedSourceX {
    static constexpr bool kSingleJobs = true;
    uint32_t *counter;
    const uint32_t *progress;
    const int32_t *holder;
    uint32_t n_tickets, n_images, jobs_per_image;
    JobGeo g;
    uint32_t *left_count, *left_list;      // jobs given up: the clean-up launch's work
    uint32_t patience;                     // polls before a job is given up
    int32_t *status;
    int lane;
    __device__ __forceinline__ uint32_t draw() const {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    }
    __device__ __forceinline__ uint32_t take(uint32_t t) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)t); }
    __device__ __forceinline__ uint32_t first_job(uint32_t ticket) const {
        if constexpr (BY_ROWS) {
            const uint32_t rows = n_images * g.mcv, pc = ticket / rows, rr = ticket - pc * rows, img = rr / g.mcv, row = rr - img * g.mcv;
            return img * jobs_per_image + row * g.pieces + pc;
        } else {
            const uint32_t m = ticket / n_images, img = ticket - m * n_images;
            return img * jobs_per_image + m;
        }
    }
    __device__ __forceinline__ uint32_t end_job(uint32_t ticket, uint32_t) const { return first_job(ticket) + 1u; }
    __device__ __forceinline__ uint32_t word(uint32_t img, uint32_t s) const {
        return __hip_atomic_load(progress + holder[img * g.spi + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ bool ready(uint32_t job) const {
        const uint32_t img = job / jobs_per_image, m = job - img * jobs_per_image;
        bool behind = false;               // some MCU the job needs is not complete
        if constexpr (BY_ROWS) {
            const uint32_t row = m / g.pieces, pc = m - row * g.pieces, a = pc * g.piece_mcus, b = min(a + g.piece_mcus, g.mpr);
            const uint32_t g0 = row * g.mpr + a, g1 = row * g.mpr + b - 1u;
            for (uint32_t s = g0 / g.ri; s <= g1 / g.ri; ++s) behind = behind || word(img, s) <= min(g1, (s + 1u) * g.ri - 1u) - s * g.ri;
        } else {
            const uint32_t mc = m / g.col_pieces, q = m - mc * g.col_pieces, r_lo = q * g.piece_rows, r_hi = min(r_lo + g.piece_rows, g.mcv);
            for (uint32_t r0 = r_lo; r0 < r_hi; r0 += 64u) {
                const uint32_t row = r0 + (uint32_t)lane;
                if (row < r_hi) {
                    const uint32_t gi = row * g.mpr + mc, s = gi / g.ri;
                    behind = behind || word(img, s) <= gi - s * g.ri;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(behind) != 0) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    }
    // A job whose rows are walked by a workgroup that has NOT STARTED cannot become ready while this wave waits for it: every
    // workgroup of the launch is resident at once only when the chip is the launch's alone, and another plan's kernel on
    // another stream may hold CUs.  So the wait is bounded (~5 ms: beyond any wait between resident workgroups — a consumer
    // that reaches the last columns of the batch's slowest image waits ~2 ms for them), and a job not ready by then is GIVEN UP:
    // put on the list the clean-up launch behind this one works off (k_recon_leftover: by then every block is in memory), and
    // the wave stops consuming — the tickets nobody draws are the clean-up launch's too.  Every wave therefore ends, resident
    // or not, after one bounded wait at most.
    __device__ __forceinline__ bool wait_ready(uint32_t job) const {
        for (uint32_t spins = 0; !ready(job); ++spins) {
            __builtin_amdgcn_s_sleep(16);
            if (spins >= patience) {
                if (lane == 0) left_list[__hip_atomic_fetch_add(left_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = job;
                return false;
            }
        }
        return true;
    }
};

}  // namespace

// HS, VS: the sampling factors the strip worker sees (swapped for T, the transposed problem of row-major plans)
// XWG: restart segments dealt out by length (FusedSourceX)
template <int HS, int VS, bool T, bool XWG>
__global__ __launch_bounds__(kFusedThreads) void k_fused(FusedArgs F) {
    using G = rfast::FGeo<HS, VS, 3>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_prod = F.n_prod, n_cons = F.n_cons;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (F.simd_split) {
        // a workgroup's wavefront w sits on SIMD w % 4: with this the walking wavefronts have SIMDs 0 and 1 to themselves (four
        // each) and the reconstructing ones SIMDs 2 and 3 — the role a wavefront plays follows from its SIMD, its number within
        // the role from its slot there (profiles/r06_fused_balance.txt)
        const int simd = wave & 3, slot = wave >> 2, r = slot * 2 + (simd & 1);
        if (simd < 2) wave = r < n_prod ? r : n_prod + n_cons + (r - n_prod);
        else wave = r < n_cons ? n_prod + r : n_prod + n_cons + (8 - min(n_prod, 8)) + (r - n_cons);
        wave = min(wave, kFusedThreads / 64 - 1);
    }
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(smem + kFusedLds - kFusedCtrl);
    float4 *wts = reinterpret_cast<float4 *>(smem + kFusedLds - kFusedCtrl - G::WTS_BYTES);
    if (tid < kFusedCtrl / 4) ctrl[tid] = 0;
    lanes13::stage<true>(F.L, smem, tid, kFusedThreads, n_prod);
    rfast::fill_weights<HS, VS, 3, T>(wts, tid, kFusedThreads);
    __syncthreads();

    typename std::conditional<XWG, FusedSourceX<T>, FusedSource<T>>::type src;
    const uint32_t ctrl_lds = lanes13::lds_addr(ctrl);
    const uint32_t v0 = (uint32_t)blockIdx.x * (uint32_t)F.n_pass;        // this workgroup's first virtual workgroup
    src.n_images = (uint32_t)F.n_images;
    src.jobs_per_image = (uint32_t)F.jobs_per_image;
    src.g = JobGeo{(uint32_t)F.ri, (uint32_t)F.mcus_per_row, (uint32_t)F.mcu_rows, (uint32_t)F.spi, (uint32_t)F.pieces, (uint32_t)F.piece_mcus,
                   (uint32_t)F.col_pieces, (uint32_t)F.piece_rows};
    if constexpr (XWG) {
        src.counter = F.x_counter; src.progress = F.x_progress; src.holder = F.x_holder;
        src.left_count = F.x_counter + 1; src.left_list = F.x_left; src.patience = (uint32_t)F.x_patience;
        src.n_tickets = src.n_images * src.jobs_per_image;
    } else {
        src.ctrl = ctrl_lds;
        src.v0 = v0; src.n_pass = (uint32_t)F.n_pass; src.ipp = (uint32_t)F.ipw;
        src.lpw = (uint32_t)F.L.lpw;
        src.newest = (uint32_t)F.newest; src.wpi = ((uint32_t)F.mcus_per_row + 31u) >> 5; src.n_prod_w = (uint32_t)n_prod;
        src.pref = (uint32_t)wave % max(1u, src.images_in(0));
        uint32_t mine = 0;
        for (uint32_t p = 0; p < src.n_pass; ++p) mine += src.images_in(p);
        src.n_tickets = mine * src.jobs_per_image;
    }
    src.lane = lane;
    src.status = F.L.status;

#ifdef MJ_DIAGNOSTIC     // per workgroup (100 MHz wall clock): start, first / last producer through, last wave out, tickets drawn by then
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(F.R.dump + (3u << 20)) + (size_t)(blockIdx.x & 1023) * 8;
    if (tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
#endif
    // workgroup 0 leaves the shader clock and the 100 MHz clock at its start and end: what frequency the chip held under THIS
    // launch (the launch runs the board into its power cap; mj_context_launch_clock, bench.py's `clock_mhz`)
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(F.R.dump + kDumpClockWords);
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime(); clk[1] = __builtin_amdgcn_s_memrealtime(); }
    unsigned char *my_lds;
    if (wave < n_prod) {
        if constexpr (XWG) {
            lanes13::walk<2>(F.L, smem, lane, wave, n_prod, (int)blockIdx.x, (int)gridDim.x, lanes13::lds_addr(ctrl + 2 + wave));
        } else {
            // pass by pass: the images of virtual workgroup v0 + p.  No barrier between passes — a wave's rows, windows and
            // progress word are its own, the tables are read-only —, and the word keeps rising: pass << 20 | MCUs complete
            for (int p = 0; p < F.n_pass && (int)v0 + p < F.n_virt; ++p)
                lanes13::walk<1>(F.L, smem, lane, wave, n_prod, (int)v0 + p, F.n_virt, lanes13::lds_addr(ctrl + 2 + wave), (uint32_t)p << kFusedPassShift);
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef MJ_DIAGNOSTIC
        if (lane == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            atomicMin(dbg + 1, t);
            atomicMax(dbg + 2, t);
            if constexpr (!XWG) atomicMax(dbg + 4, (unsigned long long)*(volatile uint32_t __attribute__((address_space(3))) *)(uintptr_t)ctrl_lds);
        }
#endif
        if (lane == 0) __hip_atomic_fetch_add((lds_word)(uintptr_t)lanes13::lds_addr(ctrl + 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (wave >= n_prod && wave < n_prod + n_cons) {
        // beside the producers from the start: a strip of its own under the weights, above everything the producers use
        my_lds = smem + kFusedLds - kFusedCtrl - G::WTS_BYTES - (wave - n_prod + 1) * G::WAVE_BYTES;
    } else {
        // phase 2: once every producer is through (their tables and rows are where these strips go).  Idle wavefronts and
        // producers that are through early wait here for the whole walk, however long the rows are; should the guard ever
        // expire the wave simply ends — the jobs it would have taken are drawn by the others, nothing is lost and no status is set
        {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while ((int)*(volatile uint32_t __attribute__((address_space(3))) *)(uintptr_t)(ctrl_lds + 4u) < n_prod) {
                __builtin_amdgcn_s_sleep(16);
                if (guard_expired(t0)) return;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // (its strip: from the bottom of LDS, below the strips of the consumers that were there from the start — as many
        // wavefronts as that leaves room for; 4:1:1 in row-major output has 13 KB strips: twelve fit, not sixteen)
        const int k = wave < n_prod ? wave : wave - n_cons;
        if (k >= (int)((kFusedLds - kFusedCtrl - G::WTS_BYTES) / G::WAVE_BYTES) - n_cons) return;
        my_lds = smem + k * G::WAVE_BYTES;
    }
    rfast::strips_worker<HS, VS, 3, false, T>(F.R, F.job_prefix, F.total_jobs, F.jobs_per_image, my_lds, wts, lane, (int)blockIdx.x, wave, src);
    if (blockIdx.x == 0 && lane == 0) {      // (every wave of workgroup 0 as it ends: the latest stamp stands)
        atomicMax(clk + 2, (unsigned long long)__builtin_amdgcn_s_memtime());
        atomicMax(clk + 3, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
#ifdef MJ_DIAGNOSTIC
    if (lane == 0) atomicMax(dbg + 3, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

#ifdef MJ_DIAGNOSTIC
void dbg_fused_report(const uint8_t *dump, int n_wg) {
    std::vector<unsigned long long> h((size_t)1024 * 8);
    (void)hipMemcpy(h.data(), dump + (3u << 20), h.size() * 8, hipMemcpyDeviceToHost);
    double first = 0, last = 0, end = 0, drawn = 0, end_max = 0;
    const int n = std::min(n_wg, 1024);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < n; ++i) t0 = std::min(t0, h[(size_t)i * 8]);
    for (int i = 0; i < n; ++i) {
        const unsigned long long *r = h.data() + (size_t)i * 8;
        first += (double)(r[1] - r[0]) * 0.01; last += (double)(r[2] - r[0]) * 0.01; end += (double)(r[3] - r[0]) * 0.01; drawn += (double)r[4];
        end_max = std::max(end_max, (double)(r[3] - t0) * 0.01);
    }
    fprintf(stderr, "[diag fused] %d workgroups; per workgroup (us from its start): first producer through %.0f, last %.0f, last wave out %.0f; tickets drawn when the "
                    "last producer was through %.0f; launch (first start to last end) %.0f us\n", n, first / n, last / n, end / n, drawn / n, end_max);
}
void dbg_fused_clear(uint8_t *dump) {
    std::vector<unsigned long long> z((size_t)1024 * 8, 0);
    for (size_t i = 0; i < 1024; ++i) z[i * 8 + 1] = ~0ull;
    (void)hipMemcpy(dump + (3u << 20), z.data(), z.size() * 8, hipMemcpyHostToDevice);
}
#endif


// What a fused launch's consumers left undone (FusedSourceX::wait_ready): the jobs they gave up — a list — and the tickets
// nobody drew after that.  The stage-2 kernel's geometry (four waves per workgroup, a strip each); nothing to do — the usual case —
// costs a launch of workgroups that leave at once.
template <bool BY_ROWS>
struct LeftoverSource {
    static constexpr bool kSingleJobs = true;
    FusedSourceX<BY_ROWS> map;             // the fused launch's ticket -> job numbering
    uint32_t *counter;
    const uint32_t *list;
    uint32_t n_left, first_undrawn, n_tickets;
    int lane;
    __device__ __forceinline__ uint32_t draw() const {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    }
    __device__ __forceinline__ uint32_t take(uint32_t t) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)t); }
    __device__ __forceinline__ uint32_t first_job(uint32_t ticket) const {
        return ticket < n_left ? list[ticket] : map.first_job(first_undrawn + (ticket - n_left));
    }
    __device__ __forceinline__ uint32_t end_job(uint32_t ticket, uint32_t) const { return first_job(ticket) + 1u; }
    __device__ __forceinline__ bool ready(uint32_t) const { return true; }
    __device__ __forceinline__ bool wait_ready(uint32_t) const { return true; }
};

template <int HS, int VS, bool T>
__global__ __launch_bounds__(256, 3) void k_recon_leftover(FusedArgs F) {
    using G = rfast::FGeo<HS, VS, 3>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n_left = __hip_atomic_load(F.x_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t all = (uint32_t)F.n_images * (uint32_t)F.jobs_per_image;
    const uint32_t drawn = min(__hip_atomic_load(F.x_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), all);
    if (n_left == 0 && drawn == all) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (G::SUB) {
        rfast::fill_weights<HS, VS, 3, T>(reinterpret_cast<float4 *>(smem + 4 * G::WAVE_BYTES), tid, 256);
        __syncthreads();
    }
    LeftoverSource<T> src;
    src.map.n_images = (uint32_t)F.n_images; src.map.jobs_per_image = (uint32_t)F.jobs_per_image;
    src.map.g = JobGeo{(uint32_t)F.ri, (uint32_t)F.mcus_per_row, (uint32_t)F.mcu_rows, (uint32_t)F.spi, (uint32_t)F.pieces, (uint32_t)F.piece_mcus,
                       (uint32_t)F.col_pieces, (uint32_t)F.piece_rows};
    src.counter = F.x_counter + 2; src.list = F.x_left; src.n_left = n_left; src.first_undrawn = drawn;
    src.n_tickets = n_left + (all - drawn);
    src.lane = lane;
    rfast::strips_worker<HS, VS, 3, false, T>(F.R, nullptr, F.total_jobs, F.jobs_per_image, smem + wave * G::WAVE_BYTES,
                                              reinterpret_cast<const float4 *>(smem + 4 * G::WAVE_BYTES), lane, (int)blockIdx.x, wave, src);
}

// LDS budget of a fused launch with `n_prod` producer wavefronts of `lpw` lanes: how many consumers fit beside them
static bool fused_budget(FusedShape &s, int n_dc, int hmax, int vmax, int want_consumers) {
    size_t wave_bytes, wts_bytes;
    if (hmax == 2 && vmax == 2) { wave_bytes = rfast::FGeo<2, 2, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<2, 2, 3>::WTS_BYTES; }
    else if (hmax == 2 && vmax == 1) { wave_bytes = rfast::FGeo<2, 1, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<2, 1, 3>::WTS_BYTES; }
    else if (hmax == 1 && vmax == 2) { wave_bytes = rfast::FGeo<1, 2, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<1, 2, 3>::WTS_BYTES; }
    else if (hmax == 1 && vmax == 1) { wave_bytes = rfast::FGeo<1, 1, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<1, 1, 3>::WTS_BYTES; }
    else if (hmax == 4 && vmax == 1) { wave_bytes = rfast::FGeo<4, 1, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<4, 1, 3>::WTS_BYTES; }
    else if (hmax == 1 && vmax == 4) { wave_bytes = rfast::FGeo<1, 4, 3>::WAVE_BYTES; wts_bytes = rfast::FGeo<1, 4, 3>::WTS_BYTES; }     // (4:1:1 transposed)
    else return false;
    const size_t prod = lanes13::lds_bytes(s.ac_total_bytes, n_dc, s.n_prod, s.lpw, s.ring, s.dbits);
    const size_t top = (size_t)kFusedLds - kFusedCtrl - wts_bytes;
    if (prod > top) return false;
    const int waves = kFusedThreads / 64;
    int fit = (int)((top - prod) / wave_bytes);
    fit = std::min(fit, waves - s.n_prod);
    s.n_cons = std::max(0, std::min(fit, want_consumers));
    return s.n_cons >= 1;
}

// How a fused launch would be shaped for `n_images` images of `spi` restart segments each; ok = false: take the two launches.
// A workgroup's producers have 8 x 64 lanes.  Where every workgroup's share of the batch fits them, whole images per workgroup
// and every workgroup resident at once (the benchmark: four 1080p images of 68 rows each); where it does not — 1080p 4:2:2 has
// 135 MCU rows, a restart interval of half a row twice the segments —, a workgroup takes its images in PASSES of as many
// as fit, one after the other (its consumers go on from one pass's jobs to the next's without a break).
FusedShape fused_shape(int cus, int ac_total_bytes, int n_dc, int hmax, int vmax, bool transposed, int n_images, int spi, int want_consumers,
                       int want_producers) {
    FusedShape s{};
    if (transposed) std::swap(hmax, vmax);                     // the strip worker's geometry: that of the transposed image
    if (n_images < 1 || spi < 1 || spi > 8 * 64 || cus < 1) return s;
    const int cap = (8 * 64) / spi;                            // images one pass can hold
    const int64_t v_min = ((int64_t)n_images + cap - 1) / cap;
    s.n_pass = (int)((v_min + cus - 1) / cus);
    if (s.n_pass > 64) return s;
    s.ipw = (int)(((int64_t)n_images + (int64_t)s.n_pass * cus - 1) / ((int64_t)s.n_pass * cus));     // (<= cap)
    s.n_virt = (n_images + s.ipw - 1) / s.ipw;
    s.n_wg = (s.n_virt + s.n_pass - 1) / s.n_pass;
    const int lanes = s.ipw * spi;
    s.n_prod = std::min(8, std::max(1, (lanes + 33) / 34));
    if (want_producers > 0) s.n_prod = std::min(8, std::max((lanes + 63) / 64, want_producers));      // (MJ_FUSED_PRODUCERS: fewer, fuller waves)
    s.lpw = (lanes + s.n_prod - 1) / s.n_prod;
    s.ring = 64;
    s.ac_total_bytes = ac_total_bytes;
    s.dbits = 8;                                                // (Annex K's DC codes of 9..11 bits — differences beyond +-255 in chroma, +-1023 in luma — take the canonical search)
    s.ok = fused_budget(s, n_dc, hmax, vmax, want_consumers);
    return s;
}

// ... with the restart segments dealt out by length (MODE 2 of the walk): no whole images per workgroup — one workgroup per CU,
// the segments spread over all their producer wavefronts.
FusedShape fused_shape_x(int cus, int ac_total_bytes, int n_dc, int hmax, int vmax, bool transposed, int64_t n_segs, int want_consumers) {
    FusedShape s{};
    if (transposed) std::swap(hmax, vmax);
    if (n_segs < 1 || cus < 1) return s;
    const int64_t per_wg = (n_segs + cus - 1) / cus;
    if (per_wg > 8 * 64) return s;
    s.n_prod = (int)std::min<int64_t>(8, std::max<int64_t>(1, (per_wg + 33) / 34));
    s.lpw = (int)((n_segs + (int64_t)cus * s.n_prod - 1) / ((int64_t)cus * s.n_prod));
    s.ring = 64;
    s.ac_total_bytes = ac_total_bytes;
    s.dbits = 8;
    s.xwg = true;
    s.n_wg = cus;
    s.ipw = 0; s.n_pass = 1; s.n_virt = cus;
    s.ok = fused_budget(s, n_dc, hmax, vmax, want_consumers);
    return s;
}

}  // namespace mj

extern "C" int mj_debug_fused_shape(int32_t cus, int32_t n_ac, int32_t n_dc, int32_t ac_slot_bytes, int32_t hmax, int32_t vmax, int32_t transposed,
                                    int32_t n_images, int32_t segments_per_image, int32_t want_consumers, int32_t out[8]) {
    if (!out) return MJ_ERR_INVALID;
    const mj::FusedShape s = mj::fused_shape(cus, n_ac * ac_slot_bytes, n_dc, hmax, vmax, transposed != 0, n_images, segments_per_image, want_consumers);
    out[0] = s.ok ? 1 : 0; out[1] = s.ipw; out[2] = s.n_prod; out[3] = s.lpw; out[4] = s.n_cons;
    out[5] = s.ok ? (int32_t)mj::lanes13::lds_bytes(s.ac_total_bytes, n_dc, s.n_prod, s.lpw, s.ring, s.dbits) : 0;
    out[6] = s.n_pass; out[7] = s.n_wg;
    return MJ_OK;
}

namespace mj {

hipError_t launch_fused(hipStream_t stream, const FusedShape &shape, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs,
                        int64_t n_segs, const DevImage *images, const DevHuff *huff, const uint16_t *lut11, const uint32_t *lut13,
                        int n_ac, int n_dc, uint64_t ac_slot_pk, uint64_t dc_slot_pk, uint64_t dc_tab_pk, const int ac_off[4], const int ac_bits[4],
                        int16_t *coef, int32_t *status,
                        const ReconArgs &a, int hmax, int vmax, bool transposed, int spi, int restart_interval, int mcus_per_row, int mcu_rows,
                        const int64_t *job_prefix, int64_t total_jobs, int jobs_per_image, const int32_t *by_length, const int32_t *holder,
                        uint32_t *x_words) {
    if (restart_interval < 1 || spi < 1) return hipErrorInvalidValue;
    if (!shape.ok || a.n_images < 1) return hipErrorInvalidValue;
    if (shape.xwg && (!by_length || !holder || !x_words)) return hipErrorInvalidValue;
    FusedArgs F{};
    F.L = lanes13::Args{dstream, seg_bits, segs, n_segs, images, huff, lut11, lut13, n_ac, n_dc, ac_slot_pk, dc_slot_pk, dc_tab_pk,
                        coef, status, shape.lpw, transposed ? 1 : 0, nullptr, nullptr, 0, shape.ring, shape.ac_total_bytes, shape.dbits,
                        {ac_off[0], ac_off[1], ac_off[2], ac_off[3]}, {ac_bits[0], ac_bits[1], ac_bits[2], ac_bits[3]}, shape.ipw * spi, nullptr};
    F.R = a;
    F.job_prefix = job_prefix; F.total_jobs = total_jobs; F.jobs_per_image = jobs_per_image;
    F.n_prod = shape.n_prod; F.n_cons = shape.n_cons; F.ipw = shape.ipw; F.n_pass = shape.n_pass; F.n_virt = shape.n_virt;
    F.spi = spi; F.ri = restart_interval; F.n_images = a.n_images;
    F.pieces = 1; F.piece_mcus = mcus_per_row; F.mcus_per_row = mcus_per_row; F.mcu_rows = mcu_rows;
    F.col_pieces = 1; F.piece_rows = mcu_rows;
    if (const char *e = opt("MJ_FUSED_SIMD_SPLIT")) F.simd_split = atoi(e) != 0 && shape.n_prod <= 8 && shape.n_cons <= 8;
    // newest-first consumers (FusedSource::claim): where a column is one job and one test — x-major, one pass, one MCU row per segment
    F.newest = !shape.xwg && !transposed && shape.n_pass == 1 && restart_interval == mcus_per_row && F.col_pieces == 1 &&
               shape.ipw * ((mcus_per_row + 31) / 32) <= (kFusedCtrl - kFusedBitmap) / 4;
    if (const char *e = opt("MJ_FUSED_ORDER")) F.newest = F.newest && !strcmp(e, "newest");
    else F.newest = 0;
    if (!transposed) {
        // the strip worker's jobs of an x-major plan: an MCU column in pieces of a.chunk_strips strips (the plan's numbering:
        // column by column, piece by piece) — one piece where the column is at most 24 strips
        const int tmw = fast_tile_mcus(hmax, vmax, 3, false), spc = (mcu_rows + tmw - 1) / tmw;
        F.col_pieces = (spc + a.chunk_strips - 1) / a.chunk_strips;
        F.piece_rows = a.chunk_strips * tmw;
        if (jobs_per_image != mcus_per_row * F.col_pieces) return hipErrorInvalidValue;
    }
    if (transposed) {
        // the strip worker's jobs on the transposed image are pieces of an MCU ROW of the original (its own numbering: row by
        // row, piece by piece, `chunk_strips` strips each): about 20 MCUs per piece, so that the consumers work a piece behind
        // the walk instead of a row behind it (an LDS ticket costs nothing: the stage-2 kernel's global counter wants big jobs)
        const int tmw = fast_tile_mcus(hmax, vmax, 3, true);    // MCUs per strip (the worker's HS is the image's vmax)
        int piece = 20;
        if (const char *e = opt("MJ_FUSED_PIECE")) piece = atoi(e);
        const int strips = std::max(1, piece / tmw);
        const int spc = (mcus_per_row + tmw - 1) / tmw;
        F.R.chunk_strips = strips;
        F.pieces = (spc + strips - 1) / strips;
        F.piece_mcus = strips * tmw;
        F.jobs_per_image = mcu_rows * F.pieces;
        F.total_jobs = (int64_t)a.n_images * F.jobs_per_image;
    }
    unsigned blocks = 0;
    if (shape.xwg) {
        // the ticket counter (a line of its own) and the producers' progress words start from zero
        const int64_t words = 32 + (int64_t)shape.n_wg * shape.n_prod;
        if (hipError_t e = launch_fill_words(stream, x_words, 0u, words); e != hipSuccess) return e;
        F.x_counter = x_words; F.x_progress = x_words + 32; F.x_holder = holder;
        F.x_left = x_words + 32 + (int64_t)shape.n_wg * shape.n_prod;
        F.x_patience = 2000;
        if (const char *e = opt("MJ_FUSED_PATIENCE")) F.x_patience = atoi(e);
        F.L.by_length = by_length; F.L.order_mode = 2; F.L.progress_global = F.x_progress;
        blocks = (unsigned)shape.n_wg;
    } else {
        blocks = (unsigned)shape.n_wg;
    }
    auto go = [&](auto kernel) {
        static OncePerDevice attr_once;
        attr_once.run([&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds);
        });
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kFusedThreads), kFusedLds, stream, F);
    };
#define MJ_GO(H, V, TT) do { if (shape.xwg) go(k_fused<H, V, TT, true>); else go(k_fused<H, V, TT, false>); } while (0)
    if (!transposed) {
        if (hmax == 2 && vmax == 2) MJ_GO(2, 2, false);
        else if (hmax == 2 && vmax == 1) MJ_GO(2, 1, false);
        else if (hmax == 1 && vmax == 2) MJ_GO(1, 2, false);
        else if (hmax == 1 && vmax == 1) MJ_GO(1, 1, false);
        else if (hmax == 4 && vmax == 1) MJ_GO(4, 1, false);
        else return hipErrorInvalidValue;
    } else {
        if (hmax == 2 && vmax == 2) MJ_GO(2, 2, true);
        else if (hmax == 2 && vmax == 1) MJ_GO(1, 2, true);
        else if (hmax == 1 && vmax == 2) MJ_GO(2, 1, true);
        else if (hmax == 1 && vmax == 1) MJ_GO(1, 1, true);
        else if (hmax == 4 && vmax == 1) MJ_GO(1, 4, true);
        else return hipErrorInvalidValue;
    }
#undef MJ_GO
    if (shape.xwg) {      // the clean-up launch: the jobs given up (none, when the chip was this launch's alone)
        auto left = [&](auto kernel, size_t lds) {
            hipLaunchKernelGGL(kernel, dim3(3 * device_cus()), dim3(256), lds, stream, F);
        };
#define MJ_LEFT(H, V, TT) left(k_recon_leftover<H, V, TT>, rfast::FGeo<H, V, 3>::LDS_BYTES)
        if (!transposed) {
            if (hmax == 2 && vmax == 2) MJ_LEFT(2, 2, false);
            else if (hmax == 2 && vmax == 1) MJ_LEFT(2, 1, false);
            else if (hmax == 1 && vmax == 2) MJ_LEFT(1, 2, false);
            else if (hmax == 4 && vmax == 1) MJ_LEFT(4, 1, false);
            else MJ_LEFT(1, 1, false);
        } else {
            if (hmax == 2 && vmax == 2) MJ_LEFT(2, 2, true);
            else if (hmax == 2 && vmax == 1) MJ_LEFT(1, 2, true);
            else if (hmax == 1 && vmax == 2) MJ_LEFT(2, 1, true);
            else if (hmax == 4 && vmax == 1) MJ_LEFT(1, 4, true);
            else MJ_LEFT(1, 1, true);
        }
#undef MJ_LEFT
    }
    return hipGetLastError();
}

}  // namespace mj
