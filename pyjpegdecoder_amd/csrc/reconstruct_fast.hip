// Stage 2, fast form — dequantise + 8x8 inverse DCT + chroma upsample + YCbCr->RGB for x-major output.
//
// Same contract and same results as reconstruct.hip (the exact-order kernel); what changes is how a
// block's 64 samples are obtained:
//
//   level 1     separable fp32 IDCT (row pass, LDS transpose, column pass; even/odd split, 36 flops per
//               8-point transform, constants as literals: on gfx950 v_add/v_mul/v_fma_f32 with VGPR or literal
//               operands issue in 2 cycles per wave, fp64 and almost everything else in 4 — tools/issue_rate_probe.hip).
//               Every operation rounds once, so with A = sum |dequantised coefficient| of the block
//               |S_fp32 - S_true| <= 3 * 2^-24 * A  (6 roundings per term and pass, |K| < 1/2 — DESIGN.md section 3);
//               a sample is accepted when S_fp32 is at least  kTieA * A + kTie0  away from the nearest half-integer:
//               then round(S_fp32) == round(S_ref).  About 1.6 % of the blocks of a noisy photograph fail that.
//   level 2     those blocks again, eight at a time, by the separable fp64 IDCT: |S_fp64 - S_ref| < 1e-8 for every
//               int16 input (typically 1e-13); accepted when at least 2^-20 away from a half-integer.
//   level 3     what is left (about one block in 10^4) is recomputed by the exact-order routine (the reference's
//               summation order, bit for bit).
//   DC-only     blocks (frequent in smooth images, and exact ties whenever DC*q = 4 mod 8) need no
//               sum at all: every sample is round(DC*q * T[0,0,0,0]) — one product, same as the reference
//               whose 63 other products are zeros.
//
// Work decomposition: every wavefront works alone on a strip of TMW vertically adjacent MCUs (no block-wide
// barriers; a workgroup is just four such waves sharing an LDS weight table).
//   phase A  8-lane groups, one block each: lane v loads row v of the coefficient block (16 B; stage 1
//            writes blocks as [v][u]), so the first pass needs no cross-lane traffic; the 8x8 transpose
//            between the passes goes through a conflict-free LDS scratch (row stride 72 B, group stride
//            576 B).  Results land in the wave's LDS strip as int16 [x][y] blocks.
//   phase B  lane (x, k) owns the MH consecutive pixels of column x of MCU k: 16-byte LDS reads for Y and
//            for the two chroma source rows it needs; upsample (four cell-corner weights) and colour run in
//            fp32 where fp32 is provably exact, bytes are packed with v_cvt_pk_u8_f32, and the lane writes
//            3*MH contiguous bytes; lanes k+1.. continue the same image column.
//   Chroma blocks sit in the strip WITHOUT the +128 level shift (:872 adds it, :1697-1699 take it off again):
//   phase B works on Cb-128 / Cr-128 directly; the seam outputs and the exact routines put the 128 back
//   (with the reference's int16 wrap).
//   The colour conversion is the reference's float64 expression whenever a pixel sits on (or outside the
//   range where we can rule out) an exact rounding tie — see the comments in phase B and DESIGN.md.
#include "mijpeg_internal.h"
#include "reconstruct_fast_strips.h"

#pragma clang fp contract(off)

namespace mj {

using namespace rfast;

// T: the kernel works on the transposed image (x' = y, y' = x) — its x-major output is the row-major image of the
// original; HS/VS are then the transposed sampling factors, coefficient blocks and tables are stored [u][v].
template <int HS, int VS, int NC, bool SEAMS, bool T>
__global__ __launch_bounds__(256, (HS == 4 || VS == 4) ? 2 : 3) void k_reconstruct_fast(ReconArgs a, const int64_t *__restrict__ job_prefix,
                                                          int64_t total_jobs, int jobs_per_image) {
    using G = FGeo<HS, VS, NC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (G::SUB) {
        fill_weights<HS, VS, NC, T>(reinterpret_cast<float4 *>(smem + 4 * G::WAVE_BYTES), tid, 256);
        __syncthreads();
    }
    // (jobs, tickets and why: reconstruct_fast_strips.h) every wave draws its jobs from the launch's counter word
    TicketSource src;
    src.counter = a.work_counter;
    src.jpt = (uint32_t)a.jobs_per_ticket;
    src.n_tickets = ((uint32_t)total_jobs + src.jpt - 1) / src.jpt;
    src.last_ticket = src.n_tickets + gridDim.x * 4u - 1u;
    src.lane = lane;
    strips_worker<HS, VS, NC, SEAMS, T>(a, job_prefix, total_jobs, jobs_per_image, smem + wave * G::WAVE_BYTES,
                                        reinterpret_cast<const float4 *>(smem + 4 * G::WAVE_BYTES), lane, (int)blockIdx.x, wave, src);
}

template <int HS, int VS, int NC, bool T>
static hipError_t launch_fast_t(hipStream_t stream, const ReconArgs &a, const int64_t *job_prefix, int64_t total_jobs,
                                int jobs_per_image) {
    using G = FGeo<HS, VS, NC>;
    if (total_jobs == 0) return hipSuccess;
    // persistent grid = at most what the chip can hold at once; its waves draw the jobs from a.work_counter
    auto launch = [&](auto kernel) {
        static IntPerDevice resident_of;               // per instantiation and device
        const int resident = resident_of.get([&] {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, G::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
            return per_cu * device_cus();
        });
        const int64_t want = ((total_jobs + a.jobs_per_ticket - 1) / a.jobs_per_ticket + 3) / 4;      // one ticket per wave at least
        const unsigned blocks = (unsigned)(want < resident ? want : resident);
#ifdef MJ_DIAGNOSTIC      // occupancy experiment: MJ_LDS_PAD bytes of unused LDS per workgroup
        if (getenv("MJ_LDS_PAD")) {
            const int pad = atoi(getenv("MJ_LDS_PAD"));
            int per_cu = 0, cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + pad);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, G::LDS_BYTES + pad) != hipSuccess || per_cu < 1) per_cu = 1;
            hipLaunchKernelGGL(kernel, dim3(per_cu * cus), dim3(256), G::LDS_BYTES + pad, stream, a, job_prefix, total_jobs, jobs_per_image);
            return;
        }
#endif
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), G::LDS_BYTES, stream, a, job_prefix, total_jobs, jobs_per_image);
    };
    if (a.planes || a.idct_out) launch(k_reconstruct_fast<HS, VS, NC, true, T>);
    else launch(k_reconstruct_fast<HS, VS, NC, false, T>);
    return hipGetLastError();
}

// hmax/vmax are the ORIGINAL image's sampling factors; a transposed plan runs the kernel with them swapped
int fast_tile_mcus(int hmax, int vmax, int ncomp, bool transposed) {
    const int mw = ncomp == 1 ? 8 : 8 * (transposed ? vmax : hmax), mh = ncomp == 1 ? 8 : 8 * (transposed ? hmax : vmax);
    return 64 / mw * strip_sv(mw, mh, ncomp == 1 ? 1 : 3);
}

hipError_t launch_reconstruct_fast(hipStream_t stream, const ReconArgs &a, int hmax, int vmax, int ncomp, bool transposed,
                                   const int64_t *job_prefix, int64_t total_jobs, int jobs_per_image) {
#define MJ_FAST(H, V, C) \
    (transposed ? launch_fast_t<V, H, C, true>(stream, a, job_prefix, total_jobs, jobs_per_image) \
                : launch_fast_t<H, V, C, false>(stream, a, job_prefix, total_jobs, jobs_per_image))
    if (ncomp == 1) return MJ_FAST(1, 1, 1);
    if (hmax == 1 && vmax == 1) return MJ_FAST(1, 1, 3);
    if (hmax == 2 && vmax == 1) return MJ_FAST(2, 1, 3);
    if (hmax == 1 && vmax == 2) return MJ_FAST(1, 2, 3);
    if (hmax == 2 && vmax == 2) return MJ_FAST(2, 2, 3);
    if (hmax == 4 && vmax == 1) return MJ_FAST(4, 1, 3);
#undef MJ_FAST
    return hipErrorInvalidValue;
}

}  // namespace mj
