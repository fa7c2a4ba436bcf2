// Small device-side helpers of the plan's launches: block permutations for the zig-zag seam, word fills.
#include "mijpeg_internal.h"

namespace mj {
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
// The copy SURVEY 8d's second denominator asks for ("what a plain device-to-device copy achieves here"): every lane moves 16
// bytes per instruction, a wavefront 1 KiB of consecutive addresses.  What such a copy reaches depends on its shape more than one
// would like (tools/copy_probe.hip, one MI355X box, 2 GiB, TB/s read + written: 1 024 workgroups with one load in flight per lane
// 5.50, 2 048 with four 4.48, 16 384 with four non-temporal 5.08, hipMemcpyAsync 4.82; torch's uint8 copy_ 2.1) and on which
// two buffers it runs between (4.45-4.86 for one shape over seven destinations): mj_device_copy_rate runs the shapes below and
// reports the best — a ceiling, not an average.
typedef unsigned int mj_cu32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy16(const mj_cu32x4 *__restrict__ src, mj_cu32x4 *__restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        mj_cu32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
int copy16_variants() { return 4; }
hipError_t launch_copy16(hipStream_t stream, const void *src, void *dst, int64_t bytes, int variant) {
    if (bytes <= 0) return hipSuccess;
    const int64_t n16 = bytes / 16;
    const mj_cu32x4 *s = static_cast<const mj_cu32x4 *>(src);
    mj_cu32x4 *d = static_cast<mj_cu32x4 *>(dst);
    const int cus = device_cus();
    auto grid = [&](int per_cu) { const int64_t want = (n16 + 255) / 256, cap = (int64_t)cus * per_cu; return dim3((unsigned)(want < cap ? want : cap)); };
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_copy16<1, false>), grid(4), dim3(256), 0, stream, s, d, n16); break;      // four workgroups per CU, one load in flight per lane
        case 1: hipLaunchKernelGGL((k_copy16<4, false>), grid(8), dim3(256), 0, stream, s, d, n16); break;      // eight, four in flight
        case 2: hipLaunchKernelGGL((k_copy16<4, true>), grid(32), dim3(256), 0, stream, s, d, n16); break;      // 32, four non-temporal
        default: hipLaunchKernelGGL((k_copy16<4, true>), grid(64), dim3(256), 0, stream, s, d, n16); break;     // 64, four non-temporal
    }
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
