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
// bytes per instruction, a wavefront 1 KiB of consecutive addresses, four loads in flight per lane, a grid that fills the chip
// eight workgroups deep.  (torch's uint8 copy_ moved 2.1 TB/s on this chip — below what the decode kernels themselves move.)
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
hipError_t launch_copy16(hipStream_t stream, const void *src, void *dst, int64_t bytes) {
    if (bytes <= 0) return hipSuccess;
    const int64_t n16 = bytes / 16, want = (n16 + 1023) / 1024;
    const int64_t cap = (int64_t)device_cus() * 8;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, stream, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
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
