// What does a plain device-to-device copy reach on this MI355X?  (The rooflines' second denominator: util_kernels.hip's k_copy16.)
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe.bin && ./tools/copy_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

template <int UNROLL, bool NT>
static double run(const void *s, void *d, long bytes, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, (const u32x4 *)s, (u32x4 *)d, bytes / 16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, (const u32x4 *)s, (u32x4 *)d, bytes / 16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return 2.0 * bytes / (ms / 10 * 1e-3) / 1e12;
}

int main() {
    const long bytes = 2L << 30;
    std::vector<void *> buf(8);
    for (auto &p : buf) { if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; } hipMemset(p, 1, bytes); }
    hipDeviceSynchronize();
    // warm the clocks
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_copy<4, false>), dim3(2048), dim3(256), 0, 0, (const u32x4 *)buf[0], (u32x4 *)buf[1], bytes / 16);
    hipDeviceSynchronize();
    printf("2 GiB copies, TB/s read + written (8 buffers allocated one after the other)\n");
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        printf("grid %5d x 256:  unroll 1 %.2f   unroll 4 %.2f   unroll 8 %.2f   unroll 4 nt %.2f   unroll 8 nt %.2f\n", blocks,
               run<1, false>(buf[0], buf[1], bytes, blocks), run<4, false>(buf[0], buf[1], bytes, blocks), run<8, false>(buf[0], buf[1], bytes, blocks),
               run<4, true>(buf[0], buf[1], bytes, blocks), run<8, true>(buf[0], buf[1], bytes, blocks));
    }
    printf("pairs (grid 2048, unroll 4):");
    for (int j = 1; j < 8; ++j) printf("  0->%d %.2f", j, run<4, false>(buf[0], buf[j], bytes, 2048));
    printf("\n");
    hipMemcpyAsync(buf[1], buf[0], bytes, hipMemcpyDeviceToDevice, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipMemcpyAsync(buf[1], buf[0], bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpyAsync device to device: %.2f TB/s\n", 2.0 * bytes / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
