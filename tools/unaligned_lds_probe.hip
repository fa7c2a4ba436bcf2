// does gfx950 take byte-unaligned ds_read_b128 / ds_write_b128 / ds_read_b32?  (tools probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint8_t *out, uint8_t *out2, int off) {
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(const uint8_t __attribute__((address_space(3))) *)s + threadIdx.x * 17 + off;
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int q = 0; q < 4; ++q) for (int b = 0; b < 4; ++b) out[threadIdx.x * 16 + q * 4 + b] = (uint8_t)(v[q] >> (8 * b));
    __syncthreads();
    // unaligned write: lane l writes 16 bytes of value l at 2048 + l*16 + off  (overlaps resolved by lane order? only check non-overlapping: stride 16)
    u32x4 w = {threadIdx.x * 0x01010101u, threadIdx.x * 0x01010101u, threadIdx.x * 0x01010101u, threadIdx.x * 0x01010101u};
    const uint32_t b2 = (uint32_t)(uintptr_t)(const uint8_t __attribute__((address_space(3))) *)s + 2048 + threadIdx.x * 16 + off;
    asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(b2), "v"(w) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out2[i] = s[2048 + i];
}
int main() {
    uint8_t *d, *d2; hipMalloc(&d, 1024); hipMalloc(&d2, 2048);
    int bad_total = 0;
    for (int off = 0; off < 16; ++off) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d2, off);
        uint8_t h[1024], h2[2048];
        if (hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("off %d: error %s\n", off, hipGetErrorString(hipGetLastError())); return 1; }
        hipMemcpy(h2, d2, 2048, hipMemcpyDeviceToHost);
        int bad = 0, bad2 = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) bad += h[l * 16 + j] != (uint8_t)((l * 17 + off + j) * 7 + 3);
        for (int l = 0; l < 63; ++l) for (int j = 0; j < 16; ++j) bad2 += h2[l * 16 + off + j] != (uint8_t)l;
        printf("off %2d: read mismatches %d, write mismatches %d\n", off, bad, bad2);
        bad_total += bad + bad2;
    }
    printf(bad_total ? "UNALIGNED LDS ACCESS NOT USABLE\n" : "unaligned ds_read_b128 / ds_write_b128 work\n");
    return 0;
}
