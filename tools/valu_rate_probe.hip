// micro-benchmark: issue cost of a few VALU ops on gfx950 (4 waves per SIMD, 8 independent chains per lane).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe tools/valu_rate_probe.hip; run on the GPU box.  Round 1: every form costs one
// issue slot per instruction (shl_b64 = shl_b32 = add_u64 = fma_f64: 4.9-5.2 cycles counted at 2.4 GHz, i.e. 4 cycles at the
// ~1.96 GHz the chip holds under a VALU-bound kernel) - 64-bit shifts and fp64 FMAs are not slower than 32-bit integer ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__global__ void k(uint64_t *out, int iters, uint32_t s) {
    uint64_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i + s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] << (s & 31) | 1;                                  // 64-bit shift (+or)
            if (OP == 1) { uint32_t lo = (uint32_t)a[i]; lo = lo << (s & 31) | 1; a[i] = (a[i] & 0xFFFFFFFF00000000ull) | lo; }   // 32-bit shift
            if (OP == 2) a[i] = a[i] + 0x123456789ull * s;                              // 64-bit add
            if (OP == 3) { double d = __longlong_as_double(a[i]); d = __builtin_fma(d, 1.0000001, 0.5); a[i] = __double_as_longlong(d); }
            if (OP == 4) a[i] = (a[i] >> (s & 31)) ^ 5;                                  // 64-bit lshr
            if (OP == 5) { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); hi = __builtin_amdgcn_alignbit(hi, lo, 32 - (s & 31)); lo <<= (s & 31); a[i] = ((uint64_t)hi << 32) | lo; }  // manual 64-bit shl via alignbit
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    uint64_t *d; hipMalloc(&d, 1024 * 256 * 8);
    const char *names[] = {"shl_b64+or", "shl_b32+or", "add_u64", "fma_f64", "lshr_b64+xor", "alignbit shl64"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 6; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            const int iters = 20000;
            switch (op) { case 0: k<0><<<1024, 256>>>(d, iters, 3); break; case 1: k<1><<<1024, 256>>>(d, iters, 3); break; case 2: k<2><<<1024, 256>>>(d, iters, 3); break;
                          case 3: k<3><<<1024, 256>>>(d, iters, 3); break; case 4: k<4><<<1024, 256>>>(d, iters, 3); break; case 5: k<5><<<1024, 256>>>(d, iters, 3); break; }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-16s %.2f ms  -> %.2f cycles per wave-op-group at 2.4 GHz (8 ops x 4 waves/SIMD... relative numbers matter)\n", names[op], ms, ms * 1e-3 * 2.4e9 / (20000.0 * 8 * 4));
        }
    }
    return 0;
}
