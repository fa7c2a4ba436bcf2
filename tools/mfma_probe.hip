// Does the matrix pipe buy stage 2's IDCT anything?  (round-3 review, experiment (a))
// Per SIMD and per 64 8-point transforms (one per lane, what a wave's pass of phase A does):
//   vector:   the even/odd form of reconstruct_fast.hip — 38 v_mul / v_fma / v_add_f32
//   matrix:   the even and the odd 4x4 products as v_mfma_f32_4x4x1_16B_f32 (4 steps each: 8 MFMAs, the lane's own inputs as
//             one operand, the constants as the other) + the 8 butterfly adds; and, for the dense 8x8 product, v_mfma_f32_16x16x4f32
//             with the constant side half empty (2 MFMAs per 16 transforms)
// measured alone, mixed in one wave, and in separate waves of one SIMD, at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe.bin && ./tools/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 2000;

// 38 dependent-ish fp32 operations on 8 values: the instruction mix of idct8f (constants as literals)
__device__ __forceinline__ void valu_transform(float (&f)[8]) {
    const float p = 0.3535534f * (f[0] + f[4]), q = 0.3535534f * (f[0] - f[4]);
    const float r = __builtin_fmaf(0.1913417f, f[6], 0.4619398f * f[2]);
    const float s = __builtin_fmaf(-0.4619398f, f[6], 0.1913417f * f[2]);
    const float e0 = p + r, e3 = p - r, e1 = q + s, e2 = q - s;
    const float o0 = __builtin_fmaf(0.0975452f, f[7], __builtin_fmaf(0.2777851f, f[5], __builtin_fmaf(0.4157348f, f[3], 0.4903926f * f[1])));
    const float o1 = __builtin_fmaf(-0.2777851f, f[7], __builtin_fmaf(-0.4903926f, f[5], __builtin_fmaf(-0.0975452f, f[3], 0.4157348f * f[1])));
    const float o2 = __builtin_fmaf(0.4157348f, f[7], __builtin_fmaf(0.0975452f, f[5], __builtin_fmaf(-0.4903926f, f[3], 0.2777851f * f[1])));
    const float o3 = __builtin_fmaf(-0.4903926f, f[7], __builtin_fmaf(0.4157348f, f[5], __builtin_fmaf(-0.2777851f, f[3], 0.0975452f * f[1])));
    f[0] = e0 + o0; f[7] = e0 - o0; f[1] = e1 + o1; f[6] = e1 - o1;
    f[2] = e2 + o2; f[5] = e2 - o2; f[3] = e3 + o3; f[4] = e3 - o3;
}
// the same transform with its two 4x4 products on the matrix pipe: D_b[i][j] += A_b[i] * B_b[j], 16 blocks of 4 lanes
__device__ __forceinline__ void mfma4_transform(float (&f)[8], const float (&ce)[4], const float (&co)[4]) {
    f32x4 e = {0, 0, 0, 0}, o = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e = __builtin_amdgcn_mfma_f32_4x4x1f32(ce[k], f[2 * k], e, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_4x4x1f32(co[k], f[2 * k + 1], o, 0, 0, 0);
    }
    f[0] = e.x + o.x; f[7] = e.x - o.x; f[1] = e.y + o.y; f[6] = e.y - o.y;
    f[2] = e.z + o.z; f[5] = e.z - o.z; f[3] = e.w + o.w; f[4] = e.w - o.w;
}
// dense: 16 transforms per pair of v_mfma_f32_16x16x4f32 (K = 8 in two steps), the constant side half empty
__device__ __forceinline__ void mfma16_transform(float (&f)[8], float c0, float c1) {
    f32x4 d = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // 64 transforms = 4 groups of 16
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(c0, f[2 * g], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, f[2 * g + 1], d, 0, 0, 0);
    }
    f[0] += d.x; f[1] += d.y; f[2] += d.z; f[3] += d.w;
}

// mode 0: vector only; 1: 4x4x1 only; 2: one of each per iteration in the same wave; 3: even waves vector, odd waves 4x4x1;
// 4: 16x16x4 only; 5: even waves vector, odd waves 16x16x4
__global__ __launch_bounds__(1024) void k_probe(int mode, float *out, unsigned long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float f[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = (float)(lane + i) * 1e-3f; g[i] = (float)(lane - i) * 1e-3f; }
    const float ce[4] = {0.35f, 0.46f, 0.35f, 0.19f}, co[4] = {0.49f, 0.41f, 0.27f, 0.09f};
    const bool odd = wave & 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if (mode == 0 || ((mode == 3 || mode == 5) && !odd)) { valu_transform(f); valu_transform(g); }
        else if (mode == 1 || (mode == 3 && odd)) { mfma4_transform(f, ce, co); mfma4_transform(g, ce, co); }
        else if (mode == 2) { valu_transform(f); mfma4_transform(g, ce, co); }
        else { mfma16_transform(f, ce[0], co[0]); mfma16_transform(g, ce[1], co[1]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { f[i] = f[i] * 0.999f; g[i] = g[i] * 0.999f; }      // keep the values bounded (8 more vector ops per transform pair)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) atomicMax(cycles + blockIdx.x, t1 - t0);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *d_out;
    unsigned long long *d_cyc;
    (void)hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    (void)hipMalloc(&d_cyc, (size_t)cus * 8);
    static const char *names[6] = {"vector (2 x idct8f)", "matrix 4x4x1 (2 transforms)", "one vector + one 4x4x1, same wave", "even waves vector, odd waves 4x4x1",
                                   "matrix 16x16x4, half-empty constant side", "even waves vector, odd waves 16x16x4"};
    printf("cycles per iteration (two 8-point transforms per lane + 16 scaling multiplies), slowest wave of a CU, mean over CUs\n");
    for (int mode = 0; mode < 6; ++mode) {
        printf("%-44s", names[mode]);
        for (int wps : {1, 2, 4}) {           // waves per SIMD: one workgroup of 4 * wps waves per CU
            (void)hipMemset(d_cyc, 0, (size_t)cus * 8);
            hipLaunchKernelGGL(k_probe, dim3(cus), dim3(256 * wps), 0, 0, mode, d_out, d_cyc);
            hipLaunchKernelGGL(k_probe, dim3(cus), dim3(256 * wps), 0, 0, mode, d_out, d_cyc);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(cus);
            (void)hipMemcpy(h.data(), d_cyc, (size_t)cus * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += (double)v;
            printf("  %d/SIMD: %7.1f (%.1f per wave)", wps, s / cus / kIters, s / cus / kIters / wps);
        }
        printf("\n");
    }
    return 0;
}
