// Micro-benchmark (round 3): cycles per Huffman symbol step of the lane-parallel stage 1, by formulation, lanes per wave and
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/step_probe tools/step_probe.hip && /tmp/step_probe
// Every variant runs ITERS x 8 symbol steps on a 13-bit LUT in LDS (entries: 5 bits consumed, no movement, value 7) with the
// bit buffer kept busy by an OR of a per-lane constant; each wave stamps s_memtime around its loop.  Printed: shader cycles
// per symbol step per wave (median over waves), i.e. what one wave's serial path costs at that occupancy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define LOOK \
    "v_bfe_u32 %[t0], v3, 20, 12\n\t"             \
    "v_lshl_add_u32 %[t0], %[t0], 2, %[lutb]\n\t" \
    "ds_read_b32 %[e], %[t0]\n\t"                 \
    "s_waitcnt lgkmcnt(0)\n\t"
#define FLAG "v_cmp_gt_i16 vcc, 0, %[e]\n\ts_cbranch_vccnz 9f\n\t"
#define CORE \
    "v_add_u32_sdwa %[pB], %[pB], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
    "v_lshlrev_b64 v[2:3], %[e], v[2:3]\n\t"                                                                    \
    "v_sub_u32_sdwa %[bc], %[bc], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
    "v_or_b32 v2, v2, %[k]\n\t"
#define CMPX1 "v_cmpx_gt_u32 %[storeB], %[pB]\n\t"
#define CMPX2 "v_cmpx_gt_u32 %[lastB], %[pB]\n\t"
#define WR "ds_write_b16_d16_hi %[pB], %[e]\n\t"
// exec narrowed through vcc and the scalar unit instead of v_cmpx
#define SCMP1 "v_cmp_gt_u32 vcc, %[storeB], %[pB]\n\ts_and_b64 exec, exec, vcc\n\t"
#define SCMP2 "v_cmp_gt_u32 vcc, %[lastB], %[pB]\n\ts_and_b64 exec, exec, vcc\n\t"
// no exec at all: clamp the position (v_min) and let finished lanes write their pad slot
#define CLAMP "v_min_u32 %[pB], %[pB], %[storeB]\n\t"

enum V { V_FULL, V_NOFLAG, V_NOCMPX, V_CHAIN, V_SCMP, V_CLAMP, V_ONECMPX, V_LOOKONLY, V_NOLDSWR, V_NEW, V_NEWPIPE, V_IT_OLD, V_IT_NEW, V_IT_NEWPIPE, N_V };
static const char *kNames[] = {"look + flag + core + cmpx + write + cmpx (the kernel's step)", "  without the flag test", "  without the two v_cmpx",
                               "  look + core + write only (dependent chain)", "  exec through v_cmp + s_and_b64 instead of v_cmpx",
                               "  v_min clamp instead of exec (no cmpx)", "  one v_cmpx (after the write)", "  look only (LDS round trip + 2 VALU)",
                               "  full step without the LDS write",
                               "new: look + core + one cmpx + write (flag test deferred)", "new, the write issued behind the next read (2 entry registers)",
                               "ITERATION old: refill(cmp+saveexec) + 2 full steps           [per step]",
                               "ITERATION new: refill(cmpx) + 2 new steps + deferred flag test [per step]",
                               "ITERATION new, writes behind the next read                     [per step]"};

template <int VAR>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters, int lanes) {
    __shared__ uint32_t lut[4096 + 64 * 33];
    for (int i = threadIdx.x; i < 4096; i += 256) lut[i] = (7u << 16) | (0u << 8) | 5u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint64_t bb = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
    uint32_t bc = 1u << 30, e = 0, t0, e2 = 0, voff = 0;
    uint64_t nx = threadIdx.x * 77u, tmp;
    const uint32_t lutb = (uint32_t)(uintptr_t)(uint32_t __attribute__((address_space(3))) *)lut;
    uint32_t pB = lutb + 4096 * 4 + lane * 132 + (threadIdx.x >> 6) * 0;   // rows behind the table (one wave's worth is enough: values are never read)
    const uint32_t storeB = lane < lanes ? 0xFFFFFFF0u : 0u, lastB = storeB;
    const uint32_t kk = 0x01234567u * (lane + 3) | 1u;
    __builtin_amdgcn_s_barrier();
    uint64_t t_0, t_1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_0)::"memory");
#define LOOK2 \
    "v_bfe_u32 %[t0], v3, 20, 12\n\t"             \
    "v_lshl_add_u32 %[t0], %[t0], 2, %[lutb]\n\t" \
    "ds_read_b32 %[e2], %[t0]\n\t"
#define CORE2 \
    "v_add_u32_sdwa %[pB], %[pB], %[e2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
    "v_lshlrev_b64 v[2:3], %[e2], v[2:3]\n\t"                                                                    \
    "v_sub_u32_sdwa %[bc], %[bc], %[e2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
    "v_or_b32 v2, v2, %[k]\n\t"
#define LOOKNW \
    "v_bfe_u32 %[t0], v3, 20, 12\n\t"             \
    "v_lshl_add_u32 %[t0], %[t0], 2, %[lutb]\n\t" \
    "ds_read_b32 %[e], %[t0]\n\t"
#define WR2 "ds_write_b16_d16_hi %[pB], %[e2]\n\t"
// pipelined pair: read A; (write B of the step before); wait for A; core A; cmpx; read B; write A; wait for B; core B; cmpx
#define PIPE2 LOOKNW WR2 "s_waitcnt lgkmcnt(1)\n\t" CORE CMPX2 LOOK2 WR "s_waitcnt lgkmcnt(1)\n\t" CORE2 CMPX2
#define REFILL_OLD \
    "v_cmp_ge_u32 vcc, 32, %[bc]\n\ts_and_saveexec_b64 s[42:43], vcc\n\tv_sub_u32 %[t0], 32, %[bc]\n\tv_lshlrev_b64 v[4:5], %[t0], v[6:7]\n\t" \
    "v_or_b32 v3, v3, v5\n\tv_mov_b32 v2, v4\n\tv_add_u32 %[bc], 32, %[bc]\n\tv_add_u32 %[voff], 4, %[voff]\n\ts_mov_b64 exec, s[42:43]\n\t"
#define REFILL_NEW \
    "s_mov_b64 s[42:43], exec\n\tv_cmpx_ge_u32 32, %[bc]\n\tv_sub_u32 %[t0], 32, %[bc]\n\tv_lshlrev_b64 v[4:5], %[t0], v[6:7]\n\t" \
    "v_or_b32 v3, v3, v5\n\tv_mov_b32 v2, v4\n\tv_add_u32 %[bc], 32, %[bc]\n\tv_add_u32 %[voff], 4, %[voff]\n\ts_mov_b64 exec, s[42:43]\n\t"
#define CHECK "s_mov_b64 s[44:45], exec\n\ts_mov_b64 exec, s[40:41]\n\tv_cmp_gt_i16 vcc, 0, %[e]\n\ts_mov_b64 exec, s[44:45]\n\ts_cbranch_vccnz 9f\n\t"
#define STEP_ASM(body) \
    asm volatile("s_mov_b64 s[40:41], exec\n\tv_cmpx_gt_u32 %[lastB], %[pB]\n\t" body body body body body body body body "9:\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[40:41]\n\t" \
                 : "+{v[2:3]}"(bb), [bc] "+v"(bc), [pB] "+v"(pB), [e] "+v"(e), [t0] "=&v"(t0), [e2] "+v"(e2), "+{v[6:7]}"(nx), "=&{v[4:5]}"(tmp), [voff] "+v"(voff) \
                 : [lastB] "v"(lastB), [storeB] "v"(storeB), [lutb] "v"(lutb), [k] "v"(kk)                                                         \
                 : "memory", "vcc", "s40", "s41", "s42", "s43", "s44", "s45")
    for (int it = 0; it < iters; ++it) {
        if (VAR == V_FULL) STEP_ASM(LOOK FLAG CORE CMPX1 WR CMPX2);
        if (VAR == V_NOFLAG) STEP_ASM(LOOK CORE CMPX1 WR CMPX2);
        if (VAR == V_NOCMPX) STEP_ASM(LOOK FLAG CORE WR);
        if (VAR == V_CHAIN) STEP_ASM(LOOK CORE WR);
        if (VAR == V_SCMP) STEP_ASM(LOOK FLAG CORE SCMP1 WR SCMP2);
        if (VAR == V_CLAMP) STEP_ASM(LOOK FLAG CORE CLAMP WR);
        if (VAR == V_ONECMPX) STEP_ASM(LOOK FLAG CORE WR CMPX2);
        if (VAR == V_LOOKONLY) STEP_ASM(LOOK "v_xor_b32 v3, v3, %[e]\n\t");
        if (VAR == V_NOLDSWR) STEP_ASM(LOOK FLAG CORE CMPX1 CMPX2);
        if (VAR == V_NEW) STEP_ASM(LOOK CORE CMPX2 WR);
        if (VAR == V_NEWPIPE) { asm volatile("s_mov_b64 s[40:41], exec\n\tv_cmpx_gt_u32 %[lastB], %[pB]\n\t" PIPE2 PIPE2 PIPE2 PIPE2 "9:\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[40:41]\n\t"
                 : "+{v[2:3]}"(bb), [bc] "+v"(bc), [pB] "+v"(pB), [e] "+v"(e), [t0] "=&v"(t0), [e2] "+v"(e2), "+{v[6:7]}"(nx), "=&{v[4:5]}"(tmp), [voff] "+v"(voff)
                 : [lastB] "v"(lastB), [storeB] "v"(storeB), [lutb] "v"(lutb), [k] "v"(kk) : "memory", "vcc", "s40", "s41", "s42", "s43", "s44", "s45"); }
#define IT_ASM(body) asm volatile("s_mov_b64 s[40:41], exec\n\tv_cmpx_gt_u32 %[lastB], %[pB]\n\t" body body body body "9:\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[40:41]\n\t" \
                 : "+{v[2:3]}"(bb), [bc] "+v"(bc), [pB] "+v"(pB), [e] "+v"(e), [t0] "=&v"(t0), [e2] "+v"(e2), "+{v[6:7]}"(nx), "=&{v[4:5]}"(tmp), [voff] "+v"(voff) \
                 : [lastB] "v"(lastB), [storeB] "v"(storeB), [lutb] "v"(lutb), [k] "v"(kk) : "memory", "vcc", "s40", "s41", "s42", "s43", "s44", "s45")
        if (VAR == V_IT_OLD) IT_ASM(REFILL_OLD LOOK FLAG CORE CMPX1 WR CMPX2 LOOK FLAG CORE CMPX1 WR CMPX2);
        if (VAR == V_IT_NEW) IT_ASM(REFILL_NEW LOOK CORE CMPX2 WR LOOK CORE CMPX2 WR CHECK);
        if (VAR == V_IT_NEWPIPE) IT_ASM(REFILL_NEW PIPE2 CHECK);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_1)::"memory");
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane == 0) { out[2 * gw] = t_1 - t_0; out[2 * gw + 1] = bb ^ bc ^ pB ^ e; }
}

template <int VAR>
static void run(uint64_t *d, std::vector<uint64_t> &h) {
    const int iters = 2000;
    for (int lanes : {64, 17}) {
        printf("%-62s %2d lanes", kNames[VAR], lanes);
        for (int wps = 1; wps <= 4; ++wps) {
            const int blocks = 256 * wps;
            k<VAR><<<blocks, 256>>>(d, iters, lanes);
            k<VAR><<<blocks, 256>>>(d, iters, lanes);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, blocks * 4 * 16, hipMemcpyDeviceToHost);
            std::vector<uint64_t> cyc(blocks * 4);
            for (int i = 0; i < blocks * 4; ++i) cyc[i] = h[2 * i];
            std::sort(cyc.begin(), cyc.end());
            printf("  w%d: %6.1f", wps, (double)cyc[cyc.size() / 2] / ((double)iters * 8));
        }
        printf("\n");
    }
}

template <int VAR>
static void run_all(uint64_t *d, std::vector<uint64_t> &h) {
    run<VAR>(d, h);
    if constexpr (VAR + 1 < N_V) run_all<VAR + 1>(d, h);
}

int main() {
    uint64_t *d;
    hipMalloc(&d, 2048 * 4 * 16);
    std::vector<uint64_t> h(2048 * 4 * 2);
    printf("# tools/step_probe.hip: shader cycles per symbol step per wave; wN = N waves per SIMD\n");
    run_all<0>(d, h);
    return 0;
}
