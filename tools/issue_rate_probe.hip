// Micro-benchmark (round 2): what one gfx950 SIMD issues per cycle, by instruction kind and by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate_probe tools/issue_rate_probe.hip && /tmp/issue_rate_probe
// Every kernel is ITERS x 64 inline-asm instructions on 16 independent register chains per lane (so neither the
// compiler nor a dependency can reshape the stream); each wave stamps s_memtime around its loop; the figure printed
// is  cycles of the slowest wave's loop / (instructions per wave x waves per SIMD)  = shader cycles one SIMD spends per
// wave-instruction, at 1, 2, 4 and 8 waves per SIMD (grid = 256 CUs x that many 256-thread workgroups).
// Results: DESIGN.md section 3 ("issue model").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Op { ADD_U32, FMA_F32, PK_FMA_F32, FMA_F64, ADD_F64, MUL_F64, CVT_F64_I32, RNDNE_F64, LSHL_B64, MAD_U24, MUL_LO_U32,
          PK_MUL_LO_U16, PK_ADD_I16, CVT_F32_I32, CVT_PK_U8, PERM_B32, BFE_I32, DOT2_I32_I16, MOV_DPP, SALU_ADD,
          MIX_VALU_SALU, MIX_F64_F32, PK_ADD_F32, PK_MUL_F32, MAD_I32_I16, LSHL_OR, AND_OR, MIX_F64_INT, READLANE, MAD_I64_I32,
          AND_B32, OR_B32, XOR_B32, LSHL_B32, LSHR_B32, ASHR_I32, SUB_U32, CNDMASK, CMP_GT, MIN_U32, MAX_I32, MOV_B32, ADD3_U32, OR3_B32,
          BFE_U32, ALIGNBIT, MUL_F32, ADD_F32, FMAC_F32, MUL_U24, ADD_CO, LSHL_ADD, CVT_I32_F32, RNDNE_F32, MED3_I32, MAX_F32, CVT_UB0,
          MAX3_F32, FMA_F32_3SRC, SALU_VALU_2W, LDS_RD128, LDS_WR128, LDS_RD64,
          CND_VCC_SET, CND_SGPR, CMP_SGPR, ADD_SGPR, LSHL_VAR, LSHR_B64, BFI, SUB_F32, MUL_I24, MUL_HI, CVT_SDWA, ADD_SDWA, MOV_SDWA,
          ADD_F32_ABS, MIN_F32, AND_LIT, FMA_SGPR, MUL_LIT, FMAAK, FMAMK, ADD_F32_NEG, SUBREV, XNOR, LSHL_B32_E32VAR, MUL_F32_SGPR,
          CVT_F32_U32, CVT_U32_F32, FRACT, SUB_CO, ADDC, MBCNT,
          CND_E64_VCC, CMP_CND_PAIR, CND_E32_OTHERDST, CND_E32_ALLONES, CMP_E32_ONLY, CMPS_CNDS_PAIR, BFI_SEL, VCC1, VCC_NOP, VCC_EACH, VCC_EACH_E64, SGPR_EACH_E64, N_OPS };
static const char *kNames[] = {"v_add_u32", "v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32",
                               "v_rndne_f64", "v_lshlrev_b64", "v_mad_u32_u24", "v_mul_lo_u32", "v_pk_mul_lo_u16", "v_pk_add_i16",
                               "v_cvt_f32_i32", "v_cvt_pk_u8_f32", "v_perm_b32", "v_bfe_i32", "v_dot2_i32_i16", "v_mov_b32 dpp",
                               "s_add_u32 (SALU only)", "1 v_add_u32 + 1 s_add_u32", "1 v_fma_f64 + 1 v_fma_f32", "v_pk_add_f32",
                               "v_pk_mul_f32", "v_mad_i32_i16", "v_lshl_or_b32", "v_and_or_b32", "1 v_fma_f64 + 1 v_add_u32",
                               "v_readlane_b32", "v_mad_i64_i32",
                               "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_sub_u32", "v_cndmask_b32",
                               "v_cmp_gt_u32", "v_min_u32", "v_max_i32", "v_mov_b32", "v_add3_u32", "v_or3_b32", "v_bfe_u32", "v_alignbit_b32",
                               "v_mul_f32", "v_add_f32", "v_fmac_f32", "v_mul_u32_u24", "v_add_co_u32", "v_lshl_add_u32", "v_cvt_i32_f32",
                               "v_rndne_f32", "v_med3_i32", "v_max_f32", "v_cvt_f32_ubyte0", "v_max3_f32", "v_fma_f32 (3 distinct src)",
                               "odd waves SALU, even waves v_fma_f64", "ds_read_b128", "ds_write_b128", "ds_read_b64",
                               "v_cndmask_b32 (vcc set in block)", "v_cndmask_b32 e64 sgpr mask", "v_cmp_gt_u32 e64 -> sgpr", "v_add_u32 v, s, v",
                               "v_lshlrev_b32 variable", "v_lshrrev_b64", "v_bfi_b32", "v_sub_f32", "v_mul_i32_i24", "v_mul_hi_u32",
                               "v_cvt_f32_i32 sdwa WORD_1", "v_add_u32 sdwa WORD_1", "v_mov_b32 sdwa sext WORD_0", "v_add_f32 e64 |src|",
                               "v_min_f32", "v_and_b32 literal", "v_fma_f32 sgpr const", "v_mul_f32 literal", "v_fmaak_f32", "v_fmamk_f32",
                               "v_add_f32 e64 -src", "v_subrev_u32", "v_xnor_b32", "v_lshlrev_b32 e32 var", "v_mul_f32 sgpr",
                               "v_cvt_f32_u32", "v_cvt_u32_f32", "v_fract_f32", "v_sub_co_u32", "v_addc_co_u32", "v_mbcnt_lo",
 "v_cndmask_b32_e64 ..., vcc", "v_cmp_e32 vcc + v_cndmask_e32 vcc (pair)", "v_cndmask_e32 dst != src", "v_cndmask_e32 vcc = -1", "v_cmp_gt_u32_e32 vcc", "v_cmp_e64 sgpr + v_cndmask_e64 sgpr (pair)", "ashr+xor+and+xor select (4 ops)", "s_mov vcc; 1 cndmask_e32; 7 v_add", "s_mov vcc; s_nop 7; 8 cndmask_e32", "8 x (s_and_b64 vcc; cndmask_e32)", "8 x (s_and_b64 vcc; cndmask_e64 vcc)", "8 x (s_and_b64 s[40:41]; cndmask_e64 s[40:41])"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters, uint32_t seed) {
    // 16 chains: registers v[i]; 64-bit chains use pairs
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6,
             a7 = a0 * 15 + 7;
    double d0 = a0 * 1e-3, d1 = a1 * 1e-3, d2 = a2 * 1e-3, d3 = a3 * 1e-3, d4 = a4 * 1e-3, d5 = a5 * 1e-3, d6 = a6 * 1e-3, d7 = a7 * 1e-3;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    const double c = 1.0000001;
    const uint32_t ci = 0x01010101u * (seed | 1);
    __builtin_amdgcn_s_barrier();
    uint64_t t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#define V8(ins) asm volatile(ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7) \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), \
                               "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(ci) : "scc", "vcc")
        // operands: %0-%7 = a*, %8-%15 = d*, %16-%19 = s*, %20 = c (double), %21 = ci
#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %21\n\t"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %21, %" #i "\n\t"
#define D(i) "%" D_##i
#define D_0 "8"
#define D_1 "9"
#define D_2 "10"
#define D_3 "11"
#define D_4 "12"
#define D_5 "13"
#define D_6 "14"
#define D_7 "15"
#define I_PKFMA(i) "v_pk_fma_f32 " D(i) ", " D(i) ", %20, " D(i) "\n\t"
#define I_PKADD(i) "v_pk_add_f32 " D(i) ", " D(i) ", %20\n\t"
#define I_PKMUL(i) "v_pk_mul_f32 " D(i) ", " D(i) ", %20\n\t"
#define I_FMA64(i) "v_fma_f64 " D(i) ", " D(i) ", %20, " D(i) "\n\t"
#define I_ADD64(i) "v_add_f64 " D(i) ", " D(i) ", %20\n\t"
#define I_MUL64(i) "v_mul_f64 " D(i) ", " D(i) ", %20\n\t"
#define I_CVT64(i) "v_cvt_f64_i32 " D(i) ", %" #i "\n\t"
#define I_RND64(i) "v_rndne_f64 " D(i) ", " D(i) "\n\t"
#define I_SHL64(i) "v_lshlrev_b64 " D(i) ", 1, " D(i) "\n\t"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %21, %" #i "\n\t"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %21\n\t"
#define I_PKMUL16(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %21\n\t"
#define I_PKADD16(i) "v_pk_add_i16 %" #i ", %" #i ", %21\n\t"
#define I_CVT32(i) "v_cvt_f32_i32 %" #i ", %" #i "\n\t"
#define I_CVTU8(i) "v_cvt_pk_u8_f32 %" #i ", %" #i ", 1, %" #i "\n\t"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %21, %21\n\t"
#define I_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 3, 16\n\t"
#define I_DOT2(i) "v_dot2_i32_i16 %" #i ", %" #i ", %21, %" #i "\n\t"
#define I_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_SADD(i) "s_add_u32 %16, %16, %17\n\ts_add_u32 %18, %18, %19\n\t"
#define I_MIXVS(i) "v_add_u32 %" #i ", %" #i ", %21\n\ts_add_u32 %16, %16, %17\n\t"
#define I_MIX6432(i) "v_fma_f64 " D(i) ", " D(i) ", %20, " D(i) "\n\tv_fma_f32 %" #i ", %" #i ", %21, %" #i "\n\t"
#define I_MIX64I(i) "v_fma_f64 " D(i) ", " D(i) ", %20, " D(i) "\n\tv_add_u32 %" #i ", %" #i ", %21\n\t"
#define I_MADI16(i) "v_mad_i32_i16 %" #i ", %" #i ", %21, %" #i "\n\t"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %21\n\t"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %21, %21\n\t"
#define I_RDLANE(i) "v_readlane_b32 %16, %" #i ", 3\n\tv_readlane_b32 %18, %" #i ", 5\n\t"
#define I_MADI64(i) "v_mad_i64_i32 " D(i) ", vcc, %" #i ", %21, " D(i) "\n\t"

#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %21\n\t"
#define I_OR(i) "v_or_b32 %" #i ", %" #i ", %21\n\t"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %21\n\t"
#define I_SHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define I_SHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n\t"
#define I_ASHR(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n\t"
#define I_SUB(i) "v_sub_u32 %" #i ", %" #i ", %21\n\t"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %21, vcc\n\t"
#define I_CMP(i) "v_cmp_gt_u32 vcc, %" #i ", %21\n\t"
#define I_MINU(i) "v_min_u32 %" #i ", %" #i ", %21\n\t"
#define I_MAXI(i) "v_max_i32 %" #i ", %" #i ", %21\n\t"
#define I_MOV(i) "v_mov_b32 %" #i ", %21\n\t"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %21, %21\n\t"
#define I_OR3(i) "v_or3_b32 %" #i ", %" #i ", %21, %21\n\t"
#define I_BFEU(i) "v_bfe_u32 %" #i ", %" #i ", 3, 16\n\t"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %21, 5\n\t"
#define I_MULF(i) "v_mul_f32 %" #i ", %" #i ", %21\n\t"
#define I_ADDF(i) "v_add_f32 %" #i ", %" #i ", %21\n\t"
#define I_FMAC(i) "v_fmac_f32 %" #i ", %21, %21\n\t"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %21\n\t"
#define I_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %21\n\t"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %21\n\t"
#define I_CVTI32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n\t"
#define I_RND32(i) "v_rndne_f32 %" #i ", %" #i "\n\t"
#define I_MED3(i) "v_med3_i32 %" #i ", %" #i ", %21, %21\n\t"
#define I_MAXF(i) "v_max_f32 %" #i ", %" #i ", %21\n\t"
#define I_UB0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n\t"
#define I_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %21, %21\n\t"
#define I_FMA3(i) "v_fma_f32 %" #i ", %1, %21, %2\n\t"
#define I_SADD1(i) "s_add_u32 %16, %16, %17\n\t"
#define I_LDSR128(i) "ds_read_b128 %[q" #i "], %[la]\n\t"

#define I_CNDV(i) "v_cndmask_b32 %" #i ", %" #i ", %21, vcc\n\t"
#define I_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %21, s[40:41]\n\t"
#define I_CMPS(i) "v_cmp_gt_u32_e64 s[40:41], %" #i ", %21\n\t"
#define I_ADDS(i) "v_add_u32 %" #i ", s40, %" #i "\n\t"
#define I_SHLV(i) "v_lshlrev_b32 %" #i ", %21, %" #i "\n\t"
#define I_SHR64(i) "v_lshrrev_b64 " D(i) ", 1, " D(i) "\n\t"
#define I_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %21, %21\n\t"
#define I_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %21\n\t"
#define I_MULI24(i) "v_mul_i32_i24 %" #i ", %" #i ", %21\n\t"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %21\n\t"
#define I_CVTSD(i) "v_cvt_f32_i32_sdwa %" #i ", sext(%" #i ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
#define I_ADDSD(i) "v_add_u32_sdwa %" #i ", %" #i ", %21 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define I_MOVSD(i) "v_mov_b32_sdwa %" #i ", sext(%" #i ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
#define I_ADDFA(i) "v_add_f32_e64 %" #i ", |%" #i "|, %21\n\t"
#define I_MINF(i) "v_min_f32 %" #i ", %" #i ", %21\n\t"
#define I_ANDL(i) "v_and_b32 %" #i ", 0x12345678, %" #i "\n\t"
#define I_FMAS(i) "v_fma_f32 %" #i ", %" #i ", s40, %" #i "\n\t"
#define I_MULL(i) "v_mul_f32 %" #i ", 0x3f9d70a4, %" #i "\n\t"
#define I_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %21, 0x3f9d70a4\n\t"
#define I_FMAMK(i) "v_fmamk_f32 %" #i ", %" #i ", 0x3f9d70a4, %21\n\t"
#define I_ADDFN(i) "v_add_f32_e64 %" #i ", -%" #i ", %21\n\t"
#define I_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %21\n\t"
#define I_XNOR(i) "v_xnor_b32 %" #i ", %" #i ", %21\n\t"
#define I_SHLV32(i) "v_lshlrev_b32_e32 %" #i ", %21, %" #i "\n\t"
#define I_MULFS(i) "v_mul_f32 %" #i ", s40, %" #i "\n\t"
#define I_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n\t"
#define I_CVTUF(i) "v_cvt_u32_f32 %" #i ", %" #i "\n\t"
#define I_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n\t"
#define I_SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %21\n\t"
#define I_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %21, vcc\n\t"
#define I_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %21, %" #i "\n\t"
#define PRE_VCC "s_mov_b64 vcc, 0x55555555\n\ts_mov_b64 s[40:41], 0x33333333\n\t"
#define V8P(ins) asm volatile(PRE_VCC ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7) \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), \
                               "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(ci) : "scc", "vcc", "s40", "s41")
#define BODYP(ins) V8P(ins); V8P(ins); V8P(ins); V8P(ins); V8P(ins); V8P(ins); V8P(ins); V8P(ins)

#define I_CND64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %21, vcc\n\t"
#define I_CMPCND(i) "v_cmp_gt_u32_e32 vcc, %" #i ", %21\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %21, vcc\n\t"
#define I_CNDO(i) "v_cndmask_b32_e32 %" #i ", %21, %21, vcc\n\t"
#define I_CMPE32(i) "v_cmp_gt_u32_e32 vcc, %" #i ", %21\n\t"
#define I_CMPSCNDS(i) "v_cmp_gt_u32_e64 s[40:41], %" #i ", %21\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %21, s[40:41]\n\t"
#define I_SEL4(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n\tv_xor_b32 %" #i ", %21, %" #i "\n\tv_and_b32 %" #i ", %21, %" #i "\n\tv_xor_b32 %" #i ", %21, %" #i "\n\t"
#define PRE_VCC1 "s_mov_b64 vcc, -1\n\t"
#define V8Q(ins) asm volatile(PRE_VCC1 ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7) \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), \
                               "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(ci) : "scc", "vcc", "s40", "s41")
#define BODYQ(ins) V8Q(ins); V8Q(ins); V8Q(ins); V8Q(ins); V8Q(ins); V8Q(ins); V8Q(ins); V8Q(ins)

#define I_VCC1(i) I_ADD(i)
#define I_SAND_CND(i) "s_and_b64 vcc, s[42:43], exec\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %21, vcc\n\t"
#define I_SAND_CND64(i) "s_and_b64 vcc, s[42:43], exec\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %21, vcc\n\t"
#define I_SAND_CNDS(i) "s_and_b64 s[40:41], s[42:43], exec\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %21, s[40:41]\n\t"
#define VCLOB : "scc", "vcc", "s40", "s41", "s42", "s43"
#define VOUT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), \
                               "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(ci)
#define BODY(ins) V8(ins); V8(ins); V8(ins); V8(ins); V8(ins); V8(ins); V8(ins); V8(ins)
        if (OP == ADD_U32) { BODY(I_ADD); }
        if (OP == FMA_F32) { BODY(I_FMA32); }
        if (OP == PK_FMA_F32) { BODY(I_PKFMA); }
        if (OP == FMA_F64) { BODY(I_FMA64); }
        if (OP == ADD_F64) { BODY(I_ADD64); }
        if (OP == MUL_F64) { BODY(I_MUL64); }
        if (OP == CVT_F64_I32) { BODY(I_CVT64); }
        if (OP == RNDNE_F64) { BODY(I_RND64); }
        if (OP == LSHL_B64) { BODY(I_SHL64); }
        if (OP == MAD_U24) { BODY(I_MAD24); }
        if (OP == MUL_LO_U32) { BODY(I_MULLO); }
        if (OP == PK_MUL_LO_U16) { BODY(I_PKMUL16); }
        if (OP == PK_ADD_I16) { BODY(I_PKADD16); }
        if (OP == CVT_F32_I32) { BODY(I_CVT32); }
        if (OP == CVT_PK_U8) { BODY(I_CVTU8); }
        if (OP == PERM_B32) { BODY(I_PERM); }
        if (OP == BFE_I32) { BODY(I_BFE); }
        if (OP == DOT2_I32_I16) { BODY(I_DOT2); }
        if (OP == MOV_DPP) { BODY(I_DPP); }
        if (OP == SALU_ADD) { BODY(I_SADD); }        // 2 per slot -> 128 per iteration
        if (OP == MIX_VALU_SALU) { BODY(I_MIXVS); }  // 128 per iteration
        if (OP == MIX_F64_F32) { BODY(I_MIX6432); }  // 128 per iteration
        if (OP == MIX_F64_INT) { BODY(I_MIX64I); }
        if (OP == PK_ADD_F32) { BODY(I_PKADD); }
        if (OP == PK_MUL_F32) { BODY(I_PKMUL); }
        if (OP == MAD_I32_I16) { BODY(I_MADI16); }
        if (OP == LSHL_OR) { BODY(I_LSHLOR); }
        if (OP == AND_OR) { BODY(I_ANDOR); }
        if (OP == READLANE) { BODY(I_RDLANE); }
        if (OP == MAD_I64_I32) { BODY(I_MADI64); }
        if (OP == AND_B32) { BODY(I_AND); }
        if (OP == OR_B32) { BODY(I_OR); }
        if (OP == XOR_B32) { BODY(I_XOR); }
        if (OP == LSHL_B32) { BODY(I_SHL); }
        if (OP == LSHR_B32) { BODY(I_SHR); }
        if (OP == ASHR_I32) { BODY(I_ASHR); }
        if (OP == SUB_U32) { BODY(I_SUB); }
        if (OP == CNDMASK) { BODY(I_CND); }
        if (OP == CMP_GT) { BODY(I_CMP); }
        if (OP == MIN_U32) { BODY(I_MINU); }
        if (OP == MAX_I32) { BODY(I_MAXI); }
        if (OP == MOV_B32) { BODY(I_MOV); }
        if (OP == ADD3_U32) { BODY(I_ADD3); }
        if (OP == OR3_B32) { BODY(I_OR3); }
        if (OP == BFE_U32) { BODY(I_BFEU); }
        if (OP == ALIGNBIT) { BODY(I_ALIGN); }
        if (OP == MUL_F32) { BODY(I_MULF); }
        if (OP == ADD_F32) { BODY(I_ADDF); }
        if (OP == FMAC_F32) { BODY(I_FMAC); }
        if (OP == MUL_U24) { BODY(I_MUL24); }
        if (OP == ADD_CO) { BODY(I_ADDCO); }
        if (OP == LSHL_ADD) { BODY(I_LSHLADD); }
        if (OP == CVT_I32_F32) { BODY(I_CVTI32); }
        if (OP == RNDNE_F32) { BODY(I_RND32); }
        if (OP == MED3_I32) { BODY(I_MED3); }
        if (OP == MAX_F32) { BODY(I_MAXF); }
        if (OP == CVT_UB0) { BODY(I_UB0); }
        if (OP == MAX3_F32) { BODY(I_MAX3F); }
        if (OP == FMA_F32_3SRC) { BODY(I_FMA3); }
        if (OP == SALU_VALU_2W) {   // wave-uniform branch inside the asm: odd waves run the scalar adds, even waves the fp64 FMAs
            const uint32_t odd = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1;
#define V8B(ia, ib) asm volatile("s_cmp_eq_u32 %22, 0\n\ts_cbranch_scc1 1f\n\t" ia(0) ia(1) ia(2) ia(3) ia(4) ia(5) ia(6) ia(7) "s_branch 2f\n1:\n\t" ib(0) ib(1) ib(2) ib(3) ib(4) ib(5) ib(6) ib(7) "2:\n\t" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), \
                               "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(ci), "s"(odd) : "scc", "vcc")
            V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64);
            V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64); V8B(I_SADD1, I_FMA64);
        }
        if (OP == LDS_RD128 || OP == LDS_WR128 || OP == LDS_RD64) {
            typedef uint32_t u4v __attribute__((ext_vector_type(4))); typedef uint32_t u2v __attribute__((ext_vector_type(2)));
            u4v q = {a0, a1, a2, a3};
            uint32_t la = (threadIdx.x * 16) & 0x3FFF;
            for (int r8 = 0; r8 < 8; ++r8) {
                if (OP == LDS_RD128) asm volatile(REP4("ds_read_b128 %0, %1\n\t") REP4("ds_read_b128 %0, %1 offset:4096\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(q) : "v"(la) : "memory");
                if (OP == LDS_WR128) asm volatile(REP4("ds_write_b128 %1, %0\n\t") REP4("ds_write_b128 %1, %0 offset:4096\n\t") "s_waitcnt lgkmcnt(0)" : : "v"(q), "v"(la) : "memory");
                if (OP == LDS_RD64) { u2v q2; asm volatile(REP4("ds_read_b64 %0, %1\n\t") REP4("ds_read_b64 %0, %1 offset:4096\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(q2) : "v"(la >> 1) : "memory"); q.x ^= q2.x; }
            }
            a0 ^= q.x;
        }
        if (OP == CND_VCC_SET) { BODYP(I_CNDV); }
        if (OP == CND_SGPR) { BODYP(I_CNDS); }
        if (OP == CMP_SGPR) { BODYP(I_CMPS); }
        if (OP == ADD_SGPR) { BODYP(I_ADDS); }
        if (OP == LSHL_VAR) { BODY(I_SHLV); }
        if (OP == LSHR_B64) { BODY(I_SHR64); }
        if (OP == BFI) { BODY(I_BFI); }
        if (OP == SUB_F32) { BODY(I_SUBF); }
        if (OP == MUL_I24) { BODY(I_MULI24); }
        if (OP == MUL_HI) { BODY(I_MULHI); }
        if (OP == CVT_SDWA) { BODY(I_CVTSD); }
        if (OP == ADD_SDWA) { BODY(I_ADDSD); }
        if (OP == MOV_SDWA) { BODY(I_MOVSD); }
        if (OP == ADD_F32_ABS) { BODY(I_ADDFA); }
        if (OP == MIN_F32) { BODY(I_MINF); }
        if (OP == AND_LIT) { BODY(I_ANDL); }
        if (OP == FMA_SGPR) { BODYP(I_FMAS); }
        if (OP == MUL_LIT) { BODY(I_MULL); }
        if (OP == FMAAK) { BODY(I_FMAAK); }
        if (OP == FMAMK) { BODY(I_FMAMK); }
        if (OP == ADD_F32_NEG) { BODY(I_ADDFN); }
        if (OP == SUBREV) { BODY(I_SUBREV); }
        if (OP == XNOR) { BODY(I_XNOR); }
        if (OP == LSHL_B32_E32VAR) { BODY(I_SHLV32); }
        if (OP == MUL_F32_SGPR) { BODYP(I_MULFS); }
        if (OP == CVT_F32_U32) { BODY(I_CVTFU); }
        if (OP == CVT_U32_F32) { BODY(I_CVTUF); }
        if (OP == FRACT) { BODY(I_FRACT); }
        if (OP == SUB_CO) { BODY(I_SUBCO); }
        if (OP == ADDC) { BODYP(I_ADDC); }
        if (OP == MBCNT) { BODY(I_MBCNT); }
        if (OP == CND_E64_VCC) { BODYP(I_CND64V); }
        if (OP == CMP_CND_PAIR) { BODY(I_CMPCND); }
        if (OP == CND_E32_OTHERDST) { BODYP(I_CNDO); }
        if (OP == CND_E32_ALLONES) { BODYQ(I_CNDV); }
        if (OP == CMP_E32_ONLY) { BODY(I_CMPE32); }
        if (OP == CMPS_CNDS_PAIR) { BODY(I_CMPSCNDS); }
        if (OP == BFI_SEL) { BODY(I_SEL4); }
        if (OP == VCC1) { for (int r8 = 0; r8 < 8; ++r8) asm volatile("s_mov_b64 vcc, 0x55555555\n\t" I_CNDV(0) I_ADD(1) I_ADD(2) I_ADD(3) I_ADD(4) I_ADD(5) I_ADD(6) I_ADD(7) VOUT VCLOB); }
        if (OP == VCC_NOP) { for (int r8 = 0; r8 < 8; ++r8) asm volatile("s_mov_b64 vcc, 0x55555555\n\ts_nop 7\n\t" I_CNDV(0) I_CNDV(1) I_CNDV(2) I_CNDV(3) I_CNDV(4) I_CNDV(5) I_CNDV(6) I_CNDV(7) VOUT VCLOB); }
        if (OP == VCC_EACH) { for (int r8 = 0; r8 < 8; ++r8) asm volatile("s_mov_b64 s[42:43], 0x55555555\n\t" I_SAND_CND(0) I_SAND_CND(1) I_SAND_CND(2) I_SAND_CND(3) I_SAND_CND(4) I_SAND_CND(5) I_SAND_CND(6) I_SAND_CND(7) VOUT VCLOB); }
        if (OP == VCC_EACH_E64) { for (int r8 = 0; r8 < 8; ++r8) asm volatile("s_mov_b64 s[42:43], 0x55555555\n\t" I_SAND_CND64(0) I_SAND_CND64(1) I_SAND_CND64(2) I_SAND_CND64(3) I_SAND_CND64(4) I_SAND_CND64(5) I_SAND_CND64(6) I_SAND_CND64(7) VOUT VCLOB); }
        if (OP == SGPR_EACH_E64) { for (int r8 = 0; r8 < 8; ++r8) asm volatile("s_mov_b64 s[42:43], 0x55555555\n\t" I_SAND_CNDS(0) I_SAND_CNDS(1) I_SAND_CNDS(2) I_SAND_CNDS(3) I_SAND_CNDS(4) I_SAND_CNDS(5) I_SAND_CNDS(6) I_SAND_CNDS(7) VOUT VCLOB); }
    }
    uint64_t t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
    r ^= __double_as_longlong(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[2 * gw] = t1 - t0; out[2 * gw + 1] = r; }
}

template <int OP>
static void run(uint64_t *d, std::vector<uint64_t> &h) {
    const int iters = 2000;
    const int per_iter = (OP == SALU_ADD || OP == MIX_VALU_SALU || OP == MIX_F64_F32 || OP == MIX_F64_INT || OP == READLANE || OP == CMP_CND_PAIR || OP == CMPS_CNDS_PAIR) ? 128 : (OP == BFI_SEL ? 256 : 64);
    if (OP == SALU_ADD || OP == MIX_VALU_SALU) { }
    printf("%-28s", kNames[OP]);
    for (int wps = 1; wps <= 8; wps *= 2) {
        if (wps == 8 && OP > 3) continue;
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256, 20480>>>(d, iters, 3);            // warm
        hipEventRecord(e0);
        k<OP><<<blocks, 256, 20480>>>(d, iters, 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, blocks * 4 * 16, hipMemcpyDeviceToHost);
        std::vector<uint64_t> cyc(blocks * 4);
        for (int i = 0; i < blocks * 4; ++i) cyc[i] = h[2 * i];
        std::sort(cyc.begin(), cyc.end());
        const double med = (double)cyc[cyc.size() / 2], mx = (double)cyc.back();
        const double n = (double)iters * per_iter * wps;
        printf("  w%d: %5.2f (max %5.2f) %6.3f ms/wps", wps, med / n, mx / n, ms / wps);
    }
    printf("\n");
}

template <int OP>
static void run_all(uint64_t *d, std::vector<uint64_t> &h) {
    run<OP>(d, h);
    #ifndef LAST_OP
#define LAST_OP (N_OPS - 1)
#endif
    if constexpr (OP + 1 <= LAST_OP) run_all<OP + 1>(d, h);
}

int main() {
    uint64_t *d; hipMalloc(&d, 2048 * 4 * 16);
    std::vector<uint64_t> h(2048 * 4 * 2);
#ifndef FIRST_OP
#define FIRST_OP 0
#endif
    run_all<FIRST_OP>(d, h);
    return 0;
}
