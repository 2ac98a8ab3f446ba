// Issue rate of the integer VALU instructions the tile search leans on (gfx950): one workgroup of 4 waves per CU-SIMD set, long unrolled
// chains on 8 independent accumulators; prints quad-cycles per wave-instruction relative to v_add_u32.   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN(NAME, ASM, ...) \
__global__ __launch_bounds__(256) void NAME(int *out, int a, int b, int iters) { \
  int x0 = threadIdx.x + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
  for (int i = 0; i < iters; i++) { \
    _Pragma("unroll") for (int k = 0; k < 16; k++) { \
      asm volatile(ASM : "+v"(x0) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x1) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x2) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x3) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); \
      asm volatile(ASM : "+v"(x4) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x5) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x6) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); asm volatile(ASM : "+v"(x7) : "v"(b), "s"(a) : "vcc", "s10", "s11", "s12"); \
    } } \
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7; }
CHAIN(k_add, "v_add_u32 %0, %0, %1")
CHAIN(k_mul24, "v_mul_i32_i24 %0, %0, %1")
CHAIN(k_mad24, "v_mad_i32_i24 %0, %0, %1, %0")
CHAIN(k_mullo, "v_mul_lo_u32 %0, %0, %1")
CHAIN(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
CHAIN(k_pkadd, "v_pk_add_u16 %0, %0, %1")
CHAIN(k_pkmul, "v_pk_mul_lo_u16 %0, %0, %1")
CHAIN(k_pkmad, "v_pk_mad_i16 %0, %0, %1, %0")
CHAIN(k_dot2, "v_dot2_i32_i16 %0, %0, %1, %0")
CHAIN(k_sad16, "v_sad_u16 %0, %0, %1, %0")
CHAIN(k_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
CHAIN(k_dpprow, "v_add_u32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf")
CHAIN(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
CHAIN(k_add3, "v_add3_u32 %0, %0, %1, %0")
CHAIN(k_ashr, "v_ashrrev_i32 %0, 12, %0")
CHAIN(k_bfe, "v_bfe_i32 %0, %0, 0, 24")
CHAIN(k_med3, "v_med3_i32 %0, %0, %1, %2")
CHAIN(k_cvtf, "v_cvt_f32_u32 %0, %0")
CHAIN(k_sqrt, "v_sqrt_f32 %0, %0")
CHAIN(k_rcp, "v_rcp_f32 %0, %0")
CHAIN(k_lshl64, "v_lshlrev_b32 %0, 3, %0")
CHAIN(k_cnd64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
CHAIN(k_cmp, "v_cmp_lt_i32 vcc, %0, %1")
CHAIN(k_cmps, "v_cmp_lt_i32_e64 s[10:11], %0, %1")
CHAIN(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
CHAIN(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
CHAIN(k_mov, "v_mov_b32 %0, %1")
CHAIN(k_sub, "v_sub_u32 %0, %0, %1")
CHAIN(k_and, "v_and_b32 %0, %0, %1")
CHAIN(k_max, "v_max_i32 %0, %0, %1")
CHAIN(k_lshl, "v_lshlrev_b32 %0, 2, %0")
CHAIN(k_adds, "v_add_u32 %0, %2, %0")
CHAIN(k_addlit, "v_add_u32 %0, 0x800, %0")
CHAIN(k_mad24s, "v_mad_i32_i24 %0, %0, %2, %1")
CHAIN(k_mul24lit, "v_mul_i32_i24 %0, 0xb50, %0")
CHAIN(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %1")
CHAIN(k_sube64, "v_sub_u32_e64 %0, %0, %1")
CHAIN(k_perm, "v_perm_b32 %0, %0, %1, %0")
CHAIN(k_readlane, "v_readlane_b32 s12, %0, 3")
CHAIN(k_salu, "s_add_u32 s12, s12, %2")
CHAIN(k_cmpcnd, "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
CHAIN(k_cmpcnds, "v_cmp_lt_i32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
CHAIN(k_cndvccdef, "v_cndmask_b32 %0, %1, %0, vcc")
CHAIN(k_min, "v_min_i32 %0, %0, %1")
CHAIN(k_or, "v_or_b32 %0, %0, %1")
CHAIN(k_xor, "v_xor_b32 %0, %0, %1")
CHAIN(k_lshr, "v_lshrrev_b32 %0, 2, %0")
CHAIN(k_lshlv, "v_lshlrev_b32 %0, %1, %0")
CHAIN(k_ashrv, "v_ashrrev_i32 %0, %1, %0")
CHAIN(k_fma, "v_fma_f32 %0, %0, %1, %0")
CHAIN(k_fadd, "v_add_f32 %0, %0, %1")
CHAIN(k_mulu24, "v_mul_u32_u24 %0, %0, %1")
CHAIN(k_addlsh, "v_add_lshl_u32 %0, %0, %1, 2")
CHAIN(k_and_or, "v_and_or_b32 %0, %0, %1, %0")
CHAIN(k_max3, "v_max3_i32 %0, %0, %1, %0")
CHAIN(k_cnd_e64vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
CHAIN(k_pkmin, "v_pk_min_i16 %0, %0, %1")
CHAIN(k_subrevs, "v_subrev_u32 %0, %2, %0")
CHAIN(k_sad32, "v_sad_u32 %0, %0, %1, 0")
CHAIN(k_fmax, "v_max_f32 %0, %0, %1")
CHAIN(k_fmin, "v_min_f32 %0, %0, %1")
CHAIN(k_fmed3, "v_med3_f32 %0, %0, %1, %1")
CHAIN(k_fmax3, "v_max3_f32 %0, %0, %1, %0")
CHAIN(k_fcmp, "v_cmp_lt_f32 vcc, %0, %1")
CHAIN(k_fcmpcnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
CHAIN(k_fmul, "v_mul_f32 %0, %0, %1")
CHAIN(k_fsub, "v_sub_f32 %0, %0, %1")
CHAIN(k_fmac, "v_fmac_f32 %0, %1, %1")
CHAIN(k_cvt_f_i, "v_cvt_f32_i32 %0, %0")
CHAIN(k_cvt_i_f, "v_cvt_i32_f32 %0, %0")
CHAIN(k_cvt_f_ub0, "v_cvt_f32_ubyte0 %0, %0")
CHAIN(k_fabsadd, "v_add_f32 %0, |%0|, %1")
CHAIN(k_fnegadd, "v_sub_f32 %0, %1, %0")
CHAIN(k_ffloor, "v_floor_f32 %0, %0")
CHAIN(k_frndne, "v_rndne_f32 %0, %0")
CHAIN(k_fmaxlit, "v_max_f32 %0, 0, %0")
CHAIN(k_fminlit, "v_min_f32 %0, 0x447fc000, %0")
CHAIN(k_h_add, "v_add_f16 %0, %0, %1")
CHAIN(k_h_max, "v_max_f16 %0, %0, %1")
CHAIN(k_pkh_add, "v_pk_add_f16 %0, %0, %1")
CHAIN(k_pkh_max, "v_pk_max_f16 %0, %0, %1")
CHAIN(k_pkh_min, "v_pk_min_f16 %0, %0, %1")
CHAIN(k_pkh_fma, "v_pk_fma_f16 %0, %0, %1, %0")
CHAIN(k_pkh_mul, "v_pk_mul_f16 %0, %0, %1")
CHAIN(k_maxu, "v_max_u32 %0, %0, %1")
CHAIN(k_maxu16, "v_max_u16 %0, %0, %1")
CHAIN(k_addu16, "v_add_u16 %0, %0, %1")
CHAIN(k_subu16, "v_sub_u16 %0, %0, %1")
CHAIN(k_bfi, "v_bfi_b32 %0, %0, %1, %0")
CHAIN(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
CHAIN(k_xad, "v_xad_u32 %0, %0, %1, %0")
CHAIN(k_lshlor, "v_lshl_or_b32 %0, %0, 2, %1")
CHAIN(k_or3, "v_or3_b32 %0, %0, %1, %0")
CHAIN(k_not, "v_not_b32 %0, %0")
CHAIN(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
CHAIN(k_ffbh, "v_ffbh_u32 %0, %0")
CHAIN(k_absdiff, "v_sub_u32 %0, %0, %1\n v_max_i32 %0, %0, %1")
CHAIN(k_dppmov, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
CHAIN(k_dppfadd, "v_add_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf")
CHAIN(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
CHAIN(k_movsdwa, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
__global__ __launch_bounds__(256) void k_mad64(int *out, int a, int b, int iters) {
  unsigned long long x0 = threadIdx.x + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 32; k++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x0) : "v"(b), "v"(a) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x1) : "v"(b), "v"(a) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x2) : "v"(b), "v"(a) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x3) : "v"(b), "v"(a) : "vcc");
    } }
  out[blockIdx.x * 256 + threadIdx.x] = (int)(x0 ^ x1 ^ x2 ^ x3);
}
template <typename K> static double run(K kern, int *d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(256), 0, 0, d, 3, 5, 16);
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(256), 0, 0, d, 3, 5, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  int *d; hipMalloc(&d, 256 * 4 * 256 * 4);
  const int iters = 2000;
  const double base = run(k_add, d, iters);
  printf("v_add_u32 %.3f ms for %d x 128 instr per wave, 4 waves/SIMD -> %.2f GHz-equivalent\n", base, iters, iters * 128.0 * 4 * 4 / (base * 1e6));
#define R(k) printf("%-10s %.3f ms  x%.2f\n", #k, run(k, d, iters), run(k, d, iters) / base)
  R(k_add); R(k_cnd64); R(k_cmp); R(k_cmps); R(k_addco); R(k_addc); R(k_mov); R(k_sub); R(k_and); R(k_max); R(k_lshl); R(k_adds); R(k_addlit); R(k_mad24s); R(k_mul24lit); R(k_lshladd); R(k_sube64); R(k_perm); R(k_readlane); R(k_salu); R(k_cmpcnd); R(k_cmpcnds); R(k_cndvccdef); R(k_cnd_e64vcc); R(k_min); R(k_pkmin); R(k_or); R(k_xor); R(k_lshr); R(k_lshlv); R(k_ashrv); R(k_fma); R(k_fadd); R(k_mulu24); R(k_addlsh); R(k_and_or); R(k_subrevs); R(k_max3); R(k_sad32); R(k_cmpcnd); R(k_cmpcnds); R(k_cndvccdef); R(k_cnd_e64vcc); R(k_min); R(k_pkmin); R(k_or); R(k_xor); R(k_lshr); R(k_lshlv); R(k_ashrv); R(k_fma); R(k_fadd); R(k_mulu24); R(k_addlsh); R(k_and_or); R(k_subrevs); R(k_max3); R(k_sad32); R(k_mul24); R(k_mad24); R(k_mullo); R(k_mulhi); R(k_pkadd); R(k_pkmul); R(k_pkmad); R(k_dot2); R(k_sad16); R(k_dpp); R(k_dpprow); R(k_cndmask); R(k_add3); R(k_ashr); R(k_bfe); R(k_med3); R(k_cvtf); R(k_sqrt); R(k_rcp);
  printf("--- round 6: the float pipe, 16-bit, bit ops ---\n"); R(k_fmax); R(k_fmin); R(k_fmed3); R(k_fmax3); R(k_fcmp); R(k_fcmpcnd); R(k_fmul); R(k_fsub); R(k_fmac); R(k_cvt_f_i); R(k_cvt_i_f); R(k_cvt_f_ub0); R(k_fabsadd); R(k_fnegadd); R(k_ffloor); R(k_frndne); R(k_fmaxlit); R(k_fminlit); R(k_h_add); R(k_h_max); R(k_pkh_add); R(k_pkh_max); R(k_pkh_min); R(k_pkh_fma); R(k_pkh_mul); R(k_maxu); R(k_maxu16); R(k_addu16); R(k_subu16); R(k_bfi); R(k_alignbit); R(k_xad); R(k_lshlor); R(k_or3); R(k_not); R(k_bcnt); R(k_ffbh); R(k_absdiff); R(k_dppmov); R(k_dppfadd); R(k_sdwa); R(k_movsdwa);
  printf("%-10s %.3f ms  x%.2f (per instr; 4 chains)\n", "k_mad64", run(k_mad64, d, iters), run(k_mad64, d, iters) / base);
  return 0;
}
