// valu_microbench.hip -- issue-rate microbenchmark for the VALU instruction classes the DXT
// encoders are made of (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip
// Prints wave-instructions per cycle per CU (assuming 2.4 GHz; the measured ratio between rows is what matters).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define BODY(ASM)                                                                                         \
        for (int it = 0; it < iters; it++) {                                                              \
                REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                                   : "v"(b0), "v"(b1) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)  \
        }

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256) void bench(float *out, int iters)
{
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        float b0 = 1.0001f, b1 = 0.9999f;
        if (K == 0) BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %9")
        if (K == 1) BODY("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %9, %8\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %9, %8")
        if (K == 2) BODY("v_min3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %9, %8\n v_min3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %9, %8\n v_min3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %9, %8\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %9, %8")
        if (K == 3) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %9\n v_cmp_gt_f32 s[24:25], %2, %8\n v_cmp_gt_f32 s[26:27], %3, %9\n v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[22:23], %5, %9\n v_cmp_gt_f32 s[24:25], %6, %8\n v_cmp_gt_f32 s[26:27], %7, %9")
        if (K == 4) BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %9, vcc")
        if (K == 5) BODY("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7")
        if (K == 6) BODY("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_addc_co_u32 %7, vcc, %7, %7, vcc")
        if (K == 7) BODY("v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %9\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %9\n v_lshl_or_b32 %4, %4, 2, %8\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %8\n v_lshl_or_b32 %7, %7, 2, %9")
        if (K == 8) BODY("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9")
        if (K == 9) BODY("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc")

        if (K == 10) BODY("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9")
        if (K == 11) BODY("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %9\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %9")
        if (K == 12) BODY("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %9\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %9\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %9\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %9")
        if (K == 13) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %9, s[20:21]\n v_cmp_gt_f32 s[22:23], %2, %8\n v_cndmask_b32 %3, %3, %9, s[22:23]\n v_cmp_gt_f32 s[24:25], %4, %8\n v_cndmask_b32 %5, %5, %9, s[24:25]\n v_cmp_gt_f32 s[26:27], %6, %8\n v_cndmask_b32 %7, %7, %9, s[26:27]")
        if (K == 14) BODY("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_and_b32 %2, %2, %8\n v_or_b32 %3, %3, %9\n v_xor_b32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_and_b32 %6, %6, %8\n v_xor_b32 %7, %7, %9")
        if (K == 15) BODY("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9")
        if (K == 16) BODY("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %9, %8\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %9, %8\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %9, %8")
        if (K == 17) BODY("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %9, %8\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %9, %8\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %9, %8\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %9, %8")
        if (K == 18) BODY("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8")
        if (K == 19) BODY("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7")
        if (K == 20) BODY("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %9, %8\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %9, %8\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %9, %8\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %9, %8")
        if (K == 21) BODY("v_fract_f32 %0, %0\n v_rndne_f32 %1, %1\n v_fract_f32 %2, %2\n v_trunc_f32 %3, %3\n v_fract_f32 %4, %4\n v_floor_f32 %5, %5\n v_fract_f32 %6, %6\n v_rndne_f32 %7, %7")
        if (K == 22) BODY("v_add_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %9 row_shr:1 row_mask:0xf bank_mask:0xf")
        if (K == 23) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n s_and_b64 s[22:23], s[20:21], s[24:25]\n v_cmp_gt_f32 s[24:25], %2, %8\n s_or_b64 s[26:27], s[22:23], s[24:25]\n v_cmp_gt_f32 s[20:21], %4, %8\n s_and_b64 s[22:23], s[20:21], s[26:27]\n v_cmp_gt_f32 s[24:25], %6, %8\n s_or_b64 s[26:27], s[22:23], s[24:25]")
        if (K == 24) BODY("v_add_f32 %0, %0, %8\n v_cmp_gt_f32 s[20:21], %1, %9\n v_add_f32 %2, %2, %8\n v_cmp_gt_f32 s[22:23], %3, %9\n v_add_f32 %4, %4, %8\n v_cmp_gt_f32 s[24:25], %5, %9\n v_add_f32 %6, %6, %8\n v_cmp_gt_f32 s[26:27], %7, %9")
        if (K == 25) BODY("v_max3_f32 %0, %0, %8, %9\n v_add_f32 %1, %1, %9\n v_min3_f32 %2, %2, %8, %9\n v_add_f32 %3, %3, %9\n v_max3_f32 %4, %4, %8, %9\n v_add_f32 %5, %5, %9\n v_min3_f32 %6, %6, %8, %9\n v_add_f32 %7, %7, %9")

        if (K == 30) BODY("v_lshrrev_b32 %0, 31, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 31, %2\n v_lshlrev_b32 %3, 1, %3\n v_ashrrev_i32 %4, 31, %4\n v_lshlrev_b32 %5, 1, %5\n v_ashrrev_i32 %6, 31, %6\n v_lshrrev_b32 %7, 3, %7")
        if (K == 31) BODY("v_add_u32 %0, %0, %8\n v_sub_u32 %1, %1, %9\n v_add_u32 %2, %2, %8\n v_sub_u32 %3, %3, %9\n v_add_u32 %4, %4, %8\n v_sub_u32 %5, %5, %9\n v_add_u32 %6, %6, %8\n v_sub_u32 %7, %7, %9")
        if (K == 32) BODY("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %9, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %9, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %9, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %9, 31")
        if (K == 33) BODY("v_bfi_b32 %0, %0, %8, %9\n v_bfi_b32 %1, %1, %9, %8\n v_bfi_b32 %2, %2, %8, %9\n v_bfi_b32 %3, %3, %9, %8\n v_bfi_b32 %4, %4, %8, %9\n v_bfi_b32 %5, %5, %9, %8\n v_bfi_b32 %6, %6, %8, %9\n v_bfi_b32 %7, %7, %9, %8")
        if (K == 34) BODY("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %9, %8\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %9, %8\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %9, %8")
        if (K == 35) BODY("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %9\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %9\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %9\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %9")
        if (K == 36) BODY("v_mul_u32_u24 %0, %0, %8\n v_mad_u32_u24 %1, %1, %9, %8\n v_mul_u32_u24 %2, %2, %8\n v_mad_u32_u24 %3, %3, %9, %8\n v_mul_u32_u24 %4, %4, %8\n v_mad_u32_u24 %5, %5, %9, %8\n v_mul_u32_u24 %6, %6, %8\n v_mad_u32_u24 %7, %7, %9, %8")
        if (K == 37) BODY("v_sub_f32 %0, %0, %8\n v_alignbit_b32 %1, %1, %0, 31\n v_sub_f32 %2, %2, %8\n v_alignbit_b32 %3, %3, %2, 31\n v_sub_f32 %4, %4, %8\n v_alignbit_b32 %5, %5, %4, 31\n v_sub_f32 %6, %6, %8\n v_alignbit_b32 %7, %7, %6, 31")
        if (K == 38) BODY("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_cmp_gt_f32 s[20:21], %3, %9\n v_mul_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_cmp_gt_f32 s[22:23], %7, %9")
        if (K == 39) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %9\n v_cmp_gt_f32 s[24:25], %2, %8\n v_cmp_gt_f32 s[26:27], %3, %9\n v_mul_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9")
        if (K == 40) BODY("v_lshl_add_u32 %0, %0, 1, %8\n v_add3_u32 %1, %1, %9, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_add3_u32 %3, %3, %9, %8\n v_xad_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %9, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_xad_u32 %7, %7, %9, %8")
        if (K == 41) BODY("v_cvt_f32_u32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_i32 %7, %7")
        if (K == 42) BODY("v_or_b32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %1, %9, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_or_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_or_b32_sdwa %3, %9, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_or_b32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_or_b32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_or_b32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")

        if (K == 50) BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9")
        if (K == 51) BODY("v_cmp_gt_f32 vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc")
        if (K == 52) BODY("v_add_f32 %0, %0, %8\n v_cmp_gt_f32 vcc, %0, %9\n v_add_f32 %2, %2, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_add_f32 %4, %4, %8\n v_cmp_gt_f32 vcc, %4, %9\n v_add_f32 %6, %6, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc")

        if (K == 60) BODY("v_mul_f32 %0, 0x3e812345, %0\n v_mul_f32 %1, 0x3e812345, %1\n v_mul_f32 %2, 0x3e812345, %2\n v_mul_f32 %3, 0x3e812345, %3\n v_mul_f32 %4, 0x3e812345, %4\n v_mul_f32 %5, 0x3e812345, %5\n v_mul_f32 %6, 0x3e812345, %6\n v_mul_f32 %7, 0x3e812345, %7")
        if (K == 61) BODY("v_fmaak_f32 %0, %0, %8, 0x3e812345\n v_fmaak_f32 %1, %1, %9, 0x3e812345\n v_fmaak_f32 %2, %2, %8, 0x3e812345\n v_fmaak_f32 %3, %3, %9, 0x3e812345\n v_fmaak_f32 %4, %4, %8, 0x3e812345\n v_fmaak_f32 %5, %5, %9, 0x3e812345\n v_fmaak_f32 %6, %6, %8, 0x3e812345\n v_fmaak_f32 %7, %7, %9, 0x3e812345")
        if (K == 62) BODY("v_add_f32_e64 %0, %0, %8 clamp\n v_add_f32_e64 %1, %1, %9 clamp\n v_add_f32_e64 %2, %2, %8 clamp\n v_add_f32_e64 %3, %3, %9 clamp\n v_add_f32_e64 %4, %4, %8 clamp\n v_add_f32_e64 %5, %5, %9 clamp\n v_add_f32_e64 %6, %6, %8 clamp\n v_add_f32_e64 %7, %7, %9 clamp")
        if (K == 63) BODY("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s21, %3\n v_mul_f32 %4, s20, %4\n v_mul_f32 %5, s21, %5\n v_mul_f32 %6, s20, %6\n v_mul_f32 %7, s21, %7")
        if (K == 64) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cmp_gt_f32 s[22:23], %2, %8\n s_or_b64 s[26:27], s[24:25], s[22:23]\n v_cmp_gt_f32 s[20:21], %4, %8\n s_and_b64 s[24:25], s[20:21], s[26:27]\n v_cmp_gt_f32 s[22:23], %6, %8\n s_or_b64 s[26:27], s[24:25], s[22:23]")
        if (K == 65) BODY("v_add_f32 %0, %0, %8\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_add_f32 %2, %2, %8\n s_or_b64 s[26:27], s[24:25], s[22:23]\n v_add_f32 %4, %4, %8\n s_and_b64 s[24:25], s[20:21], s[26:27]\n v_add_f32 %6, %6, %8\n s_or_b64 s[26:27], s[24:25], s[22:23]")
        if (K == 66) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n s_nop 1\n v_cndmask_b32 %1, %1, %9, s[20:21]\n v_cmp_gt_f32 s[22:23], %2, %8\n s_nop 1\n v_cndmask_b32 %3, %3, %9, s[22:23]\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8")
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}


#define ASM_S "v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %9\n v_cmp_gt_f32 s[24:25], %2, %8\n v_cmp_gt_f32 s[26:27], %3, %9\n v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[22:23], %5, %9\n v_cmp_gt_f32 s[24:25], %6, %8\n v_cmp_gt_f32 s[26:27], %7, %9"
#define ASM_F "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %9"
#define ONE(ASM) asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
template <int NB>
__global__ __launch_bounds__(256) void bench_blocked(float *out, int iters)
{
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        float b0 = 1.0001f, b1 = 0.9999f;
        for (int it = 0; it < iters; it += NB) {   // per outer iteration: NB*8 slow then NB*8 fast  (total per 'iters' = iters*8 slow... keep 64/iter)
#pragma unroll
                for (int j = 0; j < NB * 4; j++) { ONE(ASM_S) }
#pragma unroll
                for (int j = 0; j < NB * 4; j++) { ONE(ASM_F) }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// packed: operands are register pairs
#define BODYP(ASM)                                                                                        \
        for (int it = 0; it < iters; it++) {                                                              \
                REP8(asm volatile(ASM : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1));)      \
        }
template <int K>
__global__ __launch_bounds__(256) void benchp(float *out, int iters)
{
        float2_ p0 = { (float) threadIdx.x, 1 }, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f;
        float2_ q0 = { 1.0001f, 0.9999f }, q1 = { 0.9999f, 1.0001f };
        if (K == 0) BODYP("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %4")
        if (K == 1) BODYP("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %5, %4\n v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %5, %4\n v_pk_fma_f32 %3, %3, %4, %5")
        if (K == 2) BODYP("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n v_pk_mul_f32 %0, %0, %5\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %4")

        if (K == 3) BODYP("v_pk_add_f32 %0, %0, %4 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %5 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %2, %2, %4 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %5 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %0, %0, %5 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %2, %2, %5 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]")
        if (K == 4) BODYP("v_pk_mov_b32 %0, %4, %5\n v_pk_mov_b32 %1, %5, %4\n v_pk_mov_b32 %2, %4, %5\n v_pk_mov_b32 %3, %5, %4\n v_pk_mov_b32 %0, %5, %4\n v_pk_mov_b32 %1, %4, %5\n v_pk_mov_b32 %2, %5, %4\n v_pk_mov_b32 %3, %4, %5")
        out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p1.y + p2.x + p3.y;
}


// fp64 / conversion / integer-multiply / LDS classes (the DXT decoders, dxt_decode.hip): operands are register pairs where the
// instruction wants them.  UG_MB_F64=1
#define BODYD(ASM)                                                                                        \
        for (int it = 0; it < iters; it++) {                                                              \
                REP8(asm volatile(ASM : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(e0), "v"(e1), "v"(j0), "v"(j1));) \
        }
template <int K>
__global__ __launch_bounds__(256) void bench64(float *out, int iters)
{
        __shared__ double lds[2048];
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i;
        __syncthreads();
        double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, e0 = 1.0000001, e1 = 0.9999999;
        int i0 = threadIdx.x * 8, i1 = i0 + 1024, i2 = i0 + 2048, i3 = i0 + 4096, j0 = 12345, j1 = 777;
        if (K == 0) BODYD("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %8")
        if (K == 1) BODYD("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %9\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %9\n v_mul_f64 %0, %0, %9\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %9\n v_mul_f64 %3, %3, %8")
        if (K == 2) BODYD("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %9, %8\n v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %9, %8\n v_fma_f64 %3, %3, %8, %9")
        if (K == 3) BODYD("v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3\n v_cvt_i32_f64 %4, %1\n v_cvt_i32_f64 %5, %0\n v_cvt_i32_f64 %6, %3\n v_cvt_i32_f64 %7, %2")
        if (K == 4) BODYD("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7\n v_cvt_f64_u32 %0, %5\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %7\n v_cvt_f64_u32 %3, %6")
        if (K == 5) BODYD("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
        if (K == 6) BODYD("v_rndne_f64 %0, %0\n v_trunc_f64 %1, %1\n v_rndne_f64 %2, %2\n v_trunc_f64 %3, %3\n v_rndne_f64 %0, %0\n v_floor_f64 %1, %1\n v_rndne_f64 %2, %2\n v_floor_f64 %3, %3")
        if (K == 7) BODYD("v_mul_lo_u32 %4, %4, %10\n v_mul_lo_u32 %5, %5, %11\n v_mul_lo_u32 %6, %6, %10\n v_mul_lo_u32 %7, %7, %11\n v_mul_hi_u32 %4, %4, %10\n v_mul_hi_u32 %5, %5, %11\n v_mul_hi_u32 %6, %6, %10\n v_mul_hi_u32 %7, %7, %11")
        if (K == 8) BODYD("v_med3_i32 %4, %4, %10, %11\n v_med3_i32 %5, %5, %11, %10\n v_med3_i32 %6, %6, %10, %11\n v_med3_i32 %7, %7, %11, %10\n v_min3_u32 %4, %4, %10, %11\n v_min3_u32 %5, %5, %11, %10\n v_min_u32 %6, %6, %10\n v_max_i32 %7, %7, %11")
        if (K == 9) BODYD("ds_read_b32 %4, %4\n ds_read_b32 %5, %5\n ds_read_b32 %6, %6\n ds_read_b32 %7, %7\n s_waitcnt lgkmcnt(0)\n v_and_b32 %4, 0x3ff8, %4\n v_and_b32 %5, 0x3ff8, %5\n v_and_b32 %6, 0x3ff8, %6\n v_and_b32 %7, 0x3ff8, %7")
        if (K == 10) BODYD("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %4, 8, %4\n v_xor_b32 %5, 8, %5\n v_xor_b32 %6, 8, %6\n v_xor_b32 %7, 8, %7")
        if (K == 11) BODYD("v_cmp_gt_f64 vcc, %0, %8\n v_cmp_gt_f64 vcc, %1, %9\n v_cmp_gt_f64 vcc, %2, %8\n v_cmp_gt_f64 vcc, %3, %9\n v_cmp_gt_f64 vcc, %0, %9\n v_cmp_gt_f64 vcc, %1, %8\n v_cmp_gt_f64 vcc, %2, %9\n v_cmp_gt_f64 vcc, %3, %8")
        if (K == 12) BODYD("v_mad_u64_u32 %0, vcc, %4, %10, %0\n v_mad_u64_u32 %1, vcc, %5, %11, %1\n v_mad_u64_u32 %2, vcc, %6, %10, %2\n v_mad_u64_u32 %3, vcc, %7, %11, %3\n v_mad_u64_u32 %0, vcc, %5, %10, %0\n v_mad_u64_u32 %1, vcc, %4, %11, %1\n v_mad_u64_u32 %2, vcc, %7, %10, %2\n v_mad_u64_u32 %3, vcc, %6, %11, %3")
        if (K == 13) BODYD("v_cvt_pk_u8_f32 %4, %10, 1, %4\n v_cvt_pk_u8_f32 %5, %11, 2, %5\n v_cvt_pk_u8_f32 %6, %10, 3, %6\n v_cvt_pk_u8_f32 %7, %11, 0, %7\n v_cvt_pk_u8_f32 %4, %10, 0, %4\n v_cvt_pk_u8_f32 %5, %11, 1, %5\n v_cvt_pk_u8_f32 %6, %10, 2, %6\n v_cvt_pk_u8_f32 %7, %11, 3, %7")
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float) (d0 + d1 + d2 + d3) + (float) (i0 + i1 + i2 + i3);
}

static int g_bpc = 8;
static int g_scale = 1;   // kernels launched with iters / g_scale
template <class F>
static void run(const char *name, F launch)
{
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        launch(10);
        hipDeviceSynchronize();
        const int iters = 20000;
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = 256.0 * g_bpc * 4;           // blocks * waves per block
        const double instr = waves * (iters / g_scale) * 64.0;         // 8 reps x 8 instructions
        const double per_s = instr / (ms * 1e-3);
        printf("%-28s %8.3f ms  %7.2f G wave-instr/s  = %5.3f wave-instr/clk/CU @2.4GHz (%.1f T lane-ops/s)\n", name, ms, per_s / 1e9,
               per_s / 256 / 2.4e9, per_s * 64 / 1e12);
}

int main()
{
        float *out;
        hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
        const int bpc = getenv("UG_MB_BPC") ? atoi(getenv("UG_MB_BPC")) : 8;   // 256-thread blocks per CU = waves per SIMD
        g_bpc = bpc;
        const dim3 g(256 * bpc), b(256);
#define R(name, K) run(name, [&](int it) { hipLaunchKernelGGL(bench<K>, g, b, 0, 0, out, it); })
#define RP(name, K) run(name, [&](int it) { hipLaunchKernelGGL(benchp<K>, g, b, 0, 0, out, it); })
        if (getenv("UG_MB_ALL")) {
        R("v_add_f32", 0); R("v_mul_f32", 8); R("v_fma_f32", 1); R("v_min3/max3_f32", 2); R("v_cmp_gt_f32 (sgpr dst)", 3);
        R("v_cvt_f32_ubyteN", 5); R("v_addc_co_u32", 6); R("v_lshl_or_b32", 7); R("v_cmp+v_cndmask (vcc dep)", 9);
        R("v_min/max_f32 (2-op)", 10); R("v_sub_f32", 11); R("v_cmp_gt_f32 e32 (vcc)", 12); R("v_cmp(sgpr)+v_cndmask(sgpr)", 13); R("v_and/or/xor_b32", 14); R("v_mov_b32", 15); R("v_perm_b32", 16); R("v_med3_f32", 17); R("v_bfe_u32", 18); R("v_cvt_u32_f32", 19); R("v_fmac_f32 (VOP2)", 20); R("v_fract/rndne/trunc/floor", 21); R("v_add_f32_dpp row_shr", 22); R("v_add + v_cmp (1:1)", 24); R("v_min3/max3 + v_add (1:1)", 25);
        RP("v_pk_add_f32", 0); RP("v_pk_add_f32 op_sel bcast", 3); RP("v_pk_mov_b32", 4); RP("v_pk_mul_f32", 2); RP("v_pk_fma_f32", 1);
        }
        R("v_mul_f32 literal", 60); R("v_mul_f32 sgpr operand", 63); R("v_fmaak_f32 literal", 61); R("v_add_f32_e64 clamp", 62); R("v_cmp + SALU 1:1 (instr=both)", 64); R("v_add + SALU 1:1 (instr=both)", 65); R("cmp,nop,cndmask x2 + 2 add", 66);
        R("v_add_f32 (ref)", 0); R("v_cmp_gt_f32 (ref)", 3);
        if (getenv("UG_MB_INT")) { R("v_lshr/lshl/ashr_b32", 30); R("v_add/sub_u32", 31); R("v_alignbit_b32", 32); R("v_bfi_b32", 33); R("v_and_or_b32", 34); R("v_bcnt_u32_b32", 35);
        R("v_mul/mad_u32_u24", 36); R("v_sub_f32 + v_alignbit (dep)", 37); R("6 fast : 2 cmp", 38); R("4 cmp : 4 fast (blocked)", 39); R("v_lshl_add/add3/xad_u32", 40); R("v_cvt_f32_u32/i32", 41); R("v_or_b32_sdwa byte", 42); }
#define R64(name, K) run(name, [&](int it) { hipLaunchKernelGGL(bench64<K>, g, b, 0, 0, out, it / 4); })
        if (getenv("UG_MB_F64")) { // iters / 4: these are slow; the printed rate assumes `iters`, so multiply the printed rate by 1 (run() scales by its own iters) -- see below
                g_scale = 4;
                R64("v_add_f64", 0); R64("v_mul_f64", 1); R64("v_fma_f64", 2); R64("v_cvt_i32_f64", 3); R64("v_cvt_f64_i32/u32", 4); R64("v_rcp_f64", 5);
                R64("v_rndne/trunc/floor_f64", 6); R64("v_mul_lo/hi_u32", 7); R64("v_med3_i32/min3_u32/min/max", 8); R64("v_cmp_gt_f64", 11); R64("v_mad_u64_u32", 12);
                R64("v_cvt_pk_u8_f32", 13);
                R64("ds_read_b32 x4 + 4 valu (instr = 8)", 9); R64("ds_read_b64 x4 + 4 valu (instr = 8)", 10);
                g_scale = 1;
        }
        return 0;
}
