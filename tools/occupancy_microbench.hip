// occupancy_microbench.hip -- VALU issue rate of a slow:fast instruction mix as a function of waves per SIMD (gfx950).
// Occupancy is limited with dynamic LDS (256-thread groups = one wave per SIMD; k groups per CU = k waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o occupancy_microbench occupancy_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define BODY(ASM)                                                                                         \
        for (int it = 0; it < iters; it++) {                                                              \
                REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                                   : "v"(b0), "v"(b1) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)  \
        }

template <int K>
__global__ __launch_bounds__(256) void bench(float *out, int iters)
{
        extern __shared__ float lds[];
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        float b0 = 1.0001f, b1 = 0.9999f;
        // K=0: 4 slow (cmp->sgpr) : 4 fast interleaved; K=1: all fast; K=2: all slow; K=3: 5 slow : 3 fast; K=4: dependent pairs
        if (K == 0) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_add_f32 %1, %1, %9\n v_cmp_gt_f32 s[22:23], %2, %8\n v_mul_f32 %3, %3, %9\n v_cmp_gt_f32 s[24:25], %4, %8\n v_sub_f32 %5, %5, %9\n v_cmp_gt_f32 s[26:27], %6, %8\n v_add_f32 %7, %7, %9")
        if (K == 1) BODY("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n v_add_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9")
        if (K == 2) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_min3_f32 %1, %1, %9, %8\n v_cmp_gt_f32 s[22:23], %2, %8\n v_max3_f32 %3, %3, %9, %8\n v_cmp_gt_f32 s[24:25], %4, %8\n v_cvt_f32_ubyte0 %5, %5\n v_cmp_gt_f32 s[26:27], %6, %8\n v_bfe_u32 %7, %7, 8, 8")
        if (K == 3) BODY("v_cmp_gt_f32 s[20:21], %0, %8\n v_add_f32 %1, %1, %9\n v_cmp_gt_f32 s[22:23], %2, %8\n v_min3_f32 %3, %3, %9, %8\n v_cmp_gt_f32 s[24:25], %4, %8\n v_sub_f32 %5, %5, %9\n v_addc_co_u32 %6, s[26:27], %6, %6, s[20:21]\n v_mul_f32 %7, %7, %9")
        // dependent mix like the encoder's: sub -> mul -> add chain per accumulator pair, compare, mask logic on SALU, addc
        if (K == 4) BODY("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_add_f32 %2, %0, %1\n v_cmp_gt_f32 s[20:21], %2, %3\n v_cmp_gt_f32 s[22:23], %2, %4\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_addc_co_u32 %5, s[26:27], %5, %5, s[24:25]")
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[0];
}

template <int K>
static void run(const char *name, int valu_per_body, float *out)
{
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        printf("%s\n", name);
        for (int wps = 1; wps <= 8; wps++) {
                const size_t lds = (160 * 1024 / wps) - 512; // wps groups of 4 waves fit per CU
                hipFuncSetAttribute((const void *) bench<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
                const int groups = 256 * wps * 4; // 4 rounds of full occupancy
                bench<K><<<groups, 256, lds>>>(out, 10);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                bench<K><<<groups, 256, lds>>>(out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double winstr = (double) groups * 4 * iters * 8.0 * valu_per_body;
                printf("  %d waves/SIMD: %8.3f ms  %7.1f G wave-instr/s  (%.3f /clk/CU @2.4GHz)\n", wps, ms, winstr / ms / 1e6, winstr / (ms * 1e-3) / 256 / 2.4e9);
        }
}

int main()
{
        float *out; hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(float));
        run<1>("all fast (add/mul/sub)", 8, out);
        run<2>("all slow (cmp/min3/cvt/pk)", 8, out);
        run<0>("4 slow : 4 fast interleaved", 8, out);
        run<3>("5 slow : 3 fast (cmp, min3, addc)", 8, out);
        run<4>("dependent encoder-like chain (8 VALU + 1 SALU)", 8, out);
        return 0;
}
