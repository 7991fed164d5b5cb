// jpeg_fdct_device.h -- the per-lane 8x8 forward DCT + quantiser of the JPEG path (one lane owns one block: 64 fp32 registers in, 32 packed
// int16 pairs out, zig-zag order), shared by the stand-alone front-end kernels (jpeg_fdct.hip) and the fused encoder kernel
// (jpeg_entropy.hip).  Specified by oracle/jpeg_oracle.c (level shift -- folded into the row pass's DC term here, see fdct8x8 --, AAN float FDCT rows-then-columns, fp32 reciprocal quantiser with the
// AAN scale folded in, rintf, zig-zag); same operation order, no FMA contraction (-ffp-contract=off).  In UltraGrid this stage is inside
// gpujpeg_encoder_encode() (src/video_compress/gpujpeg.cpp:624, external libgpujpeg).
#ifndef UG_JPEG_FDCT_DEVICE_H
#define UG_JPEG_FDCT_DEVICE_H

#include "ug_common.h"

namespace ug_jpeg {

__device__ __forceinline__ void aan_1d(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5, float &d6, float &d7)
{
        constexpr float c4 = 0.707106781f, c6 = 0.382683433f, c2mc6 = 0.541196100f, c2pc6 = 1.306562965f;
        const float t0 = d0 + d7, t7 = d0 - d7;
        const float t1 = d1 + d6, t6 = d1 - d6;
        const float t2 = d2 + d5, t5 = d2 - d5;
        const float t3 = d3 + d4, t4 = d3 - d4;
        // even part
        float t10 = t0 + t3;
        const float t13 = t0 - t3;
        float t11 = t1 + t2;
        float t12 = t1 - t2;
        d0 = t10 + t11;
        d4 = t10 - t11;
        const float z1 = (t12 + t13) * c4;
        d2 = t13 + z1;
        d6 = t13 - z1;
        // odd part
        t10 = t4 + t5;
        t11 = t5 + t6;
        t12 = t6 + t7;
        const float z5 = (t10 - t12) * c6;
        const float z2 = c2mc6 * t10 + z5;
        const float z4 = c2pc6 * t12 + z5;
        const float z3 = t11 * c4;
        const float z11 = t7 + z3, z13 = t7 - z3;
        d5 = z13 + z2;
        d3 = z13 - z2;
        d1 = z11 + z4;
        d7 = z11 - z4;
}

// b = the samples AS THEY ARE (0..255).  The level shift of 128 (T.81 A.3.1; oracle/jpeg_oracle.c subtracts it from every sample) is applied
// behind the row pass, as 8 subtractions per block instead of 64: the one row output it reaches is d0, the sum of the row's eight samples
// (every other output is built from differences of samples, or of sums of equally many), and d0 and every value on the way to it are integers
// below 2^24 -- exact in fp32 with or without the shift.  d0 - 1024 here and the d0 of shifted samples are the same float: same bits out.
__device__ __forceinline__ void fdct8x8(float (&b)[64])
{
#pragma unroll
        for (int r = 0; r < 8; r++) {
                aan_1d(b[8 * r], b[8 * r + 1], b[8 * r + 2], b[8 * r + 3], b[8 * r + 4], b[8 * r + 5], b[8 * r + 6], b[8 * r + 7]);
                b[8 * r] -= 1024.0f;
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
                aan_1d(b[c], b[8 + c], b[16 + c], b[24 + c], b[32 + c], b[40 + c], b[48 + c], b[56 + c]);
        }
}

// (a + b + 1) >> 1 on the four bytes of a word at once (v_lerp_u8): the vertical chroma average of uyvy_to_i420 (to_planar.c:364-367)
__device__ __forceinline__ uint32_t avg_bytes(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }

// One row of a chroma block from UYVY, for a wave whose lanes L (Cb) and L + 32 (Cr) work on the same MCU: each of the two loads HALF of the
// MCU's 32-byte row piece (lane L the first 16 bytes = pixel pairs 0-3, lane L + 32 the second = pairs 4-7; `w` = those four words, for 4:2:0
// already averaged with the line below by avg_bytes) and they trade what the other needs -- U bytes against V bytes -- with ONE
// v_permlane32_swap: packed U of all lanes in `u`, packed V in `v`; the swap exchanges u's upper half with v's lower half, after which u
// holds samples 0-3 and v samples 4-7 of the lane's own component in both halves of the wave.  4 v_perm + 1 swap + 8 v_cvt_f32_ubyte per row,
// and half the loads (and registers in flight) of every lane fetching all 32 bytes to use a quarter of them.
__device__ __forceinline__ void chroma_row_from_uyvy(const uint32_t (&w)[4], float *q8)
{
        const uint32_t t0 = __builtin_amdgcn_perm(w[1], w[0], 0x06020400u); // U0 U1 V0 V1
        const uint32_t t1 = __builtin_amdgcn_perm(w[3], w[2], 0x06020400u); // U2 U3 V2 V3
        const uint32_t u = __builtin_amdgcn_perm(t1, t0, 0x05040100u), v = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, v, false, false); // [0]: u with its lanes 32-63 <- v's lanes 0-31; [1]: v with its lanes 0-31 <- u's lanes 32-63
        const uint32_t lo = sw[0], hi = sw[1];
#pragma unroll
        for (int x = 0; x < 4; x++) {
                q8[x] = (float) ((lo >> (8 * x)) & 0xff);
                q8[4 + x] = (float) ((hi >> (8 * x)) & 0xff);
        }
}

// zig-zag scan (T.81 Figure A.6): kZig[k] = natural index of the k-th coefficient
__device__ constexpr uint8_t kZig[64] = {
        0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

// Quantise + zig-zag one block into 32 packed words (int16 pairs).
// rintf(x) for |x| < 2^22 is computed as the low bits of fl(x + 1.5*2^23): in [2^23, 2^24) the fp32 ulp is 1, so the
// addition itself performs the round-to-nearest-even, and the low 16 mantissa bits are the two's-complement int16.
// Two v_add_f32 + one v_perm_b32 per coefficient pair instead of 2x(v_rndne, v_cvt) + pack; bit-identical to
// (int16_t) rintf(coef * div) of oracle/jpeg_oracle.c.
__device__ __forceinline__ void quant_pack(const float (&b)[64], const float *__restrict__ div, uint32_t (&w)[32])
{
        constexpr float kMagic = 12582912.0f; // 1.5 * 2^23
#pragma unroll
        for (int k = 0; k < 64; k += 2) {
                const int i0 = kZig[k], i1 = kZig[k + 1];
                const float q0 = b[i0] * div[i0] + kMagic;
                const float q1 = b[i1] * div[i1] + kMagic;
                // bytes {q1.b1, q1.b0, q0.b1, q0.b0}
                w[k / 2] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x05040100u);
        }
}

// row pitch (bytes) of a block staged in LDS: 128 B of coefficients + 16: conflict-free 128-bit accesses of consecutive lanes
constexpr int kLdsPitch = 144;

} // namespace ug_jpeg
#endif
