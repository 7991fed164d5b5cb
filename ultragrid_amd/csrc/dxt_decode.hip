// dxt_decode.hip -- DXT1 / DXT1_YUV / DXT5-YCoCg block decode to RGB / BGR / RGBA / UYVY on gfx950.
//
// Receiver-side counterpart of dxt_encode.hip.  The reference decodes with OpenGL
// (src/video_decompress/dxt_glsl.c:142-189 -> dxt_compress/dxt_decoder.c: fixed-function S3TC fetch +
// display_dxt5ycocg_fp.glsl [+ rgba_to_yuv422.glsl]); the only CPU statement of the bitstream semantics is
// the stand-alone tool cuda_dxt/dxt62tga.c:24-106, which this file follows operation for operation in
// fp64 (-ffp-contract=off): it is bit-identical to oracle/dxt_decode_oracle.c, and that oracle is pinned to
// the compiled dxt62tga binary (tests/test_oracle_dxt.py).
//
// Mapping: one lane = one 4x4 block, a wave = 64 consecutive blocks of a block row: the 16-byte (DXT5) /
// 8-byte (DXT1) loads and the four row stores (64 x 16 B RGBA, 64 x 12 B RGB, 64 x 8 B UYVY) are contiguous
// across lanes.  Algorithmic bytes per pixel = 1 (DXT5) or 0.5 (DXT1) read + 2 (UYVY) / 3 (RGB) / 4 (RGBA) written.
//
// DXT5-YCoCg, round 3: the 11 fp64 operations + 3 conversions per pixel of dxt62tga.c made the kernel VALU-issue bound
// (667 VALU instructions per wave of 64 blocks, every fp64 / conversion / integer-multiply instruction ~4.4 cycles per SIMD:
// profiles/r03_dxt_decode_pmc.txt, r03_valu_microbench_f64.txt).  The tool's result for a pixel is clamp(trunc(t)), t = a fp64 value
// within 1e-12 of the rational  255 a_k + 255 (Co_c -+ Cg_c) + 0.5  (a_k one of 8 luma entries, (Co_c, Cg_c) one of 4 palette entries):
// a cheaper association changes t in the last bits only, which reaches the output only where t lies that close to an integer --
// "too rare to matter, impossible to exclude" (DESIGN.md r02).  It CAN be excluded at run time: the kernel computes
// T = A_k + D_c in 32-bit fixed point (2^-20 units; |T / 2^20 - t| < 3.5 units by construction, see dxt5_tables_fixed), takes
// floor(T / 2^20) wherever T lies at least 4 units away from an integer boundary, and re-decodes the blocks where some T
// does not (about 8 in 2^20 values) with the tool's own fp64 statements (decode_block_exact) -- bit-identical output by
// construction, ~1/2 of the instructions.  The same scheme turns rgba_to_yuv422.glsl's fp32 arithmetic for UYVY output into
// 32-bit multiply-adds with the float path as the per-pair fallback.
#include "ug_common.h"

namespace {

__device__ __forceinline__ uint32_t clamp8(double s)
{
        // dxt62tga.c:14-21: (int)(s + 0.5), then clamp to [0, 255]
        const int is = (int) (s + 0.5);
        return (uint32_t) (is > 255 ? 255 : (is < 0 ? 0 : is));
}

// float -> unorm8 framebuffer write of the receiver's shaders.  GL rounds to nearest and leaves exact .5 ties to the implementation:
// AWAY = false (UG_DXT_TIES_EVEN, default): ties to even, what Mesa llvmpipe does when it executes the reference's rgba_to_yuv422.glsl
// (pinned byte for byte, tests/test_oracle_dxt.py); AWAY = true (UG_DXT_TIES_AWAY): floor(x * 255 + 0.5).
template <bool AWAY>
__device__ __forceinline__ uint8_t unorm8_out(float x)
{
        x = __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); // the clamp of the write (the values here are finite; -0 and +0 end as the same byte): one operation, not two compares and two selects
        return AWAY ? (uint8_t) (int) (x * 255.0f + 0.5f) : (uint8_t) (int) rintf(x * 255.0f);
}

// dxt_compress/rgba_to_yuv422.glsl:27-46 on two 8-bit RGB texels -> one UYVY word.  `unorm` = the 256 values v / 255.0f (the
// texel fetch), computed once per workgroup with the IEEE division and kept in LDS: a table read instead of six divisions per pair.
template <bool AWAY>
__device__ __forceinline__ uint32_t rgb_pair_to_uyvy(uint32_t p1, uint32_t p2, const float *unorm)
{
        float yuv[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++) {
                const uint32_t p = i ? p2 : p1;
                const float r = unorm[p & 0xff], g = unorm[(p >> 8) & 0xff], b = unorm[(p >> 16) & 0xff];
                yuv[i][0] = (float) (1.0 / 16.0) + ((r * 0.2126f + g * 0.7152f) + b * 0.0722f) * 0.8588f;
                yuv[i][1] = 0.5f + ((-r * 0.1145f - g * 0.3854f) + b * 0.5f) * 0.8784f;
                yuv[i][2] = 0.5f + ((r * 0.5f - g * 0.4541f) - b * 0.0458f) * 0.8784f;
        }
        const float U = yuv[0][1] * 0.5f + yuv[1][1] * 0.5f, V = yuv[0][2] * 0.5f + yuv[1][2] * 0.5f;
        return (uint32_t) unorm8_out<AWAY>(U) | (uint32_t) unorm8_out<AWAY>(yuv[0][0]) << 8 | (uint32_t) unorm8_out<AWAY>(V) << 16 |
               (uint32_t) unorm8_out<AWAY>(yuv[1][0]) << 24;
}

// one copy of the shader arithmetic for the rare pairs the fixed-point form hands back (a call inside a divergent branch)
template <bool AWAY>
__device__ __noinline__ uint32_t rgb_pair_to_uyvy_call(uint32_t p1, uint32_t p2, const float *unorm) { return rgb_pair_to_uyvy<AWAY>(p1, p2, unorm); }

// fill the v / 255.0f table (256 lanes of the 64x4 workgroup, one division each); call before any early return
template <int OUT>
__device__ __forceinline__ void fill_unorm(float *unorm)
{
        if (OUT == UG_PF_UYVY) {
                const int t = threadIdx.y * 64 + threadIdx.x;
                unorm[t] = (float) t / 255.0f;
                __syncthreads();
        }
}


// rgba_to_yuv422.glsl:27-46 on two 8-bit RGB texels in 32-bit fixed point (2^-24 of a code value), for the DXT5-YCoCg decoder.
// In exact arithmetic the shader's values are linear in the bytes (the v / 255 of the texel fetch cancels against the * 255 of the write):
//      Y'  = 15.9375 + cm (c1 R + c2 G + c3 B)                               cm = 0.8588f, c1..c3 = 0.2126f, 0.7152f, 0.0722f
//      Cb  = 127.5 + 0.5 cu (0.5 SB - c4 SR - c5 SG),  SR = R0 + R1 ...      cu = 0.8784f, c4, c5 = 0.1145f, 0.3854f
//      Cr  = 127.5 + 0.5 cu (0.5 SR - c6 SG - c7 SB)                         c6, c7 = 0.4541f, 0.0458f
// The shader's fp32 evaluation (rgb_pair_to_uyvy above) stays within 1.1e-4 of these (Y': six roundings of 2^-24 on partial sums
// <= 1, times 0.8588, one on the sum, times 255, one on the product; Cb / Cr: 1.04e-4 incl. the + 0.5f of the AWAY rule); the 24-bit
// coefficients below add <= 0.5 * 255 * 3 (Y') or 0.5 * 510 * 3 (Cb, Cr) units = 2.3e-5 / 4.6e-5.  Guard: kGuardUyvy = 3072 units = 1.83e-4
// on each side of a rounding boundary (x.5): outside it the rounded fixed-point value is the shader's byte whatever the tie rule; a pair
// with a value inside it is converted again by the shader's own operations.  All sums stay in [15.5, 240] * 2^24 < 2^32, unsigned.
constexpr int kGuardUyvy = 3072;
constexpr double kCm = (double) 0.8588f, kCu = (double) 0.8784f, kTwo24 = 16777216.0;
constexpr uint32_t kYr = (uint32_t) (kCm * (double) 0.2126f * kTwo24 + 0.5), kYg = (uint32_t) (kCm * (double) 0.7152f * kTwo24 + 0.5),
                   kYb = (uint32_t) (kCm * (double) 0.0722f * kTwo24 + 0.5), kY0 = (uint32_t) (15.9375 * kTwo24) + (1u << 23) + kGuardUyvy;
constexpr uint32_t kUb = (uint32_t) (0.5 * kCu * 0.5 * kTwo24 + 0.5), kUr = (uint32_t) (0.5 * kCu * (double) 0.1145f * kTwo24 + 0.5),
                   kUg = (uint32_t) (0.5 * kCu * (double) 0.3854f * kTwo24 + 0.5);
constexpr uint32_t kVr = (uint32_t) (0.5 * kCu * 0.5 * kTwo24 + 0.5), kVg = (uint32_t) (0.5 * kCu * (double) 0.4541f * kTwo24 + 0.5),
                   kVb = (uint32_t) (0.5 * kCu * (double) 0.0458f * kTwo24 + 0.5);
constexpr uint32_t kC0 = (uint32_t) (127.5 * kTwo24) + (1u << 23) + kGuardUyvy;

// k * x (+ acc) on 24-bit operands as ONE instruction each, the constant from a scalar register.  Written out: left to itself the compiler turns
// "c - k * x" into a full 32-bit multiply by -k behind an AND that re-establishes the 24 bits (v_and + v_mul_lo_u32 + v_add for what
// v_mad_u32_u24 does) -- 12 instructions for the chroma of a pixel pair instead of 8.
#define UG_MUL24(dst, k, x) asm("v_mul_u32_u24 %0, %1, %2" : "=v"(dst) : "s"(k), "v"(x))
#define UG_MAD24(dst, k, x, acc) asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(dst) : "s"(k), "v"(x), "v"(acc))
// Cb / Cr of a pixel pair from the channel sums (0 .. 510): (kA * sa + kC0) - (kB * sb + kC * sc), all modulo 2^32
__device__ __forceinline__ uint32_t chroma_fixed(uint32_t ka, uint32_t sa, uint32_t kb, uint32_t sb, uint32_t kc, uint32_t sc)
{
        uint32_t neg, pos;
        UG_MUL24(neg, kb, sb);
        UG_MAD24(neg, kc, sc, neg);
        UG_MAD24(pos, ka, sa, kC0);
        return pos - neg;
}

// returns the UYVY word as the fixed-point values round; `near` = the smallest distance (in 2^-24 units, biased by the guard) of the four
// values from a rounding boundary: < 2 * kGuardUyvy means the word must not be trusted
__device__ __forceinline__ uint32_t uyvy_pair_fixed(uint32_t r0, uint32_t g0, uint32_t b0, uint32_t r1, uint32_t g1, uint32_t b1, uint32_t &near)
{
        const uint32_t y0 = __umul24(kYr, r0) + (__umul24(kYg, g0) + (__umul24(kYb, b0) + kY0));
        const uint32_t y1 = __umul24(kYr, r1) + (__umul24(kYg, g1) + (__umul24(kYb, b1) + kY0));
        const uint32_t sr = r0 + r1, sg = g0 + g1, sb = b0 + b1;
        const uint32_t u = chroma_fixed(kUb, sb, kUr, sr, kUg, sg);
        const uint32_t v = chroma_fixed(kVr, sr, kVg, sg, kVb, sb);
        const uint32_t m = 0xFFFFFFu;
        near = min(min(y0 & m, y1 & m), min(u & m, v & m));
        // the integer parts are the top bytes: U | Y0 << 8 | V << 16 | Y1 << 24
        return __builtin_amdgcn_perm(y0, u, 0x0c0c0703u) | __builtin_amdgcn_perm(y1, v, 0x07030c0cu);
}

// The same on PACKED bytes (what V_ASHR_PK_U8_I32 leaves): d0 = R0 | G0 << 8 | B0 << 16 | R1 << 24, d1 = G1 | B1 << 8 | (anything) << 16.
// SDWA operand selects read the bytes in place: a 24-bit multiply or an add takes its 8-bit operand straight out of the packed word.
#define UG_MUL24_BYTE(dst, k, packed, n) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #n : "=v"(dst) : "v"(k), "v"(packed))
#define UG_ADD_BYTES(dst, a, i, b, j) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #i " src1_sel:BYTE_" #j : "=v"(dst) : "v"(a), "v"(b))
__device__ __forceinline__ uint32_t uyvy_pair_fixed_packed(uint32_t d0, uint32_t d1, uint32_t &near)
{
        const uint32_t cyr = kYr, cyg = kYg, cyb = kYb; // in registers: SDWA takes no literal
        uint32_t a0, a1, a2, b0, b1, b2, sr, sg, sb;
        UG_MUL24_BYTE(a0, cyr, d0, 0); UG_MUL24_BYTE(a1, cyg, d0, 1); UG_MUL24_BYTE(a2, cyb, d0, 2);
        UG_MUL24_BYTE(b0, cyr, d0, 3); UG_MUL24_BYTE(b1, cyg, d1, 0); UG_MUL24_BYTE(b2, cyb, d1, 1);
        UG_ADD_BYTES(sr, d0, 0, d0, 3); UG_ADD_BYTES(sg, d0, 1, d1, 0); UG_ADD_BYTES(sb, d0, 2, d1, 1);
        const uint32_t y0 = (a0 + a1) + (a2 + kY0), y1 = (b0 + b1) + (b2 + kY0);
        const uint32_t u = chroma_fixed(kUb, sb, kUr, sr, kUg, sg);
        const uint32_t v = chroma_fixed(kVr, sr, kVg, sg, kVb, sb);
        const uint32_t m = 0xFFFFFFu;
        near = min(min(y0 & m, y1 & m), min(u & m, v & m));
        return __builtin_amdgcn_perm(y0, u, 0x0c0c0703u) | __builtin_amdgcn_perm(y1, v, 0x07030c0cu);
}

// x / C for the constant divisors of the decoder (255, 31, 63, 7, 5, 3), correctly rounded without the generic IEEE division
// sequence (v_div_scale x2, v_rcp, 4-5 fma, v_div_fmas, v_div_fixup): q = RN(x * RN(1/C)), one exact residual r = fma(-q, C, x),
// one correction RN(q + r * RN(1/C)).  For these divisors and the (finite, small) sets of numerators the decoder produces the
// result is bit-identical to x / C; that is verified exhaustively on the device by ug_hip_selftest_dxt_decode()
// (tests/test_gpu_dxt_decode.py), not assumed.
template <int C>
__device__ __forceinline__ double div_const(double x)
{
        constexpr double rc = 1.0 / (double) C;
        const double q = x * rc;
        const double r = __builtin_fma(-q, (double) C, x);
        return __builtin_fma(r, rc, q);
}

struct OutArgs {
        static constexpr bool kEdge = false;
        uint8_t *dst;
        long pitch;
        int rs, gs, bs;
};
// Pictures whose width or height is not a multiple of 4: the stream holds (w+3)/4 x (h+3)/4 blocks, the picture shown is w x h
// (dxt_decoder.c:146-149,368-389; dxt_util.h:59-67).  They run the EDGE instantiations of the kernels below, whose row stores drop the
// lines and pixels past the picture and take whatever alignment a line of such a width has; multiples of 4 run the code they always ran.
struct OutArgsE : OutArgs {
        static constexpr bool kEdge = true;
        int w, h;
};
template <bool EDGE> struct ArgsOf { typedef OutArgs type; };
template <> struct ArgsOf<true> { typedef OutArgsE type; };
typedef uint32_t u32_any __attribute__((aligned(1)));

// EDGE only: one row of block column bx as the bytes it occupies in memory -- 4 words RGBA, 3 words RGB / BGR, 2 words UYVY
__device__ __forceinline__ void put_row_edge(int words, const OutArgsE &o, int y, int bx, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
        if (y >= o.h) return; // (scalar) block rows end below the picture
        const int valid = o.w - 4 * bx; // pixel columns of this block inside the picture; `words` = bytes per pixel
        uint8_t *at = o.dst + (long) y * o.pitch + bx * (4 * words);
        const uint32_t w[4] = { w0, w1, w2, w3 };
        if (valid >= 4) {
                // (scalar choice) lines that keep the alignment of the wide streaming stores -- a width that IS a multiple of 4 with a height that is not,
                // 2048 x 858 -- take them as the other instantiation does; else dwords wherever they lie (3 * width bytes per line: no alignment at all)
                if (words == 4 && (o.pitch & 15) == 0) {
                        ug::st_stream((uint4 *) at, make_uint4(w0, w1, w2, w3));
                } else if (words == 2 && (o.pitch & 7) == 0) {
                        ug::st_stream((uint2 *) at, make_uint2(w0, w1));
                } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                                if (k < words) *((u32_any *) at + k) = w[k];
                        }
                }
        } else { // the block the right edge cuts: one lane per block row
#pragma unroll
                for (int i = 0; i < 12; i++) {
                        if (i < valid * words) at[i] = (uint8_t) (w[i / 4] >> (8 * (i % 4)));
                }
        }
}

// one pixel in the form the row store wants it: RGBA = the final 32-bit pixel; RGB = R | G << 8 | B << 16; BGR = B | G << 8 | R << 16;
// UYVY = R | G << 8 | B << 16 (the shader input)
template <int OUT>
__device__ __forceinline__ uint32_t pack_px(const OutArgs &o, uint32_t R, uint32_t G, uint32_t B)
{
        if (OUT == UG_PF_RGBA) {
                const uint32_t am = 0xFFFFFFFFu ^ (0xFFu << o.rs) ^ (0xFFu << o.gs) ^ (0xFFu << o.bs);
                return ((am | R << o.rs) | G << o.gs) | B << o.bs;
        }
        return OUT == UG_PF_BGR ? (B | G << 8 | R << 16) : (R | G << 8 | B << 16);
}

// store one decoded row (4 pixels as pack_px made them) of block column bx
template <int OUT, bool AWAY, class A>
__device__ __forceinline__ void store_words(const A &o, int y, int bx, const uint32_t (&p)[4], const float *unorm)
{
        if constexpr (A::kEdge) {
                if (OUT == UG_PF_RGBA) {
                        put_row_edge(4, o, y, bx, p[0], p[1], p[2], p[3]);
                } else if (OUT == UG_PF_RGB || OUT == UG_PF_BGR) {
                        put_row_edge(3, o, y, bx, p[0] | p[1] << 24, (p[1] >> 8) | p[2] << 16, (p[2] >> 16) | p[3] << 8, 0);
                } else {
                        put_row_edge(2, o, y, bx, rgb_pair_to_uyvy<AWAY>(p[0], p[1], unorm), rgb_pair_to_uyvy<AWAY>(p[2], p[3], unorm), 0, 0);
                }
                return;
        }
        uint8_t *row = o.dst + (long) y * o.pitch;
        if (OUT == UG_PF_RGBA) {
                ug::st_stream((uint4 *) row + bx, make_uint4(p[0], p[1], p[2], p[3]));
        } else if (OUT == UG_PF_RGB || OUT == UG_PF_BGR) {
                uint32_t *d = (uint32_t *) row + 3 * bx;
                ug::st_stream(d, p[0] | p[1] << 24);
                ug::st_stream(d + 1, (p[1] >> 8) | p[2] << 16);
                ug::st_stream(d + 2, (p[2] >> 16) | p[3] << 8);
        } else { // UYVY
                ug::st_stream((uint2 *) row + bx, make_uint2(rgb_pair_to_uyvy<AWAY>(p[0], p[1], unorm), rgb_pair_to_uyvy<AWAY>(p[2], p[3], unorm)));
        }
}

// the same from 4 pixels packed R | G<<8 | B<<16 (the DXT1 palettes)
template <int OUT, bool AWAY, class A>
__device__ __forceinline__ void store_row(const A &o, int y, int bx, const uint32_t (&px)[4], const float *unorm)
{
        uint32_t p[4];
#pragma unroll
        for (int i = 0; i < 4; i++) p[i] = (OUT == UG_PF_RGB || OUT == UG_PF_UYVY) ? px[i] : pack_px<OUT>(o, px[i] & 0xff, (px[i] >> 8) & 0xff, px[i] >> 16);
        store_words<OUT, AWAY>(o, y, bx, p, unorm);
}

// ---- DXT5-YCoCg ------------------------------------------------------------------------------------------------------------------
// dxt62tga.c:24-106 for one block, the tool's fp64 statements one for one (the reference semantics, and the fallback of the
// fixed-point path below).  Tables in registers, selected with compare trees: this runs for a few blocks per thousand only.
__device__ __forceinline__ double sel8(const double (&t)[8], int i)
{
        const double lo = (i & 2) ? ((i & 1) ? t[3] : t[2]) : ((i & 1) ? t[1] : t[0]);
        const double hi = (i & 2) ? ((i & 1) ? t[7] : t[6]) : ((i & 1) ? t[5] : t[4]);
        return (i & 4) ? hi : lo;
}

// `pal` = this lane's column of a [4][64] table of (Co, Cg) pairs in LDS
template <int OUT, bool AWAY, class A>
__device__ __noinline__ void decode_block_exact(uint4 q, A o, int bx, int by, double2 *pal, const float *unorm) // (o by value: a reference would put the kernel's copy into scratch memory)
{
        unsigned long long ac = (unsigned long long) q.x | (unsigned long long) q.y << 32;
        unsigned long long cc = (unsigned long long) q.z | (unsigned long long) q.w << 32;
        // dxt62tga.c:60-62 alpha endpoints, :36-58 the two interpolation modes
        double ta[8];
        const double a0 = div_const<255>((double) (ac & 0xFF)), a1 = div_const<255>((double) ((ac >> 8) & 0xFF));
        ta[0] = a0;
        ta[1] = a1;
        if (a0 > a1) {
#pragma unroll
                for (int k = 2; k < 8; k++) ta[k] = div_const<7>((double) (8 - k) * a0 + (double) (k - 1) * a1);
        } else {
#pragma unroll
                for (int k = 2; k < 6; k++) ta[k] = div_const<5>((double) (6 - k) * a0 + (double) (k - 1) * a1);
                ta[6] = 0.0;
                ta[7] = 1.0;
        }
        // dxt62tga.c:63-74 colour endpoints and the two thirds; :24-27 per-entry scale / Co / Cg
        {
                double r[4], g[4], b[4];
                b[0] = div_const<31>((double) (cc & 0x1F));         g[0] = div_const<63>((double) ((cc >> 5) & 0x3F));  r[0] = div_const<31>((double) ((cc >> 11) & 0x1F));
                b[1] = div_const<31>((double) ((cc >> 16) & 0x1F)); g[1] = div_const<63>((double) ((cc >> 21) & 0x3F)); r[1] = div_const<31>((double) ((cc >> 27) & 0x1F));
                b[2] = div_const<3>(2.0 * b[0] + b[1]); g[2] = div_const<3>(2.0 * g[0] + g[1]); r[2] = div_const<3>(2.0 * r[0] + r[1]);
                b[3] = div_const<3>(b[0] + 2.0 * b[1]); g[3] = div_const<3>(g[0] + 2.0 * g[1]); r[3] = div_const<3>(r[0] + 2.0 * r[1]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        const double scale = 1.0 / (31.875 * b[k] + 1.0);
                        pal[64 * k] = make_double2((r[k] - 5.01960814E-01) * scale, (g[k] - 5.01960814E-01) * scale);
                }
        }
        ac >>= 16;
        cc >>= 32;
        for (int y = 0; y < 4; y++) { // (not unrolled: code size, this path is rare)
                uint32_t p[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                        const int ai = (int) (ac & 7), ci = (int) (cc & 3);
                        ac >>= 3;
                        cc >>= 2;
                        const double a = sel8(ta, ai);
                        const double2 cocg = pal[64 * ci];
                        const double Co = cocg.x, Cg = cocg.y;
                        const uint32_t R = clamp8(((a + Co) - Cg) * 255.0);
                        const uint32_t G = clamp8((a + Cg) * 255.0);
                        const uint32_t B = clamp8(((a - Co) - Cg) * 255.0);
                        p[x] = pack_px<OUT>(o, R, G, B);
                }
                store_words<OUT, AWAY>(o, 4 * by + y, bx, p, unorm);
        }
}

// The fixed-point tables of one block.  Units: 2^-20 of an 8-bit step ("ulp" below).
//   A_k (8 luma entries)  ~ 2^20 (255 a_k + 0.5) + kGuard.  255 a_k is the rational (w0 a0 + w1 a1) / 7 (or / 5) of the two endpoint
//                         BYTES, so A_k = (a0 << 20) + w1 * delta with delta = RN((a1 - a0) 2^20 / 7): error <= 6 * 0.5 = 3 ulp.
//   D_c, E_c, F_c (4 palette entries) ~ 2^20 * 255 * (Co - Cg), Cg, (-Co - Cg).  With nb, nr in [0, 93], ng in [0, 189] the integers
//                         3 x endpoint or 2 x one endpoint + the other:  b = nb / 93, r = nr / 93, g = ng / 189,
//                         scale = 1 / (31.875 b + 1) = 744 / (255 nb + 744), so each is RN(fp64 product): error <= 0.5 ulp + 1e-8.
// dxt62tga.c's own fp64 value t (before the truncation) differs from the rational it approximates by < 1e-12 (five roundings of
// quantities below 512).  Hence |A_k + X_c - kGuard - 2^20 t| < 3.5 + 1e-6: wherever (A_k + X_c) mod 2^20 >= 2 kGuard = 8 the integer
// part of (A_k + X_c) / 2^20 IS trunc(t) for t >= 0 and <= 0 for t < 0 (clamped to 0 either way); elsewhere the block is re-decoded exactly.
constexpr int kFix = 20, kGuard = 4;
constexpr int kOff = (1 << (kFix - 1)) + kGuard;

__device__ __forceinline__ int rn_i32(double x) { return (int) __builtin_rint(x); }

// gfx950's V_ASHR_PK_U8_I32: D.b[0] = sat_u8(S0 >> S2), D.b[1] = sat_u8(S1 >> S2) -- shift, clamp to [0, 255] and pack, for two values, in
// one instruction.  It writes ONE HALF of the destination (the low one, or the high one with op_sel:[0,0,0,1]) and leaves the other as it
// was (tools/ashr_pk_probe.hip, profiles/r03_ashr_pk_probe.txt).  ROCm 7.2's compiler also forms it from min(max(x >> s, 0), 255) pairs
// and then takes bits 31:16 for zero, which they are not: that source pattern is avoided in this file (clamp_shift below is opaque to
// the compiler), and the instruction is issued by hand: two of them make a finished 4-byte pixel.
static_assert(kFix == 20, "the shift is spelled out in the instruction strings below");
__device__ __forceinline__ uint32_t pk2_u8(int b0, int b1, int b2, int b3) // sat_u8(b_i >> 20) in byte i
{
        uint32_t d;
        asm("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "=v"(d) : "v"(b0), "v"(b1));
        asm("v_ashr_pk_u8_i32 %0, %1, %2, 20 op_sel:[0,0,0,1]" : "+v"(d) : "v"(b2), "v"(b3));
        return d;
}
__device__ __forceinline__ uint32_t clamp_shift(int t) // min(max(t >> 20, 0), 255), not recognisable as the pattern above
{
        int r;
        asm("v_ashrrev_i32 %0, 20, %1\n\tv_med3_i32 %0, %0, 0, %2" : "=&v"(r) : "v"(t), "v"(255));
        return (uint32_t) r;
}

// one block: fixed-point tables -> 16 pixels -> rows stored; guarded blocks decoded again exactly
template <int OUT, bool AWAY, int MODE, class ARGS>
__device__ __forceinline__ void dxt5_decode_one(const uint4 q, const ARGS &o, int bx, int by, int (*lds_a)[64], const double *lds_rs, const float *unorm, int lane,
                                                unsigned *flagged)
{
        bool redo = MODE == 1;
        if (MODE != 1) {
                // ---- luma table ----
                const int a0 = (int) (q.x & 0xFF), a1 = (int) ((q.x >> 8) & 0xFF);
                const bool m7 = a0 > a1; // dxt62tga.c:36: a0 > a1 <=> the bytes compare so (x / 255 is monotone)
                const int delta = rn_i32((double) (a1 - a0) * (m7 ? 1048576.0 / 7.0 : 1048576.0 / 5.0));
                int e = (a0 << kFix) + kOff;
                lds_a[0][lane] = e;
                lds_a[1][lane] = (a1 << kFix) + kOff;
#pragma unroll
                for (int k = 2; k < 6; k++) {
                        e += delta;
                        lds_a[k][lane] = e;
                }
                e += delta;
                lds_a[6][lane] = m7 ? e : kOff;                         // 5-interpolant mode: entry 6 = 0.0, entry 7 = 1.0 (:55-56)
                lds_a[7][lane] = m7 ? e + delta : (255 << kFix) + kOff;
                // ---- palette table: column j = the channel that goes to output byte j (RGBA: by the shifts; BGR: B, G, R; else R, G, B) ----
                const int b0 = (int) (q.z & 0x1F), g0 = (int) ((q.z >> 5) & 0x3F), r0 = (int) ((q.z >> 11) & 0x1F);
                const int b1 = (int) ((q.z >> 16) & 0x1F), g1 = (int) ((q.z >> 21) & 0x3F), r1 = (int) (q.z >> 27);
                const int nb[4] = { 3 * b0, 3 * b1, 2 * b0 + b1, b0 + 2 * b1 }, ng[4] = { 3 * g0, 3 * g1, 2 * g0 + g1, g0 + 2 * g1 },
                          nr[4] = { 3 * r0, 3 * r1, 2 * r0 + r1, r0 + 2 * r1 };
                const int col_r = OUT == UG_PF_RGBA ? o.rs >> 3 : (OUT == UG_PF_BGR ? 2 : 0), col_b = OUT == UG_PF_RGBA ? o.bs >> 3 : (OUT == UG_PF_BGR ? 0 : 2);
                const int col_g = OUT == UG_PF_RGBA ? o.gs >> 3 : 1; // (wave-uniform: the three stores below go through scalar address arithmetic)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        const double s = lds_rs[nb[k]];
                        const double u = __builtin_fma((double) nr[k], 1.0 / 93.0, -5.01960814E-01);  // r - c
                        const double v = __builtin_fma((double) ng[k], 1.0 / 189.0, -5.01960814E-01); // g - c
                        lds_a[8 + 4 * col_r + k][lane] = rn_i32((u - v) * s);
                        lds_a[8 + 4 * col_g + k][lane] = rn_i32(v * s);
                        lds_a[8 + 4 * col_b + k][lane] = rn_i32(-(u + v) * s);
                }
                // ---- pixels ---- (LDS traffic is lane-private: no barrier, a wave's own ds ops are ordered)
                const uint32_t aidx[2] = { (q.x >> 16) | (q.y << 16), q.y >> 8 }; // alpha indices of pixels 0-7 (24 bits) / 8-15
                const uint32_t mask = (1u << kFix) - 1u;
                const char *const colbase = (const char *) &lds_a[0][lane];
                const int opaque = 0x7FF00000; // >> 20 saturates to 255: the alpha byte
                uint32_t near = 0xFFFFFFFFu;
#pragma unroll
                for (int y = 0; y < 4; y++) {
                        int t0[4], t1[4], t2[4];
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                                const int i = 4 * y + x;
                                // row = 64 lanes x 4 B = 256 B: address = this lane's column + (index << 8); one v_bfe_u32 + one v_lshl_add_u32 per index
                                // (the empty asm keeps the compiler from folding the field extraction into shift + and + add: three instructions)
                                uint32_t ai = __builtin_amdgcn_ubfe(aidx[i >> 3], 3 * (i & 7), 3), ci = __builtin_amdgcn_ubfe(q.w, 2 * i, 2);
                                asm("" : "+v"(ai), "+v"(ci));
                                const int A = *(const int *) (colbase + (ai << 8));
                                const char *pc = colbase + 8 * 256 + (ci << 8);
                                t0[x] = A + *(const int *) pc;
                                t1[x] = A + *(const int *) (pc + 4 * 256);
                                t2[x] = A + *(const int *) (pc + 8 * 256);
                                near = min(near, min(min((uint32_t) t0[x] & mask, (uint32_t) t1[x] & mask), (uint32_t) t2[x] & mask));
                        }
                        uint8_t *row = o.dst + (long) (4 * by + y) * o.pitch;
                        if (OUT == UG_PF_RGBA) {
                                const uint4 v4 = make_uint4(pk2_u8(t0[0], t1[0], t2[0], opaque), pk2_u8(t0[1], t1[1], t2[1], opaque),
                                                            pk2_u8(t0[2], t1[2], t2[2], opaque), pk2_u8(t0[3], t1[3], t2[3], opaque));
                                if constexpr (ARGS::kEdge) put_row_edge(4, o, 4 * by + y, bx, v4.x, v4.y, v4.z, v4.w);
                                else ug::st_stream((uint4 *) row + bx, v4);
                        } else if (OUT == UG_PF_RGB || OUT == UG_PF_BGR) {
                                if constexpr (ARGS::kEdge) {
                                        put_row_edge(3, o, 4 * by + y, bx, pk2_u8(t0[0], t1[0], t2[0], t0[1]), pk2_u8(t1[1], t2[1], t0[2], t1[2]),
                                                     pk2_u8(t2[2], t0[3], t1[3], t2[3]), 0);
                                } else {
                                        uint32_t *d = (uint32_t *) row + 3 * bx;
                                        ug::st_stream(d, pk2_u8(t0[0], t1[0], t2[0], t0[1]));
                                        ug::st_stream(d + 1, pk2_u8(t1[1], t2[1], t0[2], t1[2]));
                                        ug::st_stream(d + 2, pk2_u8(t2[2], t0[3], t1[3], t2[3]));
                                }
                        } else { // UYVY: rgba_to_yuv422.glsl in fixed point; a pair with a value near a rounding boundary goes through the shader's own fp32 operations
                                uint32_t w2[2];
#pragma unroll
                                for (int k = 0; k < 2; k++) {
#ifndef UG_DXT5_UYVY_UNPACKED
                                        const uint32_t d0 = pk2_u8(t0[2 * k], t1[2 * k], t2[2 * k], t0[2 * k + 1]); // R0 G0 B0 R1
                                        uint32_t d1, un;                                                            // G1 B1 -  -
                                        asm("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "=v"(d1) : "v"(t1[2 * k + 1]), "v"(t2[2 * k + 1]));
                                        w2[k] = uyvy_pair_fixed_packed(d0, d1, un);
                                        if (un < 2u * kGuardUyvy) w2[k] = rgb_pair_to_uyvy_call<AWAY>(d0 & 0xFFFFFFu, d0 >> 24 | (d1 & 0xFFFFu) << 8, unorm);
#else // A/B variant: clamp each sample on its own, multiply-adds on whole registers (12 % more instructions per block)
                                        const uint32_t ra = clamp_shift(t0[2 * k]), ga = clamp_shift(t1[2 * k]), ba = clamp_shift(t2[2 * k]);
                                        const uint32_t rb = clamp_shift(t0[2 * k + 1]), gb = clamp_shift(t1[2 * k + 1]), bb = clamp_shift(t2[2 * k + 1]);
                                        uint32_t un;
                                        w2[k] = uyvy_pair_fixed(ra, ga, ba, rb, gb, bb, un);
                                        if (un < 2u * kGuardUyvy) w2[k] = rgb_pair_to_uyvy_call<AWAY>(ra | ga << 8 | ba << 16, rb | gb << 8 | bb << 16, unorm);
#endif
                                }
                                if constexpr (ARGS::kEdge) put_row_edge(2, o, 4 * by + y, bx, w2[0], w2[1], 0, 0);
                                else ug::st_stream((uint2 *) row + bx, make_uint2(w2[0], w2[1]));
                        }
                }
                redo = MODE == 0 && near < 2u * kGuard;
                if (flagged != nullptr && near < 2u * kGuard) atomicAdd(flagged, 1u);
        }
        // the guarded blocks again, exactly (their rows are stored a second time, behind the first store of the same lane); the palette
        // of the exact path overlays this lane's column of the fixed-point one, which is not read any more
        if (redo) decode_block_exact<OUT, AWAY>(q, o, bx, by, (double2 *) &lds_a[0][0] + lane, unorm);
}

template <int OUT, bool AWAY, int MODE, bool EDGE> // MODE 0: fixed point + guard + exact fallback (the product); 1: exact only; 2: fixed point only (tests)
__global__ __launch_bounds__(256) void dxt5ycocg_decode_kernel(const uint4 *__restrict__ src, typename ArgsOf<EDGE>::type o, int bw, int bh, unsigned *__restrict__ flagged)
{
        // per-lane tables, one region per wave, entry-major inside it so that a wave's accesses to one entry are contiguous (conflict-free
        // whatever the indices): rows 0-7 the luma entries, rows 8 + 4 j + k = palette entry k of output byte j.  The exact path overlays
        // the region of ITS wave with 4 x 64 (Co, Cg) pairs of doubles (4 KiB of the 5 KiB) once the wave has read its fixed-point tables.
        __shared__ __attribute__((aligned(16))) int lds_tab[4][20][64];
        __shared__ double lds_rs[96];
        __shared__ float unorm[OUT == UG_PF_UYVY ? 256 : 1];
        const int t = threadIdx.y * 64 + threadIdx.x, lane = threadIdx.x;
        int (*const lds_a)[64] = lds_tab[threadIdx.y];
        if (OUT == UG_PF_UYVY) unorm[t] = (float) t / 255.0f;
        if (MODE != 1 && t < 94) lds_rs[t] = (1048576.0 * 255.0 * 744.0) / (255.0 * (double) t + 744.0); // 2^20 * 255 * scale(nb = t)
        if (OUT == UG_PF_UYVY || MODE != 1) __syncthreads();
        // (one block row per wave: a loop over several rows with the next row's blocks in flight was measured -- 9.3-9.8 us per 4K frame
        // against 9.0 -- and dropped, profiles/r03_dxt_decode_after.txt)
        const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
        if (bx >= bw || by >= bh) return;
        dxt5_decode_one<OUT, AWAY, MODE>(src[(long) by * bw + bx], o, bx, by, lds_a, lds_rs, unorm, lane, flagged);
}

// YUV = true: DXT1_YUV -- the palette holds Y, Cb, Cr and goes through the display matrix of
// dxt_compress/display_dxt1_yuv_fp.glsl:21-32 (fp32, one operation per shader operation) before the 8-bit write.
template <int OUT, bool YUV, bool AWAY, bool EDGE>
__global__ __launch_bounds__(256) void dxt1_decode_kernel(const uint2 *__restrict__ src, typename ArgsOf<EDGE>::type o, int bw, int bh)
{
        __shared__ float unorm[OUT == UG_PF_UYVY ? 256 : 1];
        fill_unorm<OUT>(unorm);
        const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
        if (bx >= bw || by >= bh) return;
        const uint2 q = src[(long) by * bw + bx];
        const uint32_t c0 = q.x & 0xffff, c1 = q.x >> 16;
        double p[4][3];
        p[0][0] = div_const<31>((double) ((c0 >> 11) & 0x1F)); p[0][1] = div_const<63>((double) ((c0 >> 5) & 0x3F)); p[0][2] = div_const<31>((double) (c0 & 0x1F));
        p[1][0] = div_const<31>((double) ((c1 >> 11) & 0x1F)); p[1][1] = div_const<63>((double) ((c1 >> 5) & 0x3F)); p[1][2] = div_const<31>((double) (c1 & 0x1F));
#pragma unroll
        for (int k = 0; k < 3; k++) {
                if (c0 > c1) {
                        p[2][k] = div_const<3>(2.0 * p[0][k] + p[1][k]);
                        p[3][k] = div_const<3>(p[0][k] + 2.0 * p[1][k]);
                } else { // 3-colour + transparent-black mode (never produced by our encoder)
                        p[2][k] = (p[0][k] + p[1][k]) / 2.0;
                        p[3][k] = 0.0;
                }
        }
        uint32_t pal[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
                if (YUV) {
                        const float col0 = (float) p[k][0], col1 = (float) p[k][1], col2 = (float) p[k][2];
                        const float Y = 1.1643f * (col0 - 0.0625f), U = 1.1384f * (col1 - 0.5f), V = 1.1384f * (col2 - 0.5f);
                        const float G = (Y - 0.39173f * U) - 0.81290f * V, B = Y + 2.017f * U, R = Y + 1.5958f * V;
                        pal[k] = (uint32_t) unorm8_out<AWAY>(R) | (uint32_t) unorm8_out<AWAY>(G) << 8 | (uint32_t) unorm8_out<AWAY>(B) << 16;
                } else {
                        pal[k] = clamp8(p[k][0] * 255.0) | clamp8(p[k][1] * 255.0) << 8 | clamp8(p[k][2] * 255.0) << 16;
                }
        }
        uint32_t idx = q.y;
        if (OUT == UG_PF_UYVY) {
                // rgba_to_yuv422.glsl on a block with four colours: Y', and the halves of Cb and Cr that the pair average adds up, are functions of
                // the palette entry alone -- computed once per entry with the shader's own operations (rgb_pair_to_uyvy above), looked up per pixel
                // ... through a column of the lane's own in LDS (entry k of lane t at [k][t]: consecutive lanes in consecutive banks whatever
                // their indices; nobody else reads it, so no barrier): one address + one ds_read per look-up, where selecting among four
                // registers by a 2-bit index costs three v_cndmask and their compares -- four look-ups per pixel pair
                __shared__ float col_u[4][256], col_v[4][256];
                const int t = threadIdx.y * 64 + threadIdx.x;
                uint32_t y4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        const float r = unorm[pal[k] & 0xff], g = unorm[(pal[k] >> 8) & 0xff], b = unorm[(pal[k] >> 16) & 0xff];
                        const float yy = (float) (1.0 / 16.0) + ((r * 0.2126f + g * 0.7152f) + b * 0.0722f) * 0.8588f;
                        const float uu = 0.5f + ((-r * 0.1145f - g * 0.3854f) + b * 0.5f) * 0.8784f;
                        const float vv = 0.5f + ((r * 0.5f - g * 0.4541f) - b * 0.0458f) * 0.8784f;
                        y4 |= (uint32_t) unorm8_out<AWAY>(yy) << (8 * k);
                        col_u[k][t] = uu * 0.5f;
                        col_v[k][t] = vv * 0.5f;
                }
#pragma unroll
                for (int y = 0; y < 4; y++) {
                        uint32_t word[2];
#pragma unroll
                        for (int p = 0; p < 2; p++) {
                                const uint32_t ca = idx & 3, cb = (idx >> 2) & 3;
                                idx >>= 4;
                                const float ua = col_u[ca][t], ub = col_u[cb][t], va = col_v[ca][t], vb = col_v[cb][t];
                                word[p] = (uint32_t) unorm8_out<AWAY>(ua + ub) | ((y4 >> (8 * ca)) & 0xff) << 8 | (uint32_t) unorm8_out<AWAY>(va + vb) << 16 |
                                          ((y4 >> (8 * cb)) & 0xff) << 24;
                        }
                        if constexpr (EDGE) put_row_edge(2, o, 4 * by + y, bx, word[0], word[1], 0, 0);
                        else ug::st_stream((uint2 *) (o.dst + (long) (4 * by + y) * o.pitch) + bx, make_uint2(word[0], word[1]));
                }
                return;
        }
#pragma unroll
        for (int y = 0; y < 4; y++) {
                uint32_t px[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                        const uint32_t ci = idx & 3;
                        idx >>= 2;
                        const uint32_t lo = (ci & 1) ? pal[1] : pal[0], hi = (ci & 1) ? pal[3] : pal[2];
                        px[x] = (ci & 2) ? hi : lo;
                }
                store_row<OUT, AWAY>(o, 4 * by + y, bx, px, unorm);
        }
}


// ---- exhaustive check of div_const<> against the IEEE division on every numerator the decoders can produce ----
__device__ __forceinline__ unsigned differs(double a, double b) { return __double_as_longlong(a) != __double_as_longlong(b) ? 1u : 0u; }

__global__ void selftest_div_kernel(unsigned *mismatches)
{
        // one thread per (i, j) in 256 x 256
        const int i = blockIdx.x, j = threadIdx.x;
        unsigned bad = 0;
        volatile double d255 = 255.0, d31 = 31.0, d63 = 63.0, d7 = 7.0, d5 = 5.0, d3 = 3.0; // keep the reference divisions real divisions
        const double a0 = (double) i / d255, a1 = (double) j / d255;
        bad += differs(div_const<255>((double) i), a0);
        for (int k = 2; k < 8; k++) {
                const double n = (double) (8 - k) * a0 + (double) (k - 1) * a1;
                bad += differs(div_const<7>(n), n / d7);
        }
        for (int k = 2; k < 6; k++) {
                const double n = (double) (6 - k) * a0 + (double) (k - 1) * a1;
                bad += differs(div_const<5>(n), n / d5);
        }
        if (i < 64 && j < 64) { // 6-bit (green) endpoints and their thirds
                const double g0 = (double) i / d63, g1 = (double) j / d63;
                bad += differs(div_const<63>((double) i), g0);
                bad += differs(div_const<3>(2.0 * g0 + g1), (2.0 * g0 + g1) / d3);
                bad += differs(div_const<3>(g0 + 2.0 * g1), (g0 + 2.0 * g1) / d3);
        }
        if (i < 32 && j < 32) { // 5-bit (red / blue) endpoints and their thirds
                const double b0 = (double) i / d31, b1 = (double) j / d31;
                bad += differs(div_const<31>((double) i), b0);
                bad += differs(div_const<3>(2.0 * b0 + b1), (2.0 * b0 + b1) / d3);
                bad += differs(div_const<3>(b0 + 2.0 * b1), (b0 + 2.0 * b1) / d3);
        }
        if (bad) atomicAdd(mismatches, bad);
}

// test hook (ug_hip_dxt_decode_debug): which DXT5-YCoCg path runs, and where the guarded blocks are counted
int g_dxt5_mode = 0;
unsigned *g_dxt5_flagged = nullptr;

template <int OUT, bool AWAY, bool EDGE>
int launch_decode_e(ug_dxt_t in, const void *src, const typename ArgsOf<EDGE>::type &o, int w, int h, hipStream_t st)
{
        const int bw = (w + 3) / 4, bh = (h + 3) / 4; // dxt_util.h:59-67
        const dim3 block(64, 4), grid((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4));
        if (in == UG_DXT5_YCOCG) {
                // The fixed-point RGBA path keeps one palette column per output byte 0..2 and writes alpha to byte 3: right for the shifts
                // that are a permutation of {0, 8, 16}.  Any other byte placement (a shift of 24: alpha in front, as vc_copylineRGBA's callers
                // may ask, dxt_glsl.c:178) goes through the exact kernel, whose pack_px places the channels by shifting.
                const bool rgb_low = OUT != UG_PF_RGBA || ((1 << (o.rs >> 3)) | (1 << (o.gs >> 3)) | (1 << (o.bs >> 3))) == 7;
                if (g_dxt5_mode == 1 || !rgb_low) {
                        hipLaunchKernelGGL((dxt5ycocg_decode_kernel<OUT, AWAY, 1, EDGE>), grid, block, 0, st, (const uint4 *) src, o, bw, bh, g_dxt5_flagged);
                } else if (g_dxt5_mode == 2) {
                        hipLaunchKernelGGL((dxt5ycocg_decode_kernel<OUT, AWAY, 2, EDGE>), grid, block, 0, st, (const uint4 *) src, o, bw, bh, g_dxt5_flagged);
                } else {
                        hipLaunchKernelGGL((dxt5ycocg_decode_kernel<OUT, AWAY, 0, EDGE>), grid, block, 0, st, (const uint4 *) src, o, bw, bh, g_dxt5_flagged);
                }
        } else if (in == UG_DXT1_YUV) {
                hipLaunchKernelGGL((dxt1_decode_kernel<OUT, true, AWAY, EDGE>), grid, block, 0, st, (const uint2 *) src, o, bw, bh);
        } else {
                hipLaunchKernelGGL((dxt1_decode_kernel<OUT, false, AWAY, EDGE>), grid, block, 0, st, (const uint2 *) src, o, bw, bh);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

template <int OUT, bool AWAY>
int launch_decode_t(ug_dxt_t in, const void *src, const OutArgs &o, int w, int h, hipStream_t st)
{
        if ((w & 3) || (h & 3)) {
                OutArgsE e;
                (OutArgs &) e = o;
                e.w = w;
                e.h = h;
                return launch_decode_e<OUT, AWAY, true>(in, src, e, w, h, st);
        }
        return launch_decode_e<OUT, AWAY, false>(in, src, o, w, h, st);
}

template <int OUT>
int launch_decode(ug_dxt_t in, int ties, const void *src, const OutArgs &o, int w, int h, hipStream_t st)
{
        // the tie rule only reaches the outputs that pass through a shader's unorm8 write: UYVY, and the DXT1_YUV display matrix
        if (ties == UG_DXT_TIES_AWAY && (OUT == UG_PF_UYVY || in == UG_DXT1_YUV)) {
                return launch_decode_t<OUT, true>(in, src, o, w, h, st);
        }
        return launch_decode_t<OUT, false>(in, src, o, w, h, st);
}

} // namespace

extern "C" int ug_hip_dxt_decode_ex(ug_dxt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev, int width, int height,
                                    int dst_pitch, int rshift, int gshift, int bshift, int ties, ug_hip_stream_t stream)
{
        if (ties != UG_DXT_TIES_EVEN && ties != UG_DXT_TIES_AWAY) {
                ug::set_last_error_msg("ug_hip_dxt_decode: unknown tie rule");
                return UG_HIP_EINVAL;
        }
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_dxt_decode");
        if (!src_dev || !dst_dev || width <= 0 || height <= 0 || (15 & (uintptr_t) dst_dev) ||
            ((in == UG_DXT5_YCOCG ? 15 : 7) & (uintptr_t) src_dev) || ((height + 3) / 4 + 3) / 4 > 65535) {
                ug::set_last_error_msg("ug_hip_dxt_decode: bad size or alignment");
                return UG_HIP_EINVAL;
        }
        // any width and height >= 1 (dxt_decoder.c:146-149: the texture is created width x height over a stream of whole blocks);
        // the 4:2:2 output is made of pixel pairs (rgba_to_yuv422.glsl renders width / 2 texels)
        const bool edge = (width & 3) || (height & 3);
        if (out == UG_PF_UYVY && (width & 1)) {
                ug::set_last_error_msg("ug_hip_dxt_decode: UYVY output needs an even width");
                return UG_HIP_EINVAL;
        }
        if (in != UG_DXT1 && in != UG_DXT1_YUV && in != UG_DXT5_YCOCG) {
                ug::set_last_error_msg("ug_hip_dxt_decode: unknown compressed format");
                return UG_HIP_EUNSUPP;
        }
        if (out == UG_PF_RGBA && (((rshift | gshift | bshift) & 7) || (unsigned) rshift > 24 || (unsigned) gshift > 24 || (unsigned) bshift > 24)) {
                ug::set_last_error_msg("ug_hip_dxt_decode: RGBA component shifts must be 0, 8, 16 or 24");
                return UG_HIP_EINVAL;
        }
        if (dst_pitch == 0) {
                dst_pitch = ug::linesize(out, width);
        }
        if (dst_pitch < 0 || !ug::span_ok(dst_pitch, height)) return ug::refuse_size("ug_hip_dxt_decode");
        OutArgs o = { (uint8_t *) dst_dev, dst_pitch, rshift, gshift, bshift };
        hipStream_t st = (hipStream_t) stream;
        switch (out) {
        case UG_PF_RGBA:
                if (dst_pitch & (edge ? 3 : 15)) break;
                return launch_decode<UG_PF_RGBA>(in, ties, src_dev, o, width, height, st);
        case UG_PF_RGB:
                if ((dst_pitch & 3) && !edge) break;
                return launch_decode<UG_PF_RGB>(in, ties, src_dev, o, width, height, st);
        case UG_PF_BGR:
                if ((dst_pitch & 3) && !edge) break;
                return launch_decode<UG_PF_BGR>(in, ties, src_dev, o, width, height, st);
        case UG_PF_UYVY:
                if (dst_pitch & (edge ? 3 : 7)) break;
                return launch_decode<UG_PF_UYVY>(in, ties, src_dev, o, width, height, st);
        default:
                ug::set_last_error_msg("ug_hip_dxt_decode: unsupported output format");
                return UG_HIP_EUNSUPP;
        }
        ug::set_last_error_msg("ug_hip_dxt_decode: destination pitch not aligned for this output format");
        return UG_HIP_EINVAL;
}

extern "C" int ug_hip_dxt_decode(ug_dxt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev, int width, int height,
                                 int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream)
{
        return ug_hip_dxt_decode_ex(in, out, src_dev, dst_dev, width, height, dst_pitch, rshift, gshift, bshift, UG_DXT_TIES_DEFAULT, stream);
}

// Test hook, not part of the product API's contract: selects the DXT5-YCoCg decode path of the calls that follow (0 = product: fixed point
// + guard + exact fallback; 1 = the exact fp64 statements only; 2 = fixed point only, no fallback) and, with a device counter, counts the
// blocks whose fixed-point values came within the guard band of an integer.  Process-wide; tests restore mode 0 / nullptr.
extern "C" int ug_hip_dxt_decode_debug(int mode, unsigned *flagged_blocks_dev)
{
        if (mode < 0 || mode > 2) return UG_HIP_EINVAL;
        g_dxt5_mode = mode;
        g_dxt5_flagged = flagged_blocks_dev;
        return UG_HIP_SUCCESS;
}

// Runs the exhaustive comparison of the decoders' constant-divisor quotients with the IEEE division; *mismatches must come back 0.
extern "C" int ug_hip_selftest_dxt_decode(unsigned *mismatches, ug_hip_stream_t stream)
{
        if (!mismatches) return UG_HIP_EINVAL;
        unsigned *dev = nullptr;
        UG_HIP_TRY(hipMalloc((void **) &dev, sizeof *dev));
        hipStream_t st = (hipStream_t) stream;
        hipError_t err = hipMemsetAsync(dev, 0, sizeof *dev, st);
        if (err == hipSuccess) {
                hipLaunchKernelGGL(selftest_div_kernel, dim3(256), dim3(256), 0, st, dev);
                err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipMemcpyAsync(mismatches, dev, sizeof *dev, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        (void) hipFree(dev);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_selftest_dxt_decode");
                return UG_HIP_ERUNTIME;
        }
        return UG_HIP_SUCCESS;
}
