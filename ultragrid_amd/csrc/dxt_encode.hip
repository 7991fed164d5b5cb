// dxt_encode.hip -- fused pixel-format unpack + colour conversion + DXT1 / DXT5-YCoCg 4x4
// block encode for gfx950 (CDNA4).
//
// Replaces, in one pass and with no intermediate buffer:
//   - the CPU decoder_t line loop of the compress modules (cuda_dxt.cpp:206-220,
//     dxt_glsl.cpp:277-289): v210 -> UYVY (pixfmt_conv.c:86-130) is folded into the load;
//   - cuda_yuv422_to_yuv444 / yuv422_to_yuv444.glsl (chroma replication, a full-frame pass
//     in the reference: cuda_dxt.cu:697-732, dxt_encoder.c:482-542);
//   - the block encoders dxt_kernel<...> (cuda_dxt.cu:622-695) / fp_compress_dxt5ycocg
//     (compress_dxt5ycocg_fp.glsl:326-377) / fp_compress_dxt1 (compress_dxt1_fp.glsl:177-229).
//
// Numerics contract: bit-identical to oracle/dxt_oracle.c, the strict-fp32 restatement of
// the shaders.  Every source-level operation is one IEEE binary32 operation; this file is
// compiled with -ffp-contract=off.  Where an operation is fused or strength-reduced below,
// the comment states why the result is bit-identical (exact products by powers of two).
//
// Mapping: one lane encodes one 4x4 block (v210: three consecutive blocks = one 32-byte
// group pair per row).  Consecutive lanes take consecutive blocks of a block-row, so a
// wave's row loads are contiguous (64 x 8 B UYVY, 64 x 12 B RGB, 64 x 32 B v210) and its
// 64 x 16 B (DXT5) / 64 x 8 B (DXT1) stores form one contiguous 1 KiB / 512 B burst.
// The work is VALU-bound: no LDS staging is needed because each input
// byte is read by exactly one lane.  LDS only holds each lane's own lookup columns of the fast
// index stages (IndexTables below): the threshold / palette pair a pixel's one open comparison needs.
//
// Frame sizes that are not multiples of 4 (dxt_util.h:59-67, dxt_encoder.c:362-364,380-394; what -c RTDXT accepts): the stream holds
// (w+3)/4 x (h+3)/4 blocks and pixels past the picture repeat its last column / last line (oracle/dxt_oracle.c, encode_rows, says how
// that is pinned to the executed shaders and where the reference itself slips).  Such frames run the EDGE instantiation of the same
// kernel: line numbers are clamped on the scalar unit, the one lane per block row that holds the cut block fills its registers
// through Loader::load_edge (nothing past the last byte of a line is read), and lines may start at any alignment the packed width
// gives them.  Frames whose sizes ARE multiples of 4 never see any of it: their instantiation is the code it was before.
#include "ug_common.h"

#ifndef UG_DXT_TBUF
#define UG_DXT_TBUF 1 // typed buffer loads do the byte -> float conversion (0: v_cvt_f32_ubyte in the shader; A/B switch)
#endif

namespace {

constexpr float kInv255  = 0.00392156862745f;           // cuda_dxt.cu:666
constexpr float kOffset  = (float) (128.0 / 255.0);     // compress_dxt5ycocg_fp.glsl:25
constexpr float kInsetC  = (float) ((8.0 / 255.0) / 16.0);
constexpr float kInsetY  = (float) ((16.0 / 255.0) / 32.0);

__device__ __forceinline__ float clamp01(float v) { return fminf(1.0f, fmaxf(0.0f, v)); }

// GLSL round() leaves the direction of exact .5 ties to the implementation, and so does the order in which dot(vec3) is summed.
// Both choices are a RUN-TIME option of the library (UG_DXT_TIES_*, include/ug_mi355x.h), compiled as a template parameter:
//   TIES_EVEN (default): round() ties to even, dot(vec3) summed from the last component -- what the reference's shaders compute when
//                        they are EXECUTED (Mesa llvmpipe, oracle/glsl_ref.c; tests/golden/dxt_glsl_ref.npz pins every block);
//   TIES_AWAY          : roundf() half away from zero, dot() left to right -- the reference's CUDA text (cuda_dxt.cu:106-108,122-124).
// (uint32_t) rintf(x) is v_rndne_f32 + v_cvt_u32_f32.  (uint32_t) roundf(x) for 0 <= x < 2^22: for x >= 0.5 the fp32 sum x + 0.5f
// truncates to the right integer: it is exact unless it lands in a higher binade than x, which only happens for
// x in [2^k - 0.5, 2^k), where both roundf(x) and the (rounded) sum's integer part are 2^k.  Below 0.5 the sum can
// round up to 1.0 (x = 0.5 - 2^-25), so that range is selected to 0 explicitly.  4 instructions instead of roundf's 7.
template <bool AWAY>
__device__ __forceinline__ uint32_t round_u32(float x)
{
        if (!AWAY) {
                return (uint32_t) rintf(x);
        }
        const uint32_t r = (uint32_t) (x + 0.5f);
        return x < 0.5f ? 0u : r;
}

// x / 14.0f without the IEEE division sequence (v_div_scale x2, v_rcp, v_div_fmas, v_div_fixup + 4 fma: five of them on the slow
// pipe, the reciprocal at quarter rate): q = RN(x * RN(1/14)), exact residual, one correction.  Bit-identical to the division for
// x = 0 and EVERY fp32 x in [2^-100, 1], checked exhaustively on the device (ug_hip_selftest_dxt_encode); below 2^-125 the
// quotient is denormal and the residual no longer exact, so callers route (0, 2^-100) to the real division.
constexpr uint32_t kDiv14Exact = 0x0d800000u; // bits of 2^-100
__device__ __forceinline__ float div14(float x)
{
        constexpr float rc = 1.0f / 14.0f;
        const float q = x * rc;
        const float r = __builtin_fmaf(-q, 14.0f, x);
        return __builtin_fmaf(r, rc, q);
}

// GLSL mix(a,b,q) = a*(1-q) + b*q, w = 1-q precomputed in fp32 (cuda_dxt.cu:126-128)
__device__ __forceinline__ float lerp_w(float a, float b, float w, float q)
{
        float p0 = a * w;
        float p1 = b * q;
        return p0 + p1;
}

__device__ __forceinline__ uint32_t palette_index(float d0, float d1, float d2, float d3)
{
        // compress_dxt5ycocg_fp.glsl:237-244
        uint32_t b0 = d0 > d3, b1 = d1 > d2, b2 = d0 > d2, b3 = d1 > d3, b4 = d2 > d3;
        return (b0 & b4) | (((b1 & b2) | (b0 & b3)) << 1);
}


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_any __attribute__((aligned(1))); // a dword wherever it lies (lines of 3 * width bytes, EDGE instantiations only)

// acc = 2*acc + bit in ONE VALU instruction: the boolean lives in an SGPR lane mask (it is the
// result of v_cmp / SALU mask logic) and is consumed as the carry-in of v_addc_co_u32.
typedef unsigned long long lanemask_t;
// lane mask of a comparison: the SGPR pair v_cmp writes; mask logic on these runs on the scalar unit
#define LANEMASK(cmp) __builtin_amdgcn_ballot_w64(cmp)
__device__ __forceinline__ uint32_t shift_in(uint32_t acc, lanemask_t mask)
{
        lanemask_t carry_out;
        uint32_t r;
        asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(acc), "s"(mask));
        return r;
}

// acc + carry (v_addc with a zero addend): adds the boolean of a lane mask to a count
__device__ __forceinline__ uint32_t add_mask(uint32_t acc, lanemask_t mask)
{
        lanemask_t carry_out;
        uint32_t r;
        asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(acc), "s"(mask));
        return r;
}
// v_cvt_u32_f32 as the hardware defines it (truncation, negative -> 0, no poison for out-of-range input as with a C cast)
__device__ __forceinline__ uint32_t cvt_u32_sat(float x)
{
        uint32_t r;
        asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
        return r;
}
// bit i of x -> bit 2i (x < 2^16)
__device__ __forceinline__ uint32_t spread16(uint32_t x)
{
        x = (x | (x << 8)) & 0x00ff00ffu;
        x = (x | (x << 4)) & 0x0f0f0f0fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        return x;
}

// Per-wave LDS tables of the fast index stages (UG_DXT_FAST_INDEX): every lane owns one column, nothing is shared
// between lanes, so no barrier is involved.  Row-major by entry: a wave's access to one row is 64 consecutive words / 16-byte units.
#ifndef UG_DXT_NO_FAST_INDEX
#define UG_DXT_FAST_INDEX 1
#else
#define UG_DXT_FAST_INDEX 0
#endif
// pixels per group of the fast stages' software pipelines (table reads of group n + 1 in flight while group n is evaluated).
// Measured, interleaved A/B on UYVY->DXT5 4K: colour groups of 2 / 4 and luma groups of 4 / 8 are equal within 0.3 %; 8 / 16 spill.
#ifndef UG_DXT1_FAST_GROUP
#define UG_DXT1_FAST_GROUP 2
#endif
#ifndef UG_DXT5_FAST_GROUP
#define UG_DXT5_FAST_GROUP 4
#endif
#ifndef UG_DXT5_ALPHA_GROUP
#define UG_DXT5_ALPHA_GROUP 8
#endif
// Diagnostics (ug_hip_dxt_encode_stats): waves that left a fast index stage for the reference's full form, counted in those (cold) paths only
__device__ unsigned long long g_full_form_waves[2]; // [0] colour indices, [1] alpha indices
__device__ __forceinline__ void count_full_form(int which)
{
        if ((int) threadIdx.x == __builtin_amdgcn_readfirstlane((int) threadIdx.x)) {
                atomicAdd(&g_full_form_waves[which], 1ull);
        }
}
struct IndexTables {
        float *alpha;     // [8][64] floats: thresholds in DESCENDING order, row 7 = -inf
        float4 *colour;   // DXT5-YCoCg: [3][64] (A.x, A.y, B.x, B.y) of the palette pair whose bisector crosses zone k; DXT1: [6][64], rows 2k / 2k + 1 = A / B (xyz)
};

// ---------------------------------------------------------------------------------------
// colour front ends: bytes -> normalised (c0,c1,c2) per pixel
// ---------------------------------------------------------------------------------------
struct Px16 {
        float a[16], b[16], c[16];
};

// ConvertYUVToRGB (compress_dxt5ycocg_fp.glsl:12-23) for a pixel pair sharing chroma.
__device__ __forceinline__ void yuv_pair_to_rgb(float y0, float y1, float u, float v, Px16 &p, int i)
{
        const float U = u - 0.5f, V = v - 0.5f;
        const float rv = 1.7926f * V, gu = 0.2132f * U, gv = 0.5328f * V, bu = 2.1124f * U;
        const float Y0 = 1.1643f * (y0 - 0.0625f), Y1 = 1.1643f * (y1 - 0.0625f);
        p.a[i] = Y0 + rv;
        p.b[i] = (Y0 - gu) - gv;
        p.c[i] = Y0 + bu;
        p.a[i + 1] = Y1 + rv;
        p.b[i + 1] = (Y1 - gu) - gv;
        p.c[i + 1] = Y1 + bu;
}

__device__ __forceinline__ float byte_f(uint32_t w, int k) { return (float) ((w >> (8 * k)) & 0xffu) * kInv255; }

// 4 pixels of one row from three RGB words (12 B)
__device__ __forceinline__ void row_rgb(const uint32_t *w, Px16 &p, int i)
{
        p.a[i + 0] = byte_f(w[0], 0); p.b[i + 0] = byte_f(w[0], 1); p.c[i + 0] = byte_f(w[0], 2);
        p.a[i + 1] = byte_f(w[0], 3); p.b[i + 1] = byte_f(w[1], 0); p.c[i + 1] = byte_f(w[1], 1);
        p.a[i + 2] = byte_f(w[1], 2); p.b[i + 2] = byte_f(w[1], 3); p.c[i + 2] = byte_f(w[2], 0);
        p.a[i + 3] = byte_f(w[2], 1); p.b[i + 3] = byte_f(w[2], 2); p.c[i + 3] = byte_f(w[2], 3);
}

template <int IN>
struct Loader;

// ---- RGB / YUV444: 12 B per block row ----
template <bool YUV>
struct Loader3 {
        static constexpr int kBlocks = 1;
        uint32_t w[4][3];
        template <bool EDGE = false>
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        if (EDGE) { // 3 * width bytes per line: a line starts wherever it starts
                                const u32_any *p = (const u32_any *) (src + ((uint32_t) rows[r] * pitch + unit_x * 12u));
                                w[r][0] = p[0]; w[r][1] = p[1]; w[r][2] = p[2];
                        } else {
                                const uint32_t *p = (const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 12u));
                                w[r][0] = p[0]; w[r][1] = p[1]; w[r][2] = p[2];
                        }
                }
        }
        // The block at the right edge of a picture whose width is not a multiple of 4: `valid` (1..3) of its pixel columns exist, the
        // others repeat the last one (GL_CLAMP_TO_EDGE, dxt_encoder.c:362-364).  Nothing past the last byte of a line is read.
        __device__ __forceinline__ void load_edge(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4], int valid)
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint8_t *p = src + ((uint32_t) rows[r] * pitch + unit_x * 12u);
                        uint32_t b[12];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                                const int jj = 3 * min(j, valid - 1);
                                b[3 * j] = p[jj]; b[3 * j + 1] = p[jj + 1]; b[3 * j + 2] = p[jj + 2];
                        }
#pragma unroll
                        for (int k = 0; k < 3; k++) w[r][k] = b[4 * k] | b[4 * k + 1] << 8 | b[4 * k + 2] << 16 | b[4 * k + 3] << 24;
                }
        }
        __device__ __forceinline__ void block(int, Px16 &p) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        row_rgb(w[r], p, 4 * r);
                }
                if (YUV) { // cuda_dxt.cu:686-691 (per pixel, no chroma sharing in 4:4:4)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                                const float U = p.b[i] - 0.5f, V = p.c[i] - 0.5f;
                                const float Y = 1.1643f * (p.a[i] - 0.0625f);
                                p.a[i] = Y + 1.7926f * V;
                                p.b[i] = (Y - 0.2132f * U) - 0.5328f * V;
                                p.c[i] = Y + 2.1124f * U;
                        }
                }
        }
};

// ---- RGBA: 16 B per block row, alpha ignored (compress_dxt1_fp.glsl:41 reads .rgb) ----
struct LoaderRGBAWords {
        static constexpr int kBlocks = 1;
        uint4 w[4];
        template <bool EDGE = false>
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        if (EDGE) { // 4 * width bytes per line: 4-byte alignment is all a line has
                                const uint32_t *q = (const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 16u));
                                w[r] = make_uint4(q[0], q[1], q[2], q[3]);
                        } else {
                                w[r] = *(const uint4 *) (src + ((uint32_t) rows[r] * pitch + unit_x * 16u));
                        }
                }
        }
        __device__ __forceinline__ void load_edge(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4], int valid)
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t *q = (const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 16u));
                        w[r] = make_uint4(q[0], q[min(1, valid - 1)], q[min(2, valid - 1)], q[min(3, valid - 1)]);
                }
        }
        __device__ __forceinline__ void block(int, Px16 &p) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q[4] = { w[r].x, w[r].y, w[r].z, w[r].w };
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                                p.a[4 * r + c] = byte_f(q[c], 0);
                                p.b[4 * r + c] = byte_f(q[c], 1);
                                p.c[4 * r + c] = byte_f(q[c], 2);
                        }
                }
        }
};

// ---- UYVY: 8 B per block row; chroma replicated (yuv422_to_yuv444.glsl:22-30) ----
template <bool CONVERT>
struct LoaderUYVY {
        static constexpr int kBlocks = 1;
        uint2 w[4];
        template <bool EDGE = false>
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        if (EDGE) { // 2 * width bytes per line, width even: 4-byte alignment
                                const uint32_t *q = (const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                                w[r] = make_uint2(q[0], q[1]);
                        } else {
                                w[r] = *(const uint2 *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                        }
                }
        }
        // width = 2 (mod 4): one U Y0 V Y1 group exists; the second one repeats its last pixel: U Y1 V Y1
        __device__ __forceinline__ void load_edge(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4], int)
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q = *(const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                        w[r] = make_uint2(q, (q & 0xffff00ffu) | (q >> 24) << 8);
                }
        }
        __device__ __forceinline__ void block(int, Px16 &p) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q[2] = { w[r].x, w[r].y };
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                                const float u = byte_f(q[k], 0), y0 = byte_f(q[k], 1);
                                const float v = byte_f(q[k], 2), y1 = byte_f(q[k], 3);
                                const int i = 4 * r + 2 * k;
                                if (CONVERT) {
                                        yuv_pair_to_rgb(y0, y1, u, v, p, i);
                                } else { // DXT1_YUV: YCbCr stored in the RGB channels (dxt_encoder.c:318-323)
                                        p.a[i] = y0; p.b[i] = u; p.c[i] = v;
                                        p.a[i + 1] = y1; p.b[i + 1] = u; p.c[i + 1] = v;
                                }
                        }
                }
        }
};
#ifndef UG_DXT_LDS_CONV
#define UG_DXT_LDS_CONV 0 // experiment of round 4: 1 = the byte -> term maps of ConvertYUVToRGB are 256-entry tables in LDS
#endif
#if UG_DXT_LDS_CONV
// Every term of ConvertYUVToRGB is a function of ONE byte: Y' = 1.1643 * (y / 255 - 0.0625), (1.7926, 0.5328) * (v / 255 - 0.5),
// (0.2132, 2.1124) * (u / 255 - 0.5).  The workgroup builds the three tables once, with the very statements of yuv_pair_to_rgb -- the entries
// are bit-identical by construction --, and a pixel pair then costs 4 table reads (the LDS pipe is idle in this kernel) + the 8 additions
// instead of 22 VALU operations.  Raw word loads; the byte -> table offset is one shift with a byte selector.
struct ConvTables {
        float y[256];
        float2 v[256]; // (rv, gv)
        float2 u[256]; // (gu, bu)
};
template <bool CONVERT>
struct LoaderUYVYLds {
        static constexpr int kBlocks = 1;
        static constexpr bool kLdsConv = CONVERT;
        uint2 w[4];
        const ConvTables *lut;
        template <bool EDGE = false>
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        if (EDGE) { // 2 * width bytes per line, width even: 4-byte alignment
                                const uint32_t *q = (const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                                w[r] = make_uint2(q[0], q[1]);
                        } else {
                                w[r] = *(const uint2 *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                        }
                }
        }
        // width = 2 (mod 4): one U Y0 V Y1 group exists; the second one repeats its last pixel: U Y1 V Y1
        __device__ __forceinline__ void load_edge(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4], int)
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q = *(const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                        w[r] = make_uint2(q, (q & 0xffff00ffu) | (q >> 24) << 8);
                }
        }
        __device__ __forceinline__ void block(int, Px16 &p) const
        {
                const char *const ty = (const char *) lut->y, *const tv = (const char *) lut->v, *const tu = (const char *) lut->u;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q[2] = { w[r].x, w[r].y };
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                                const int i = 4 * r + 2 * k;
                                if (CONVERT) {
                                        const uint32_t ou = (q[k] & 0xffu) << 3, oy0 = ((q[k] >> 8) & 0xffu) << 2;
                                        const uint32_t ov = ((q[k] >> 16) & 0xffu) << 3, oy1 = (q[k] >> 24) << 2;
                                        const float2 gb = *(const float2 *) (tu + ou), rg = *(const float2 *) (tv + ov);
                                        const float Y0 = *(const float *) (ty + oy0), Y1 = *(const float *) (ty + oy1);
                                        p.a[i] = Y0 + rg.x;
                                        p.b[i] = (Y0 - gb.x) - rg.y;
                                        p.c[i] = Y0 + gb.y;
                                        p.a[i + 1] = Y1 + rg.x;
                                        p.b[i + 1] = (Y1 - gb.x) - rg.y;
                                        p.c[i + 1] = Y1 + gb.y;
                                } else {
                                        const float u = byte_f(q[k], 0), y0 = byte_f(q[k], 1), v = byte_f(q[k], 2), y1 = byte_f(q[k], 3);
                                        p.a[i] = y0; p.b[i] = u; p.c[i] = v;
                                        p.a[i + 1] = y1; p.b[i + 1] = u; p.c[i + 1] = v;
                                }
                        }
                }
        }
};
template <> struct Loader<UG_PF_UYVY> : LoaderUYVYLds<true> {};
template <> struct Loader<UG_PF_UYVY_RAW> : LoaderUYVYLds<false> {};
#elif UG_DXT_TBUF
// Same, but the texture-address unit does the byte -> float conversion: typed buffer loads with format 8_8_8_8 USCALED return
// float(byte) for the four bytes of a word (exact), so the 32 v_cvt_f32_ubyte of a block -- slow-pipe VALU work -- disappear; the
// multiplication by kInv255 stays in the shader arithmetic.  Costs 24 more VGPRs (the raw block is held as 32 floats).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <bool CONVERT>
struct LoaderUYVYTyped {
        static constexpr int kBlocks = 1;
        f32x4 f[4][2];
        // width = 2 (mod 4): one U Y0 V Y1 group exists; the second one repeats its last pixel: U Y1 V Y1.  (float) byte is what the
        // USCALED typed load returns.
        __device__ __forceinline__ void load_edge(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4], int)
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint32_t q = *(const uint32_t *) (src + ((uint32_t) rows[r] * pitch + unit_x * 8u));
                        const float u = (float) (q & 0xffu), y0 = (float) ((q >> 8) & 0xffu), v = (float) ((q >> 16) & 0xffu), y1 = (float) (q >> 24);
                        f[r][0] = f32x4{ u, y0, v, y1 };
                        f[r][1] = f32x4{ u, y1, v, y1 };
                }
        }
        template <bool EDGE = false> // (the typed loads take 4-byte elements: a line that is only 4-byte aligned is no different)
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
                const uint64_t base = (uint64_t) src;
                // V#: base, stride 0, num_records = max, dst_sel xyzw = R G B A, data format 8_8_8_8, type buffer
                const i32x4 desc = { (int) (uint32_t) base, (int) ((uint32_t) (base >> 32) & 0xffffu), -1, 0x52FAC };
                uint32_t off[4];
#pragma unroll
                for (int r = 0; r < 4; r++) off[r] = (uint32_t) rows[r] * pitch + unit_x * 8u;
                asm volatile("tbuffer_load_format_xyzw %0, %8, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n"
                             "tbuffer_load_format_xyzw %1, %8, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:4\n"
                             "tbuffer_load_format_xyzw %2, %9, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n"
                             "tbuffer_load_format_xyzw %3, %9, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:4\n"
                             "tbuffer_load_format_xyzw %4, %10, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n"
                             "tbuffer_load_format_xyzw %5, %10, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:4\n"
                             "tbuffer_load_format_xyzw %6, %11, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n"
                             "tbuffer_load_format_xyzw %7, %11, %12, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:4\n"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(f[0][0]), "=&v"(f[0][1]), "=&v"(f[1][0]), "=&v"(f[1][1]), "=&v"(f[2][0]), "=&v"(f[2][1]), "=&v"(f[3][0]), "=&v"(f[3][1])
                             : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(desc)
                             : "memory");
        }
        __device__ __forceinline__ void block(int, Px16 &p) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                                const float u = f[r][k].x * kInv255, y0 = f[r][k].y * kInv255;
                                const float v = f[r][k].z * kInv255, y1 = f[r][k].w * kInv255;
                                const int i = 4 * r + 2 * k;
                                if (CONVERT) {
                                        yuv_pair_to_rgb(y0, y1, u, v, p, i);
                                } else {
                                        p.a[i] = y0; p.b[i] = u; p.c[i] = v;
                                        p.a[i + 1] = y1; p.b[i + 1] = u; p.c[i + 1] = v;
                                }
                        }
                }
        }
};
template <> struct Loader<UG_PF_UYVY> : LoaderUYVYTyped<true> {};
template <> struct Loader<UG_PF_UYVY_RAW> : LoaderUYVYTyped<false> {};

#else
template <> struct Loader<UG_PF_UYVY> : LoaderUYVY<true> {};
template <> struct Loader<UG_PF_UYVY_RAW> : LoaderUYVY<false> {};
#endif
// RGB / YUV444 / RGBA keep word loads: measured with typed loads RGBA is equal and RGB (three 4-byte typed loads per 12-byte row) twice
// as slow -- the address unit, not the VALU, becomes the limit.
template <> struct Loader<UG_PF_RGB> : Loader3<false> {};
template <> struct Loader<UG_PF_YUV444> : Loader3<true> {};
template <> struct Loader<UG_PF_RGBA> : LoaderRGBAWords {};

// ---- v210: 12 px = 3 blocks = 32 B per row.  The 10-bit samples come in UYVY order, three
// per little-endian word; the reference converts to 8-bit UYVY by >>2 first
// (vc_copylinev210, pixfmt_conv.c:86-130, selected by cuda_dxt.cpp:162) ----
template <>
struct Loader<UG_PF_V210> {
        static constexpr int kBlocks = 3;
        uint32_t w[4][8];
        template <bool EDGE = false> // (a v210 line is padded to 128 bytes whatever the width, video_codec.c:507-521: whole units are always there)
        __device__ __forceinline__ void load(const uint8_t *src, uint32_t pitch, uint32_t unit_x, const int (&rows)[4])
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const uint4 *p = (const uint4 *) (src + ((uint32_t) rows[r] * pitch + unit_x * 32u));
                        const uint4 q0 = p[0], q1 = p[1];
                        w[r][0] = q0.x; w[r][1] = q0.y; w[r][2] = q0.z; w[r][3] = q0.w;
                        w[r][4] = q1.x; w[r][5] = q1.y; w[r][6] = q1.z; w[r][7] = q1.w;
                }
        }
        // sample s (0..23) of the row, top 8 bits of the 10-bit field
        __device__ __forceinline__ float samp(int r, int s) const
        {
                return (float) ((w[r][s / 3] >> (10 * (s % 3) + 2)) & 0xffu) * kInv255;
        }
        __device__ __forceinline__ void block(int k, Px16 &p) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                                const int s = 8 * k + 4 * pr; // U Y0 V Y1
                                yuv_pair_to_rgb(samp(r, s + 1), samp(r, s + 3), samp(r, s), samp(r, s + 2), p, 4 * r + 2 * pr);
                        }
                }
        }
        // the same for a block that may be the cut one of a width = 2 (mod 4): `half` = only its first pixel pair exists, the
        // second repeats the pair's last pixel (Y1 with the pair's chroma)
        __device__ __forceinline__ void block_edge(int k, Px16 &p, bool half) const
        {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const int s = 8 * k;
                        const float u0 = samp(r, s), y00 = samp(r, s + 1), v0 = samp(r, s + 2), y01 = samp(r, s + 3);
                        yuv_pair_to_rgb(y00, y01, u0, v0, p, 4 * r);
                        const float u1 = half ? u0 : samp(r, s + 4), y10 = half ? y01 : samp(r, s + 5), v1 = half ? v0 : samp(r, s + 6), y11 = half ? y01 : samp(r, s + 7);
                        yuv_pair_to_rgb(y10, y11, u1, v1, p, 4 * r + 2);
                }
        }
};

// ---------------------------------------------------------------------------------------
// DXT5-YCoCg block encode (compress_dxt5ycocg_fp.glsl:326-377 / cuda_dxt.cu:471-509)
// ---------------------------------------------------------------------------------------
template <bool AWAY>
__device__ __forceinline__ uint4 encode_dxt5ycocg(Px16 &p, const IndexTables &tab)
{
        // ConvertRGBToYCoCg (glsl:27-34).  2.0*x and *0.25 are exact (powers of two), so
        //   (r + 2g + b)*0.25      : fma(g,2,r) == r + 2g bit-for-bit (2g exact)
        //   (2r - 2b)*0.25 + off   : 2r-2b == 2(r-b) exactly, *0.25 exact -> (r-b)*0.5 + off,
        //                            and fma(r-b, 0.5, off) == ((r-b)*0.5) + off (product exact)
        //   (-r + 2g - b)*0.25+off : fma(g,2,-r) == -r + 2g ; fma(t,0.25,off) == t*0.25 + off
#pragma unroll
        for (int i = 0; i < 16; i++) {
                const float r = p.a[i], g = p.b[i], b = p.c[i];
                const float t = __builtin_fmaf(g, 2.0f, r);
                p.a[i] = (t + b) * 0.25f;
                p.b[i] = __builtin_fmaf(r - b, 0.5f, kOffset);
                p.c[i] = __builtin_fmaf(__builtin_fmaf(g, 2.0f, -r) - b, 0.25f, kOffset);
        }
        float *Y = p.a, *Co = p.b, *Cg = p.c;

        // FindMinMaxColorsBox (glsl:69-78)
        float mnY = Y[0], mxY = Y[0], mnCo = Co[0], mxCo = Co[0], mnCg = Cg[0], mxCg = Cg[0];
#pragma unroll
        for (int i = 1; i < 16; i++) {
                mnY = fminf(mnY, Y[i]);    mxY = fmaxf(mxY, Y[i]);
                mnCo = fminf(mnCo, Co[i]); mxCo = fmaxf(mxCo, Co[i]);
                mnCg = fminf(mnCg, Cg[i]); mxCg = fmaxf(mxCg, Cg[i]);
        }

        // SelectYCoCgDiagonal (glsl:169-183): sequential sum, i = 0..15
        {
                const float midx = (mxCo + mnCo) * 0.5f, midy = (mxCg + mnCg) * 0.5f;
                float cov = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                        const float tx = Co[i] - midx, ty = Cg[i] - midy;
                        cov = cov + tx * ty;
                }
                if (cov < 0.0f) {
                        const float t = mxCg; mxCg = mnCg; mnCg = t;
                }
        }

        // ScaleYCoCg (glsl:150-167)
        uint32_t scale = 1;
        float fs = 1.0f, rfs = 1.0f; // float(scale) and its exact reciprocal
        {
                const float m0 = fmaxf(fabsf(mnCo - kOffset), fabsf(mnCg - kOffset));
                const float m1 = fmaxf(fabsf(mxCo - kOffset), fabsf(mxCg - kOffset));
                const float m = fmaxf(m0, m1);
                if (m < (float) (64.0 / 255.0)) { scale = 2; fs = 2.0f; rfs = 0.5f; }
                if (m < (float) (32.0 / 255.0)) { scale = 4; fs = 4.0f; rfs = 0.25f; }
        }

        // EmitEndPointsYCoCgDXT5 (glsl:185-215) with InsetCoCgBBox (glsl:92-97).
        // "/ 16.0" and "/ float(scale)" are multiplications by exact powers of two.
        uint32_t w_end;
        float cmx[2], cmn[2];
        {
                const float q[2] = { 31.0f, 63.0f };
                const float mx_in[2] = { mxCo, mxCg }, mn_in[2] = { mnCo, mnCg };
                uint32_t imax[2], imin[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                        float a = (mx_in[k] - kOffset) * fs + kOffset;
                        float b = (mn_in[k] - kOffset) * fs + kOffset;
                        const float inset = (a - b) * 0.0625f - kInsetC;
                        b = clamp01(b + inset);
                        a = clamp01(a - inset);
                        imax[k] = round_u32<AWAY>(a * q[k]);
                        imin[k] = round_u32<AWAY>(b * q[k]);
                }
                w_end = ((imax[0] << 11) | (imax[1] << 5) | (scale - 1)) |
                        (((imin[0] << 11) | (imin[1] << 5) | (scale - 1)) << 16);
                imax[0] = (imax[0] << 3) | (imax[0] >> 2);
                imax[1] = (imax[1] << 2) | (imax[1] >> 4);
                imin[0] = (imin[0] << 3) | (imin[0] >> 2);
                imin[1] = (imin[1] << 2) | (imin[1] >> 4);
                const float inv255 = (float) (1.0 / 255.0);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                        cmx[k] = ((float) imax[k] * inv255 - kOffset) * rfs + kOffset;
                        cmn[k] = ((float) imin[k] * inv255 - kOffset) * rfs + kOffset;
                }
        }

        // InsetYBBox (glsl:86-91)
        {
                const float inset = (mxY - mnY) * 0.03125f - kInsetY;
                mnY = clamp01(mnY + inset);
                mxY = clamp01(mxY - inset);
        }
        // EmitAlphaEndPointsYCoCgDXT5 (glsl:252-259)
        uint32_t w0 = (round_u32<AWAY>(mnY * 255.0f) << 8) | round_u32<AWAY>(mxY * 255.0f);
        uint32_t w1 = 0;
        // EmitAlphaIndicesYCoCgDXT5 (glsl:262-312): count c = #{k : a <= ab_k}, index = f(c).
        {
                const float inv7 = (float) (1.0 / 7.0);
                // (mxY - mnY) / 14.0f.  div14() is bit-identical to the division for x = 0 and for every x >= 2^-100 (exhaustive
                // device self-test); differences of byte-derived luma never fall in between, but the contract does not rest
                // on that: a wave that sees such a value takes the IEEE division.
                const float range = mxY - mnY;
#ifdef UG_DXT_IEEE_DIV14 // A/B switch: the plain division
                const float mid = range / 14.0f;
#else
                float mid = div14(range);
                if (__builtin_expect(__any((__float_as_uint(range) - 1u) < (kDiv14Exact - 1u)), 0)) {
                        mid = range / 14.0f;
                }
#endif
                float ab[8];
                ab[1] = mnY + mid;
#pragma unroll
                for (int k = 2; k <= 7; k++) {
                        ab[k] = ((float) (8 - k) * mxY + (float) (k - 1) * mnY) * inv7 + mid;
                }
                // Thresholds in ascending order are T1..T7 = ab1, ab7, ab6, ab5, ab4, ab3, ab2 whenever they are
                // monotone (always, except degenerate blocks whose clamped min == max).  For a monotone set the
                // count is a 3-step binary search (3 compares + 4 selects instead of 7 compares + 7 adds); it is
                // the SAME function of (a, ab[]) as the reference's linear count, so results stay bit-identical.
                const float T1 = ab[1], T2 = ab[7], T3 = ab[6], T4 = ab[5], T5 = ab[4], T6 = ab[3], T7 = ab[2];
                // Monotone for sure when the clamped range exceeds 2^-10: consecutive thresholds are range/7 apart in exact
                // arithmetic and each carries < 2^-22 of rounding error (operands <= 7, results <= 1), so 2^-10/7 of spacing
                // cannot be overturned.  After the inset the range is >= 16/255/16 unless both ends clamp to the same bound,
                // so one compare decides for practically every block; only a wave that sees a narrower range evaluates the six
                // explicit comparisons.
                bool all_mono, all_wide = false; // both wave-uniform
                (void) all_wide;
                if (__builtin_expect(__all(range > 0.0009765625f), 1)) {
                        all_mono = true;
                        all_wide = true;
                } else {
                        asm volatile("; explicit monotonicity check" ::: "memory");
#if UG_DXT_FAST_INDEX
                        count_full_form(1);
#endif
                        all_mono = __all((T1 <= T2) & (T2 <= T3) & (T3 <= T4) & (T4 <= T5) & (T5 <= T6) & (T6 <= T7));
                }
                // raw counts, 3 bits per pixel: lo = px 0..9 (30 bits), hi = px 10..15 (18 bits)
                uint32_t lo = 0, hi = 0;
#if UG_DXT_FAST_INDEX && !defined(UG_FORCE_ALPHA_LINEAR)
                // Fast form of the same count, valid under the wave-uniform range > 2^-10 test above.  The thresholds sit range/7
                // apart (+- 2^-21), so the position of a among them is known from ONE multiply-add up to the nearest threshold:
                //   t = 7 - (a - mnY) * 7/range (clamped to [0, 8)) puts threshold T(8-m) at t = m - 1/2  (m = 1..7, descending thresholds D_m = T(8-m));
                //   g = trunc(t) (saturating at 0): every D_m with m <= g lies >= 0.49 steps above a, every D_m with m >= g + 2 lies
                //   >= 0.49 steps below it (error of t < 6e-4 steps: rcp 1 ulp, mnY*inv rounded at magnitude <= 7168, fma rounding), so
                //   c = #{m : a <= D_m} = g + (a <= D_(g+1)) -- and THAT comparison is the reference's own, on the reference's own
                //   fp32 threshold (read back from the per-lane LDS column).  Row 7 = -inf serves g = 7 (a below every threshold).
                // 1 fma + cvt + address + compare + 2 integer ops per pixel instead of 3 compares + 4 selects + 3 carries.
                if (__builtin_expect(all_wide, 1)) {
                        float *const ta = tab.alpha;
                        ta[0 * 64] = T7; ta[1 * 64] = T6; ta[2 * 64] = T5; ta[3 * 64] = T4;
                        ta[4 * 64] = T3; ta[5 * 64] = T2; ta[6 * 64] = T1; ta[7 * 64] = -__builtin_inff();
                        // t / 8 with the clamp modifier (luma of YUV sources can lie far outside [mnY, mxY]: g must stay in 0..7), then * (8 - ulp)
                        const float inv = 0.875f * __builtin_amdgcn_rcpf(range);
                        const float t0 = __builtin_fmaf(mnY, inv, 0.875f), ninv = -inv;
                        constexpr int kA = UG_DXT5_ALPHA_GROUP;
#pragma unroll
                        for (int h = 16 / kA - 1; h >= 0; h--) { // groups of kA pixels: kA table reads in flight, bounded register use
                                float D[kA];
                                uint32_t g[kA];
#pragma unroll
                                for (int j = kA - 1; j >= 0; j--) {
                                        g[j] = cvt_u32_sat(clamp01(__builtin_fmaf(Y[kA * h + j], ninv, t0)) * 7.9999995f);
                                        D[j] = ta[g[j] * 64];
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = kA - 1; j >= 0; j--) {
                                        const int i = kA * h + j;
                                        uint32_t &acc = i >= 10 ? hi : lo;
                                        acc = add_mask((acc << 3) + g[j], LANEMASK(Y[i] <= D[j]));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                        }
                } else
#endif
#ifdef UG_FORCE_ALPHA_LINEAR // test build: always take the reference-form path (tests/test_gpu_dxt.py)
                if (all_mono && lo == 0xffffffffu) {
#else
                if (__builtin_expect(all_mono, 1)) {
#endif
#pragma unroll
                        for (int i = 15; i >= 0; i--) {
                                const float a = Y[i];
                                const bool g4 = a <= T4;
                                const float t2 = g4 ? T2 : T6, t15 = g4 ? T1 : T5, t37 = g4 ? T3 : T7;
                                const bool g2 = a <= t2;
                                const float t1 = g2 ? t15 : t37;
                                const bool g1 = a <= t1;
                                uint32_t &acc = i >= 10 ? hi : lo;
                                acc = shift_in(acc, LANEMASK(g4));
                                acc = shift_in(acc, LANEMASK(g2));
                                acc = shift_in(acc, LANEMASK(g1));
                        }
                } else { // reference form, 7 compares per pixel, for every lane of the wave (degenerate blocks only)
                        // The empty asm keeps this a real branch: without it the compiler if-converts the two
                        // sides and every block pays for both.
                        asm volatile("; alpha linear fallback" ::: "memory");
#pragma unroll
                        for (int i = 15; i >= 0; i--) {
                                const float a = Y[i];
                                uint32_t c = 0;
#pragma unroll
                                for (int k = 1; k <= 7; k++) {
                                        c += (a <= ab[k]) ? 1u : 0u;
                                }
                                uint32_t &acc = i >= 10 ? hi : lo;
                                acc = (acc << 3) | c;
                        }
                }
                // index = ((c + 1) & 7) ^ (((c + 1) & 7) < 2)   (glsl:281-290), i.e. 0->0, 1..6 -> c+1, 7->1,
                // applied to all 3-bit fields of a word at once: with field bits (c2 c1 c0)
                //   i0 = (~c0 & (c1 | c2)) | (c0 & c1 & c2),  i1 = c1 ^ c0,  i2 = c2 ^ (c1 & c0)
                auto map_fields = [](uint32_t w) {
                        const uint32_t M = 0x09249249u; // bit 0 of every 3-bit field
                        const uint32_t c0 = w & M, c1 = (w >> 1) & M, c2 = (w >> 2) & M;
                        const uint32_t t = c1 & c0;
                        const uint32_t i0 = ((c0 ^ M) & (c1 | c2)) | (t & c2);
                        const uint32_t i1 = c1 ^ c0, i2 = c2 ^ t;
                        return i0 | (i1 << 1) | (i2 << 2);
                };
                lo = map_fields(lo);
                hi = map_fields(hi);
                // 48-bit index field F = lo | hi << 30: word0[31:16] = F[15:0], word1 = F[47:16]  (glsl:291-308)
                w0 |= lo << 16;
                w1 = (lo >> 16) | (hi << 14);
        }
        // EmitIndicesYCoCgDXT5 (glsl:217-250).  The four squared distances of TWO horizontally adjacent
        // pixels are computed with packed fp32 (v_pk_add/v_pk_mul: IEEE per component, so bit-identical to the
        // scalar form); the kernel is VALU-issue bound and this halves the issue slots of its largest stage.
        uint32_t w_cidx = 0;
        {
                const float q1 = (float) (1.0 / 3.0), q2 = (float) (2.0 / 3.0);
                const float w1 = 1.0f - q1, w2 = 1.0f - q2;
                float cx[4], cy[4];
                cx[0] = cmx[0]; cy[0] = cmx[1];
                cx[1] = cmn[0]; cy[1] = cmn[1];
                cx[2] = lerp_w(cx[0], cx[1], w1, q1); cy[2] = lerp_w(cy[0], cy[1], w1, q1);
                cx[3] = lerp_w(cx[0], cx[1], w2, q2); cy[3] = lerp_w(cy[0], cy[1], w2, q2);
#if UG_DXT_FAST_INDEX
                // Fast form (wave-uniform choice).  The palette lies on the segment c0 -> c1 in the order c0, c2, c3, c1 (s = 0, 1/3, 2/3, 1)
                // and the index formula of glsl:237-244 is "nearest of the four": with s = the pixel's projection on the segment,
                //   b2 = [d0 > d2] flips at s = 1/6, b0 = [d0 > d3] at 1/3, b4 = [d2 > d3] at 1/2, b1 = [d1 > d2] at 2/3, b3 = [d1 > d3] at 5/6,
                //   bit0 = b0 & b4,  bit1 = (b1 & b2) | (b0 & b3).
                // In the third of the segment that holds the pixel (zone k = trunc(3 s), s clamped to [0, 1)) exactly ONE of the three
                // decisive comparisons (b2 | b4 | b3) is open; for the other bits either the pixel is >= 1/12 of the segment away
                // from the comparison's bisector, or the bit cannot change the result (b0 near 1/3 and b1 near 2/3 are masked by
                // their partners):   k = 0: index = 2 b2;   k = 1: index = 2 + b4;   k = 2: index = 1 + 2 b3.
                // The open comparison is evaluated exactly as the reference does (both squared distances in the reference's
                // operation order, strict >), so the result is the reference's whenever the "certain" bits are certain in fp32:
                //   a computed distance is off by < 2^-22 |p - c|^2 <= 2^-22 dmax (dmax = squared diagonal of the box around pixels and
                //   end points); two distances whose bisector is m segment-lengths away differ by >= (2/3) m vv (vv = |c1 - c0|^2).
                //   With vv * 256 > dmax and m >= 1/12 - 4e-4 that is > 200 x the rounding error; c2 / c3 are off their ideal
                //   places by < 2e-7 absolute = < 6e-5 of a segment with vv >= 1e-5 (a non-zero segment of 8-bit-expanded end points
                //   has vv >= 1.5e-5), and s itself is estimated to < 3e-4 (reciprocal + 5 roundings at magnitudes <= 650).
                // A wave that holds a block outside that precondition (coincident end points over non-flat chroma) runs the full form below for all lanes.
                const float vx = cx[1] - cx[0], vy = cy[1] - cy[0];
                const float vv = vx * vx + vy * vy;
                const float e0 = fmaxf(fmaxf(mxCo, cx[0]), cx[1]) - fminf(fminf(mnCo, cx[0]), cx[1]);
                const float e1 = fmaxf(fmaxf(fmaxf(mxCg, mnCg), cy[0]), cy[1]) - fminf(fminf(fminf(mxCg, mnCg), cy[0]), cy[1]);
                const float dmax = e0 * e0 + e1 * e1;
                // A block whose 16 pixels share ONE chroma value (flat areas; its end points usually coincide) gets the reference's full
                // form for that one value, replicated; only a wave that holds such a block pays for it.
                const bool flat = (mnCo == mxCo) & (mnCg == mxCg);
                if (__builtin_expect(__all(((vv >= 1e-5f) & (vv * 256.0f > dmax)) | flat), 1)) {
                        float4 *const tc = tab.colour;
                        tc[0 * 64] = make_float4(cx[0], cy[0], cx[2], cy[2]); // zone 0: d0 > d2
                        tc[1 * 64] = make_float4(cx[2], cy[2], cx[3], cy[3]); // zone 1: d2 > d3
                        tc[2 * 64] = make_float4(cx[1], cy[1], cx[3], cy[3]); // zone 2: d1 > d3
                        const float inv = __builtin_amdgcn_rcpf(vv);
                        const float ka = vx * inv, kb = vy * inv;
                        const float kc = -(cx[0] * ka + cy[0] * kb);
                        uint32_t zones = 0, open = 0; // 2-bit zone per pixel; the open comparison's result, one bit per pixel
                        // groups of kC pixels, the table reads of the next group issued before the distances of this one
                        constexpr int kC = UG_DXT5_FAST_GROUP, kGroups = 16 / kC;
                        float4 e[2][kC];
                        uint32_t kk[2][kC];
                        auto fetch = [&](int grp, int slot) {
#pragma unroll
                                for (int j = kC - 1; j >= 0; j--) {
                                        const int i = kC * grp + j;
                                        const float s = clamp01(__builtin_fmaf(Co[i], ka, __builtin_fmaf(Cg[i], kb, kc)));
                                        kk[slot][j] = cvt_u32_sat(s * 2.9999998f); // 0, 1, 2
                                        e[slot][j] = tc[kk[slot][j] * 64];
                                }
                        };
                        fetch(kGroups - 1, (kGroups - 1) & 1);
#pragma unroll
                        for (int grp = kGroups - 1; grp >= 0; grp--) {
                                const int slot = grp & 1;
                                if (grp > 0) {
                                        fetch(grp - 1, slot ^ 1);
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = kC - 1; j >= 0; j--) {
                                        const int i = kC * grp + j;
                                        const float4 q = e[slot][j];
                                        const float ax = Co[i] - q.x, ay = Cg[i] - q.y, bx = Co[i] - q.z, by = Cg[i] - q.w;
                                        const float da = ax * ax + ay * ay, db = bx * bx + by * by; // glsl:231-235, same operation order
                                        zones = (zones << 2) + kk[slot][j];
                                        open = shift_in(open, LANEMASK(da > db));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                        }
                        const uint32_t c = spread16(open), k0 = zones & 0x55555555u, k1 = (zones >> 1) & 0x55555555u;
                        w_cidx = ((k0 & c) | k1) | ((k0 | c) << 1);
                        if (__any(flat)) {
                                asm volatile("; flat blocks" ::: "memory");
                                const float4 p02 = tc[0 * 64], p13 = tc[2 * 64]; // the palette, back from the table (not kept in registers)
                                const float px[4] = { p02.x, p13.x, p02.z, p13.z }, py[4] = { p02.y, p13.y, p02.w, p13.w };
                                float d[4];
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                        const float tx = Co[0] - px[k], ty = Cg[0] - py[k];
                                        d[k] = tx * tx + ty * ty;
                                }
                                w_cidx = flat ? palette_index(d[0], d[1], d[2], d[3]) * 0x55555555u : w_cidx;
                        }
                } else
#endif
                {
                asm volatile("; colour index full form" ::: "memory");
#if UG_DXT_FAST_INDEX
                count_full_form(0);
#endif
                f32x2 cx2[4], cy2[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        cx2[k] = (f32x2) { cx[k], cx[k] };
                        cy2[k] = (f32x2) { cy[k], cy[k] };
                }
#pragma unroll
                for (int i = 14; i >= 0; i -= 2) { // pixel pairs, last first: bits are shifted in MSB first
                        const f32x2 co = { Co[i], Co[i + 1] }, cg = { Cg[i], Cg[i + 1] };
                        f32x2 d[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                                const f32x2 tx = co - cx2[k], ty = cg - cy2[k];
                                d[k] = tx * tx + ty * ty;
                        }
#pragma unroll
                        for (int h = 1; h >= 0; h--) { // pixel i+1, then pixel i
                                const float d0 = d[0][h], d1 = d[1][h], d2 = d[2][h], d3 = d[3][h];
                                // glsl:237-244
                                const lanemask_t b0 = LANEMASK(d0 > d3), b1 = LANEMASK(d1 > d2), b2 = LANEMASK(d0 > d2),
                                                 b3 = LANEMASK(d1 > d3), b4 = LANEMASK(d2 > d3);
                                w_cidx = shift_in(w_cidx, (b1 & b2) | (b0 & b3)); // bit 2i+1
                                w_cidx = shift_in(w_cidx, b0 & b4);               // bit 2i
                        }
                }
                }
        }

        return make_uint4(w0, w1, w_end, w_cidx);
}

// ---------------------------------------------------------------------------------------
// DXT1 block encode, normative = GLSL (compress_dxt1_fp.glsl:177-229)
// ---------------------------------------------------------------------------------------
template <bool AWAY>
__device__ __forceinline__ uint2 encode_dxt1(const Px16 &p, const IndexTables &tab)
{
        const float *R = p.a, *G = p.b, *B = p.c;
        float mn[3] = { R[0], G[0], B[0] }, mx[3] = { R[0], G[0], B[0] };
#pragma unroll
        for (int i = 1; i < 16; i++) {
                mn[0] = fminf(mn[0], R[i]); mx[0] = fmaxf(mx[0], R[i]);
                mn[1] = fminf(mn[1], G[i]); mx[1] = fmaxf(mx[1], G[i]);
                mn[2] = fminf(mn[2], B[i]); mx[2] = fmaxf(mx[2], B[i]);
        }
        // SelectDiagonal (glsl:69-90)
        {
                const float cx = (mn[0] + mx[0]) * 0.5f, cy = (mn[1] + mx[1]) * 0.5f, cz = (mn[2] + mx[2]) * 0.5f;
                float cov_x = 0.0f, cov_y = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                        const float tx = R[i] - cx, ty = G[i] - cy, tz = B[i] - cz;
                        cov_x = cov_x + tx * tz;
                        cov_y = cov_y + ty * tz;
                }
                if (cov_x < 0.0f) { const float t = mx[0]; mx[0] = mn[0]; mn[0] = t; }
                if (cov_y < 0.0f) { const float t = mx[1]; mx[1] = mn[1]; mn[1] = t; }
        }
        // InsetBBox (glsl:92-97), RoundAndExpand + EmitEndPointsDXT1 (glsl:99-126)
        uint32_t cm[3], cn[3];
        {
                const float q[3] = { 31.0f, 63.0f, 31.0f };
#pragma unroll
                for (int k = 0; k < 3; k++) {
                        const float inset = (mx[k] - mn[k]) * 0.0625f - kInsetC;
                        const float lo = clamp01(mn[k] + inset), hi = clamp01(mx[k] - inset);
                        cm[k] = round_u32<AWAY>(hi * q[k]);
                        cn[k] = round_u32<AWAY>(lo * q[k]);
                }
        }
        const uint32_t code_max = (cm[0] << 11) | (cm[1] << 5) | cm[2];
        const uint32_t code_min = (cn[0] << 11) | (cn[1] << 5) | cn[2];
        cm[0] = (cm[0] << 3) | (cm[0] >> 2); cm[2] = (cm[2] << 3) | (cm[2] >> 2); cm[1] = (cm[1] << 2) | (cm[1] >> 4);
        cn[0] = (cn[0] << 3) | (cn[0] >> 2); cn[2] = (cn[2] << 3) | (cn[2] >> 2); cn[1] = (cn[1] << 2) | (cn[1] >> 4);
        const float inv255 = (float) (1.0 / 255.0);
        const bool swap = code_max < code_min;
        float c0[3], c1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
                const float hi = (float) cm[k] * inv255, lo = (float) cn[k] * inv255;
                c0[k] = swap ? lo : hi;
                c1[k] = swap ? hi : lo;
        }
        const uint32_t w_end = swap ? (code_min | (code_max << 16)) : (code_max | (code_min << 16));

        // EmitIndicesDXT1 (glsl:128-161): packed fp32 over pixel pairs, as in the DXT5-YCoCg encoder
        uint32_t w_idx = 0;
        {
                const float q1 = (float) (1.0 / 3.0), q2 = (float) (2.0 / 3.0);
                const float w1 = 1.0f - q1, w2 = 1.0f - q2;
                float c2[3], c3[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                        c2[k] = lerp_w(c0[k], c1[k], w1, q1);
                        c3[k] = lerp_w(c0[k], c1[k], w2, q2);
                }
#if UG_DXT_FAST_INDEX
                // Fast form, as in encode_dxt5ycocg (the argument does not depend on the number of components): zone of the pixel's
                // projection on c0 -> c1, the one open comparison evaluated exactly as the reference does.  A distance of three squares
                // is off by < 2^-21 dmax, still > 100 x below the smallest certain difference under the same precondition; a non-zero
                // segment of 8-bit-expanded end points has vv >= 2.4e-4.  Coincident end points (flat blocks): full form for the wave.
                const float vx = c1[0] - c0[0], vy = c1[1] - c0[1], vz = c1[2] - c0[2];
                const float vv = vx * vx + vy * vy + vz * vz;
                float dmax = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; k++) { // mn / mx may be swapped by SelectDiagonal
                        const float e = fmaxf(fmaxf(fmaxf(mx[k], mn[k]), c0[k]), c1[k]) - fminf(fminf(fminf(mx[k], mn[k]), c0[k]), c1[k]);
                        dmax = dmax + e * e;
                }
                const bool flat = (mn[0] == mx[0]) & (mn[1] == mx[1]) & (mn[2] == mx[2]); // one colour: full form for it, replicated
                if (__builtin_expect(__all(((vv >= 1e-5f) & (vv * 256.0f > dmax)) | flat), 1)) {
                        float4 *const tc = tab.colour;
                        tc[0 * 64] = make_float4(c0[0], c0[1], c0[2], 0.0f); tc[1 * 64] = make_float4(c2[0], c2[1], c2[2], 0.0f); // zone 0: d0 > d2
                        tc[2 * 64] = make_float4(c2[0], c2[1], c2[2], 0.0f); tc[3 * 64] = make_float4(c3[0], c3[1], c3[2], 0.0f); // zone 1: d2 > d3
                        tc[4 * 64] = make_float4(c1[0], c1[1], c1[2], 0.0f); tc[5 * 64] = make_float4(c3[0], c3[1], c3[2], 0.0f); // zone 2: d1 > d3
                        const float inv = __builtin_amdgcn_rcpf(vv);
                        const float ka = vx * inv, kb = vy * inv, kc = vz * inv;
                        const float kd = -((c0[0] * ka + c0[1] * kb) + c0[2] * kc);
                        uint32_t zones = 0, open = 0;
                        // groups of kG pixels; the table reads (2 x 12 bytes per pixel) of the next group are issued before the distances of this one
                        constexpr int kG = UG_DXT1_FAST_GROUP;
                        struct alignas(16) F3 { float x, y, z; };
                        F3 ea[2][kG], eb[2][kG];
                        uint32_t kk[2][kG];
                        auto fetch = [&](int grp, int slot) {
#pragma unroll
                                for (int j = kG - 1; j >= 0; j--) {
                                        const int i = kG * grp + j;
                                        const float t = clamp01(__builtin_fmaf(R[i], ka, __builtin_fmaf(G[i], kb, __builtin_fmaf(B[i], kc, kd))));
                                        kk[slot][j] = cvt_u32_sat(t * 2.9999998f); // 0, 1, 2
                                        const F3 *const q = (const F3 *) (tc + kk[slot][j] * 128);
                                        ea[slot][j] = q[0];
                                        eb[slot][j] = q[64];
                                }
                        };
                        constexpr int kGroups = 16 / kG;
                        fetch(kGroups - 1, (kGroups - 1) & 1);
#pragma unroll
                        for (int grp = kGroups - 1; grp >= 0; grp--) {
                                const int slot = grp & 1;
                                if (grp > 0) {
                                        fetch(grp - 1, slot ^ 1);
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = kG - 1; j >= 0; j--) {
                                        const int i = kG * grp + j;
                                        const F3 qa = ea[slot][j], qb = eb[slot][j];
                                        const float ax = R[i] - qa.x, ay = G[i] - qa.y, az = B[i] - qa.z;
                                        const float bx = R[i] - qb.x, by = G[i] - qb.y, bz = B[i] - qb.z;
                                        const float da = AWAY ? (ax * ax + ay * ay) + az * az : (az * az + ay * ay) + ax * ax; // as d[k] below
                                        const float db = AWAY ? (bx * bx + by * by) + bz * bz : (bz * bz + by * by) + bx * bx;
                                        zones = (zones << 2) + kk[slot][j];
                                        open = shift_in(open, LANEMASK(da > db));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                        }
                        const uint32_t c = spread16(open), k0 = zones & 0x55555555u, k1 = (zones >> 1) & 0x55555555u;
                        w_idx = ((k0 & c) | k1) | ((k0 | c) << 1);
                        if (__any(flat)) {
                                asm volatile("; flat blocks" ::: "memory");
                                const float4 pal[4] = { tc[0 * 64], tc[4 * 64], tc[1 * 64], tc[3 * 64] }; // the palette, back from the table
                                float d[4];
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                        const float tx = R[0] - pal[k].x, ty = G[0] - pal[k].y, tz = B[0] - pal[k].z;
                                        d[k] = AWAY ? (tx * tx + ty * ty) + tz * tz : (tz * tz + ty * ty) + tx * tx;
                                }
                                w_idx = flat ? palette_index(d[0], d[1], d[2], d[3]) * 0x55555555u : w_idx;
                        }
                } else
#endif
                {
                asm volatile("; colour index full form" ::: "memory");
#if UG_DXT_FAST_INDEX
                count_full_form(0);
#endif
                const float *c[4] = { c0, c1, c2, c3 };
                f32x2 cc[4][3];
#pragma unroll
                for (int k = 0; k < 4; k++) {
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                                cc[k][ch] = (f32x2) { c[k][ch], c[k][ch] };
                        }
                }
#pragma unroll
                for (int i = 14; i >= 0; i -= 2) {
                        const f32x2 r = { R[i], R[i + 1] }, g = { G[i], G[i + 1] }, bl = { B[i], B[i + 1] };
                        f32x2 d[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                                const f32x2 tx = r - cc[k][0], ty = g - cc[k][1], tz = bl - cc[k][2];
                                d[k] = AWAY ? (tx * tx + ty * ty) + tz * tz  // dot(v,v): x*x + y*y + z*z, left to right (cuda_dxt.cu:106-108)
                                            : (tz * tz + ty * ty) + tx * tx; // Mesa sums dot(vec3) from the last component
                        }
#pragma unroll
                        for (int h = 1; h >= 0; h--) {
                                const float d0 = d[0][h], d1 = d[1][h], d2 = d[2][h], d3 = d[3][h];
                                const lanemask_t b0 = LANEMASK(d0 > d3), b1 = LANEMASK(d1 > d2), b2 = LANEMASK(d0 > d2),
                                                 b3 = LANEMASK(d1 > d3), b4 = LANEMASK(d2 > d3);
                                w_idx = shift_in(w_idx, (b1 & b2) | (b0 & b3));
                                w_idx = shift_in(w_idx, b0 & b4);
                        }
                }
                }
        }
        return make_uint2(w_end, w_idx);
}

// ---------------------------------------------------------------------------------------
// kernel.  Workgroup = 64 x 4 lanes: wave w of the group encodes 64 consecutive units of block row
// blockIdx.y*4 + w (unit = Loader::kBlocks consecutive blocks), blockIdx.z = image of the batch.
// No integer division anywhere; lanes idle only at the right/bottom edge of the block grid.
// ---------------------------------------------------------------------------------------
// Occupancy target of the register allocator.  With the fast index stages the DXT5-YCoCg kernels need 100 VGPRs at 4 waves per SIMD
// or 96 at 5 (the five dwords that do not fit are spilled in the rarely taken full-form colour stage only): 5 measures 2.4 % faster
// (profiles/r03_fast_index_ab.txt).  The v210 kernels (three blocks per lane) keep the allocator's own choice.
#ifndef UG_DXT_MIN_WAVES
#define UG_DXT_MIN_WAVES (UG_DXT_FAST_INDEX ? 5 : 1)
#endif
#ifndef UG_DXT_ROWS_PER_WAVE
#define UG_DXT_ROWS_PER_WAVE 1
#endif
// A wave can walk kRowsPerWave consecutive block rows, issuing the loads of row j+1 before it encodes row j.
// Measured on MI355X (interleaved A/B, 16 x 4K UYVY->DXT5): 1 row 0.2004 ms, 2 rows 0.2157, 4 rows 0.2224, 8 rows
// 0.2230 -- the prefetching loop LOSES (VGPR 104 -> 126, no cross-row scheduling), so the default is one row per
// wave and latency hiding is left to the 4 resident waves per SIMD.
constexpr int kRowsPerWave = UG_DXT_ROWS_PER_WAVE;

template <int IN, int OUT, bool MIRROR, bool AWAY, bool EDGE>
__global__ __launch_bounds__(256, (Loader<IN>::kBlocks > 1 ? 1 : UG_DXT_MIN_WAVES)) void dxt_encode_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                         int units_per_row, int blocks_per_row, int block_rows, int height, uint32_t pitch,
                                                         size_t src_frame_stride, size_t dst_frame_stride, int width)
{
        using L = Loader<IN>;
        constexpr bool kTables = UG_DXT_FAST_INDEX, kAlpha = kTables && OUT == UG_DXT5_YCOCG;
        constexpr int kColourRows = OUT == UG_DXT5_YCOCG ? 3 : 6;
        __shared__ float lds_alpha[kAlpha ? 4 * 8 * 64 : 1];
        __shared__ float4 lds_colour[kTables ? 4 * kColourRows * 64 : 1];
        IndexTables tab;
        {
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
                tab.alpha = lds_alpha + (kAlpha ? wave * (8 * 64) + (int) threadIdx.x : 0);
                tab.colour = lds_colour + (kTables ? wave * (kColourRows * 64) + (int) threadIdx.x : 0);
        }
#if UG_DXT_LDS_CONV
        __shared__ ConvTables lds_conv;
        if (IN == UG_PF_UYVY) { // 256 threads, one entry each: the statements of yuv_pair_to_rgb
                const int b = (int) (threadIdx.y * 64 + threadIdx.x);
                const float x = (float) b * kInv255;
                const float U = x - 0.5f;
                lds_conv.y[b] = 1.1643f * (x - 0.0625f);
                lds_conv.v[b] = make_float2(1.7926f * U, 0.5328f * U);
                lds_conv.u[b] = make_float2(0.2132f * U, 2.1124f * U);
                __syncthreads();
        }
#endif
        // threadIdx.y is wave-uniform (a wave is one 64-lane row of the group): keep the block row, the row base
        // pointers and the bounds test on the scalar unit -- no per-lane 64-bit multiplies in the prologue.
        const int ux = blockIdx.x * 64 + threadIdx.x;
        const int by0 = (int) (blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y)) * kRowsPerWave;
        if (by0 >= block_rows) {
                return;
        }
        if (ux >= units_per_row) {
                return;
        }
        src += (size_t) blockIdx.z * src_frame_stride;
        dst += (size_t) blockIdx.z * dst_frame_stride;

        auto load_row = [&](L &ld, int by) {
                int rows[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                        const int y = EDGE ? min(4 * by + r, height - 1) : 4 * by + r; // (scalar) lines past the picture repeat its last line
                        rows[r] = MIRROR ? height - 1 - y : y; // cuda_dxt.cu:652-655
                }
                if constexpr (EDGE && L::kBlocks == 1) {
                        const int valid = width - 4 * ux; // pixel columns of this lane's block that lie inside the picture
                        if (valid < 4) { // one lane per block row, when width % 4 != 0
                                ld.load_edge(src, pitch, ux, rows, valid);
                                return;
                        }
                }
                ld.template load<EDGE>(src, pitch, ux, rows);
        };
        L cur, nxt;
#if UG_DXT_LDS_CONV
        if constexpr (IN == UG_PF_UYVY) { cur.lut = &lds_conv; nxt.lut = &lds_conv; }
#endif
        load_row(cur, by0);
#pragma unroll 1
        for (int j = 0; j < kRowsPerWave; j++) {
                const int by = by0 + j;
                if (by >= block_rows) { // wave-uniform
                        break;
                }
                const bool more = j + 1 < kRowsPerWave && by + 1 < block_rows;
                if (more) {
                        load_row(nxt, by + 1);
                }
                // block raster order idx = bx + (w/4)*by (cuda_dxt.cu:633)
                constexpr uint32_t kBlockBytes = OUT == UG_DXT5_YCOCG ? 16 : 8;
                uint8_t *const dst_row = dst + (size_t) by * blocks_per_row * kBlockBytes; // scalar
                const uint32_t dst_off = (uint32_t) ux * (L::kBlocks * kBlockBytes);
                // A lane with several blocks (v210: 3) keeps them and stores them together at the end: stored one by one, a thousand instructions
                // apart, the three 16-byte pieces of a lane's 48 bytes reached HBM as partial lines (WRITE_SIZE 1.5 x the output, rocprofv3)
                uint4 res[L::kBlocks];
                int n_res = 0;
#pragma unroll
                for (int k = 0; k < L::kBlocks; k++) {
                        // v210, width % 12 != 0 (1280, 2048 ...): the last 32-byte unit of a line holds one or two blocks; its loads
                        // stay inside the 128-byte-padded line (video_codec.c:507-521), the blocks past the picture are not emitted
                        if (L::kBlocks > 1 && k > 0 && ux * L::kBlocks + k >= blocks_per_row) {
                                break;
                        }
                        Px16 p;
                        if constexpr (EDGE && L::kBlocks > 1) {
                                cur.block_edge(k, p, width - 4 * (ux * L::kBlocks + k) < 4);
                        } else {
                                cur.block(k, p);
                        }
                        if (OUT == UG_DXT5_YCOCG) {
                                res[k] = encode_dxt5ycocg<AWAY>(p, tab);
                        } else {
                                const uint2 e = encode_dxt1<AWAY>(p, tab);
                                res[k] = make_uint4(e.x, e.y, 0, 0);
                        }
                        n_res = k + 1;
                }
#pragma unroll
                for (int k = 0; k < L::kBlocks; k++) {
                        if (k < n_res) {
                                // One block per lane: a store instruction covers whole lines (64 x 16 B / 64 x 8 B contiguous) and streams past L2.
                                // Three blocks per lane (v210): an instruction writes 16 of every 48 bytes and the lines complete only with the
                                // third one -- streamed, the pieces reached HBM separately (WRITE_SIZE 157 MB for 132.7 MB of output, rocprofv3),
                                // so these go through L2 as plain stores and merge there (1.001 x).
                                uint8_t *const at = dst_row + (dst_off + k * kBlockBytes);
                                if (OUT == UG_DXT5_YCOCG) {
                                        if (L::kBlocks > 1) *(uint4 *) at = res[k]; else ug::st_stream((uint4 *) at, res[k]);
                                } else {
                                        const uint2 v = make_uint2(res[k].x, res[k].y);
                                        if (L::kBlocks > 1) *(uint2 *) at = v; else ug::st_stream((uint2 *) at, v);
                                }
                        }
                }
                if (more) {
                        cur = nxt;
                }
        }
}

// exhaustive check of div14() against the IEEE division over x = 0 and every fp32 bit pattern in [2^-100, 1]
__global__ void selftest_div14_kernel(unsigned *mismatches)
{
        const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
        if (i > 0x3f800000ull || (i != 0 && i < kDiv14Exact)) return;
        const float x = __uint_as_float((unsigned) i);
        volatile float d = 14.0f;
        const float want = x / d;
        if (__float_as_uint(div14(x)) != __float_as_uint(want)) atomicAdd(mismatches, 1u);
}

struct EncodeJob {
        const void *src;
        void *dst;
        int w, h, pitch, frames;
        size_t sfs, dfs;
        hipStream_t st;
};

template <int IN, int OUT, bool AWAY>
int launch(const EncodeJob &j)
{
        using L = Loader<IN>;
        const bool mirror = j.h < 0;
        const int h = mirror ? -j.h : j.h;
        const int bpr = (j.w + 3) / 4, upr = (bpr + L::kBlocks - 1) / L::kBlocks, brows = (h + 3) / 4; // dxt_util.h:59-67
        const bool edge = (j.w & 3) || (h & 3);
        if (upr == 0 || brows == 0 || j.frames == 0) return UG_HIP_SUCCESS;
        constexpr int wg_rows = 4; // waves per workgroup; 1, 2 and 4 measure the same (0.1878-0.1889 ms)
        const int rows_per_group = wg_rows * kRowsPerWave;
        if ((uint64_t) j.pitch * (uint64_t) h > 0xffffffffull) { // in-frame byte offsets are 32-bit (scalar base + lane offset)
                ug::set_last_error_msg("ug_hip_dxt_encode: image larger than 4 GiB");
                return UG_HIP_EINVAL;
        }
        if (j.frames > 65535 || (brows + rows_per_group - 1) / rows_per_group > 65535) {
                ug::set_last_error_msg("ug_hip_dxt_encode: image too tall / too many frames for one launch");
                return UG_HIP_EINVAL;
        }
        const dim3 block(64, wg_rows), grid((unsigned) ((upr + 63) / 64), (unsigned) ((brows + rows_per_group - 1) / rows_per_group), (unsigned) j.frames);
#define UG_DXT_LAUNCH(MIRROR, EDGE) hipLaunchKernelGGL((dxt_encode_kernel<IN, OUT, MIRROR, AWAY, EDGE>), grid, block, 0, j.st, (const uint8_t *) j.src, \
                                                      (uint8_t *) j.dst, upr, bpr, brows, h, (uint32_t) j.pitch, j.sfs, j.dfs, j.w)
        if (edge) {
                if (mirror) UG_DXT_LAUNCH(true, true); else UG_DXT_LAUNCH(false, true);
        } else {
                if (mirror) UG_DXT_LAUNCH(true, false); else UG_DXT_LAUNCH(false, false);
        }
#undef UG_DXT_LAUNCH
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

template <int IN>
int launch_out(ug_dxt_t out, int ties, const EncodeJob &j)
{
        const bool away = ties == UG_DXT_TIES_AWAY;
        switch (out) {
        case UG_DXT1: return away ? launch<IN, UG_DXT1, true>(j) : launch<IN, UG_DXT1, false>(j);
        case UG_DXT5_YCOCG: return away ? launch<IN, UG_DXT5_YCOCG, true>(j) : launch<IN, UG_DXT5_YCOCG, false>(j);
        default: break; // UG_DXT1_YUV is rewritten to UG_DXT1 over raw UYVY by the caller
        }
        return UG_HIP_EUNSUPP;
}

} // namespace

extern "C" {

size_t ug_hip_dxt_size(ug_dxt_t out, int width, int height)
{
        if (!ug::dims_ok_signed(width, height)) return 0; // not a picture: no size (never a wrapped one)
        if (height < 0) height = -height;
        const size_t px = ((size_t) width + 3) / 4 * 4 * (((size_t) height + 3) / 4 * 4); // dxt_util.h:59-67: whole 4x4 blocks
        return out == UG_DXT1 || out == UG_DXT1_YUV ? px / 2 : px;
}

int ug_hip_dxt_encode_batch_ex(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int width, int height,
                               int src_pitch, int frames, size_t src_frame_stride, size_t dst_frame_stride, int ties,
                               ug_hip_stream_t stream)
{
        if (!ug::dims_ok_signed(width, height)) return ug::refuse_size("ug_hip_dxt_encode");
        const int ah = height < 0 ? -height : height;
        if (out == UG_DXT1_YUV) { // DXT1 over the Y,Cb,Cr samples: UYVY is its only input (dxt_glsl.cpp:104-110)
                if (in != UG_PF_UYVY && in != UG_PF_UYVY_RAW) {
                        ug::set_last_error_msg("ug_hip_dxt_encode: DXT1_YUV takes UYVY input only");
                        return UG_HIP_EUNSUPP;
                }
                in = UG_PF_UYVY_RAW;
                out = UG_DXT1;
        }
        if (ties != UG_DXT_TIES_EVEN && ties != UG_DXT_TIES_AWAY) {
                ug::set_last_error_msg("ug_hip_dxt_encode: unknown tie rule");
                return UG_HIP_EINVAL;
        }
        if (!src || !dst || width <= 0 || ah == 0 || frames < 0 || (15 & (uintptr_t) src) || (15 & (uintptr_t) dst)) {
                ug::set_last_error_msg("ug_hip_dxt_encode: bad size or alignment");
                return UG_HIP_EINVAL;
        }
        // Any width and height >= 1, as dxt_encoder_create takes them (dxt_glsl.cpp:150-160); the cuda_dxt.h-shaped entry points below keep
        // cuda_dxt.cu:745's multiples of 4.  A 4:2:2 line is made of pixel pairs.
        const bool edge = (width & 3) || (ah & 3);
        if ((width & 1) && (in == UG_PF_UYVY || in == UG_PF_UYVY_RAW || in == UG_PF_V210)) {
                ug::set_last_error_msg("ug_hip_dxt_encode: 4:2:2 input needs an even width");
                return UG_HIP_EINVAL;
        }
        if (src_pitch == 0) {
                src_pitch = ug::linesize(in, width);
        }
        const bool any_pitch = edge && (in == UG_PF_RGB || in == UG_PF_YUV444); // lines of 3 * width bytes start wherever they start
        if (src_pitch <= 0 || ((src_pitch & 3) && !any_pitch)) {
                ug::set_last_error_msg("ug_hip_dxt_encode: bad pitch / unsupported input format");
                return src_pitch <= 0 ? UG_HIP_EUNSUPP : UG_HIP_EINVAL;
        }
        if (!ug::span_ok(src_pitch, ah)) return ug::refuse_size("ug_hip_dxt_encode");
        if (frames > 1 && ((src_frame_stride & 15) || (dst_frame_stride & 15))) {
                ug::set_last_error_msg("ug_hip_dxt_encode: frame strides must be multiples of 16");
                return UG_HIP_EINVAL;
        }
        const EncodeJob j = { src, dst, width, height, src_pitch, frames, src_frame_stride, dst_frame_stride, (hipStream_t) stream };
        switch (in) {
        case UG_PF_RGB: return launch_out<UG_PF_RGB>(out, ties, j);
        case UG_PF_RGBA:
                if (src_pitch & (edge ? 3 : 15)) break;
                return launch_out<UG_PF_RGBA>(out, ties, j);
        case UG_PF_YUV444: return launch_out<UG_PF_YUV444>(out, ties, j);
        case UG_PF_UYVY:
                if (src_pitch & (edge ? 3 : 7)) break;
                return launch_out<UG_PF_UYVY>(out, ties, j);
        case UG_PF_UYVY_RAW:
                if (src_pitch & (edge ? 3 : 7)) break;
                return launch_out<UG_PF_UYVY_RAW>(out, ties, j);
        case UG_PF_V210:
                // 12 px = 32 B units; a last unit with fewer pixels is read whole, which the 128-byte line padding of v210
                // (vc_get_linesize, video_codec.c:507-521) always covers -- a caller-supplied pitch must cover it too
                if ((src_pitch & 15) || src_pitch < (width + 11) / 12 * 32) break;
                return launch_out<UG_PF_V210>(out, ties, j);
        default:
                ug::set_last_error_msg("ug_hip_dxt_encode: unsupported input format");
                return UG_HIP_EUNSUPP;
        }
        ug::set_last_error_msg("ug_hip_dxt_encode: pitch/width not aligned for this input format");
        return UG_HIP_EINVAL;
}

int ug_hip_dxt_encode_batch(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int width, int height,
                            int src_pitch, int frames, size_t src_frame_stride, size_t dst_frame_stride,
                            ug_hip_stream_t stream)
{
        return ug_hip_dxt_encode_batch_ex(in, out, src, dst, width, height, src_pitch, frames, src_frame_stride, dst_frame_stride,
                                          UG_DXT_TIES_DEFAULT, stream);
}

int ug_hip_dxt_encode(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int width, int height, int src_pitch,
                      ug_hip_stream_t stream)
{
        return ug_hip_dxt_encode_batch_ex(in, out, src, dst, width, height, src_pitch, 1, 0, 0, UG_DXT_TIES_DEFAULT, stream);
}

// cuda_dxt.h-shaped entry points: the interface of cuda_dxt.h:30-89 with its own limits -- sizes that are multiples of 4 (cuda_dxt.cu:745)
static int cuda_dxt_shaped(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int sx, int sy, ug_hip_stream_t s)
{
        if ((sx & 3) || (sy & 3)) {
                ug::set_last_error_msg("ug_hip_*_to_dxt*: width and height must be multiples of 4 (cuda_dxt.cu:745); ug_hip_dxt_encode takes any size");
                return UG_HIP_EINVAL;
        }
        return ug_hip_dxt_encode(in, out, src, dst, sx, sy, 0, s);
}
int ug_hip_rgb_to_dxt1(const void *src, void *out, int sx, int sy, ug_hip_stream_t s) { return cuda_dxt_shaped(UG_PF_RGB, UG_DXT1, src, out, sx, sy, s); }
int ug_hip_yuv_to_dxt1(const void *src, void *out, int sx, int sy, ug_hip_stream_t s) { return cuda_dxt_shaped(UG_PF_YUV444, UG_DXT1, src, out, sx, sy, s); }
int ug_hip_rgb_to_dxt6(const void *src, void *out, int sx, int sy, ug_hip_stream_t s) { return cuda_dxt_shaped(UG_PF_RGB, UG_DXT5_YCOCG, src, out, sx, sy, s); }
int ug_hip_yuv_to_dxt6(const void *src, void *out, int sx, int sy, ug_hip_stream_t s) { return cuda_dxt_shaped(UG_PF_YUV444, UG_DXT5_YCOCG, src, out, sx, sy, s); }

int ug_hip_time_dxt_encode(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int width, int height,
                           int src_pitch, int frames, size_t sfs, size_t dfs, int iters, ug_hip_stream_t stream,
                           float *ms_per_launch)
{
        if (!ms_per_launch || iters <= 0) return UG_HIP_EINVAL;
        hipStream_t st = (hipStream_t) stream;
        int rc = ug_hip_dxt_encode_batch(in, out, src, dst, width, height, src_pitch, frames, sfs, dfs, stream); // warm; refuses bad arguments before any device call
        if (rc != UG_HIP_SUCCESS) return rc;
        hipEvent_t e0, e1;
        UG_HIP_TRY(hipEventCreate(&e0));
        UG_HIP_TRY(hipEventCreate(&e1));
        {
                (void) hipEventRecord(e0, st);
                for (int i = 0; i < iters && rc == UG_HIP_SUCCESS; i++) {
                        rc = ug_hip_dxt_encode_batch(in, out, src, dst, width, height, src_pitch, frames, sfs, dfs, stream);
                }
                (void) hipEventRecord(e1, st);
                hipError_t e = hipEventSynchronize(e1);
                if (e != hipSuccess) {
                        ug::set_last_error(e, "hipEventSynchronize");
                        rc = UG_HIP_ERUNTIME;
                } else {
                        float ms = 0;
                        (void) hipEventElapsedTime(&ms, e0, e1);
                        *ms_per_launch = ms / iters;
                }
        }
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        return rc;
}

} // extern "C"

// Diagnostics: waves of this process's ug_hip_dxt_encode* launches on the current device that took the full form of an index stage
extern "C" int ug_hip_dxt_encode_stats(unsigned long long full_form_waves[2], int reset)
{
        UG_HIP_TRY(hipDeviceSynchronize());
        if (full_form_waves) {
                UG_HIP_TRY(hipMemcpyFromSymbol(full_form_waves, HIP_SYMBOL(g_full_form_waves), 2 * sizeof(unsigned long long)));
        }
        if (reset) {
                const unsigned long long zero[2] = { 0, 0 };
                UG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_full_form_waves), zero, sizeof zero));
        }
        return UG_HIP_SUCCESS;
}

// Device self-test of the encoder's exact strength reductions (div14): *mismatches must come back 0.
extern "C" int ug_hip_selftest_dxt_encode(unsigned *mismatches, ug_hip_stream_t stream)
{
        if (!mismatches) return UG_HIP_EINVAL;
        unsigned *dev = nullptr;
        UG_HIP_TRY(hipMalloc((void **) &dev, sizeof *dev));
        hipStream_t st = (hipStream_t) stream;
        hipError_t err = hipMemsetAsync(dev, 0, sizeof *dev, st);
        if (err == hipSuccess) {
                hipLaunchKernelGGL(selftest_div14_kernel, dim3((0x3f800000u >> 8) + 1), dim3(256), 0, st, dev);
                err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipMemcpyAsync(mismatches, dev, sizeof *dev, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        (void) hipFree(dev);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_selftest_dxt_encode");
                return UG_HIP_ERUNTIME;
        }
        return UG_HIP_SUCCESS;
}
