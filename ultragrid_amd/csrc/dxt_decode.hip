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
// across lanes.  The 8-entry luma table and the 4-entry (Co,Cg) palette of a block live in LDS so that the
// per-pixel lookups are one ds_read each instead of a chain of 64-bit selects.  HBM-bound:
// algorithmic bytes per pixel = 1 (DXT5) or 0.5 (DXT1) read + 2 (UYVY) / 3 (RGB) / 4 (RGBA) written.
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
        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
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
        uint8_t *dst;
        long pitch;
        int rs, gs, bs;
};

// store one decoded row (4 pixels, packed R | G<<8 | B<<16) of block column bx
template <int OUT, bool AWAY>
__device__ __forceinline__ void store_row(const OutArgs &o, int y, int bx, const uint32_t (&px)[4], const float *unorm)
{
        uint8_t *row = o.dst + (long) y * o.pitch;
        if (OUT == UG_PF_RGBA) {
                const uint32_t am = 0xFFFFFFFFu ^ (0xFFu << o.rs) ^ (0xFFu << o.gs) ^ (0xFFu << o.bs);
                uint32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        v[i] = am | (px[i] & 0xff) << o.rs | ((px[i] >> 8) & 0xff) << o.gs | ((px[i] >> 16) & 0xff) << o.bs;
                }
                ((uint4 *) row)[bx] = make_uint4(v[0], v[1], v[2], v[3]);
        } else if (OUT == UG_PF_RGB || OUT == UG_PF_BGR) {
                uint32_t p[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        p[i] = OUT == UG_PF_RGB ? px[i] : ((px[i] & 0xff) << 16 | (px[i] & 0xff00) | (px[i] >> 16));
                }
                uint32_t *d = (uint32_t *) row + 3 * bx;
                d[0] = p[0] | p[1] << 24;
                d[1] = (p[1] >> 8) | p[2] << 16;
                d[2] = (p[2] >> 16) | p[3] << 8;
        } else { // UYVY
                ((uint2 *) row)[bx] = make_uint2(rgb_pair_to_uyvy<AWAY>(px[0], px[1], unorm), rgb_pair_to_uyvy<AWAY>(px[2], px[3], unorm));
        }
}

template <int OUT, bool AWAY>
__global__ __launch_bounds__(256) void dxt5ycocg_decode_kernel(const uint4 *__restrict__ src, OutArgs o, int bw, int bh)
{
        // per-lane tables, entry-major so that a wave's accesses to one entry are contiguous
        __shared__ double lds_a[8][256];
        __shared__ double lds_co[4][256], lds_cg[4][256];
        __shared__ float unorm[OUT == UG_PF_UYVY ? 256 : 1];
        fill_unorm<OUT>(unorm);
        const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
        const int t = threadIdx.y * 64 + threadIdx.x;
        if (bx >= bw || by >= bh) return;
        const uint4 q = src[(long) by * bw + bx];
        unsigned long long ac = (unsigned long long) q.x | (unsigned long long) q.y << 32;
        unsigned long long cc = (unsigned long long) q.z | (unsigned long long) q.w << 32;

        // dxt62tga.c:60-62 alpha endpoints, :36-58 the two interpolation modes
        const double a0 = div_const<255>((double) (ac & 0xFF)), a1 = div_const<255>((double) ((ac >> 8) & 0xFF));
        lds_a[0][t] = a0;
        lds_a[1][t] = a1;
        if (a0 > a1) {
#pragma unroll
                for (int k = 2; k < 8; k++) lds_a[k][t] = div_const<7>((double) (8 - k) * a0 + (double) (k - 1) * a1);
        } else {
#pragma unroll
                for (int k = 2; k < 6; k++) lds_a[k][t] = div_const<5>((double) (6 - k) * a0 + (double) (k - 1) * a1);
                lds_a[6][t] = 0.0;
                lds_a[7][t] = 1.0;
        }
        // dxt62tga.c:63-74 colour endpoints and the two thirds; :24-27 per-entry scale / Co / Cg
        double r[4], g[4], b[4];
        b[0] = div_const<31>((double) (cc & 0x1F));         g[0] = div_const<63>((double) ((cc >> 5) & 0x3F));  r[0] = div_const<31>((double) ((cc >> 11) & 0x1F));
        b[1] = div_const<31>((double) ((cc >> 16) & 0x1F)); g[1] = div_const<63>((double) ((cc >> 21) & 0x3F)); r[1] = div_const<31>((double) ((cc >> 27) & 0x1F));
        b[2] = div_const<3>(2.0 * b[0] + b[1]); g[2] = div_const<3>(2.0 * g[0] + g[1]); r[2] = div_const<3>(2.0 * r[0] + r[1]);
        b[3] = div_const<3>(b[0] + 2.0 * b[1]); g[3] = div_const<3>(g[0] + 2.0 * g[1]); r[3] = div_const<3>(r[0] + 2.0 * r[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
                const double scale = 1.0 / (31.875 * b[k] + 1.0);
                lds_co[k][t] = (r[k] - 5.01960814E-01) * scale;
                lds_cg[k][t] = (g[k] - 5.01960814E-01) * scale;
        }
        ac >>= 16;
        cc >>= 32;
        // (LDS traffic is lane-private: no barrier, a wave's own ds ops are ordered)
#pragma unroll
        for (int y = 0; y < 4; y++) {
                uint32_t px[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                        const int ai = (int) (ac & 7), ci = (int) (cc & 3);
                        ac >>= 3;
                        cc >>= 2;
                        const double a = lds_a[ai][t], Co = lds_co[ci][t], Cg = lds_cg[ci][t];
                        const uint32_t R = clamp8(((a + Co) - Cg) * 255.0);
                        const uint32_t G = clamp8((a + Cg) * 255.0);
                        const uint32_t B = clamp8(((a - Co) - Cg) * 255.0);
                        px[x] = R | G << 8 | B << 16;
                }
                store_row<OUT, AWAY>(o, 4 * by + y, bx, px, unorm);
        }
}

// YUV = true: DXT1_YUV -- the palette holds Y, Cb, Cr and goes through the display matrix of
// dxt_compress/display_dxt1_yuv_fp.glsl:21-32 (fp32, one operation per shader operation) before the 8-bit write.
template <int OUT, bool YUV, bool AWAY>
__global__ __launch_bounds__(256) void dxt1_decode_kernel(const uint2 *__restrict__ src, OutArgs o, int bw, int bh)
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
                float hu[4], hv[4];
                uint32_t y4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        const float r = unorm[pal[k] & 0xff], g = unorm[(pal[k] >> 8) & 0xff], b = unorm[(pal[k] >> 16) & 0xff];
                        const float yy = (float) (1.0 / 16.0) + ((r * 0.2126f + g * 0.7152f) + b * 0.0722f) * 0.8588f;
                        const float uu = 0.5f + ((-r * 0.1145f - g * 0.3854f) + b * 0.5f) * 0.8784f;
                        const float vv = 0.5f + ((r * 0.5f - g * 0.4541f) - b * 0.0458f) * 0.8784f;
                        y4 |= (uint32_t) unorm8_out<AWAY>(yy) << (8 * k);
                        hu[k] = uu * 0.5f;
                        hv[k] = vv * 0.5f;
                }
#pragma unroll
                for (int y = 0; y < 4; y++) {
                        uint32_t word[2];
#pragma unroll
                        for (int p = 0; p < 2; p++) {
                                const uint32_t ca = idx & 3, cb = (idx >> 2) & 3;
                                idx >>= 4;
                                const float ua = (ca & 2) ? ((ca & 1) ? hu[3] : hu[2]) : ((ca & 1) ? hu[1] : hu[0]);
                                const float ub = (cb & 2) ? ((cb & 1) ? hu[3] : hu[2]) : ((cb & 1) ? hu[1] : hu[0]);
                                const float va = (ca & 2) ? ((ca & 1) ? hv[3] : hv[2]) : ((ca & 1) ? hv[1] : hv[0]);
                                const float vb = (cb & 2) ? ((cb & 1) ? hv[3] : hv[2]) : ((cb & 1) ? hv[1] : hv[0]);
                                word[p] = (uint32_t) unorm8_out<AWAY>(ua + ub) | ((y4 >> (8 * ca)) & 0xff) << 8 | (uint32_t) unorm8_out<AWAY>(va + vb) << 16 |
                                          ((y4 >> (8 * cb)) & 0xff) << 24;
                        }
                        ((uint2 *) (o.dst + (long) (4 * by + y) * o.pitch))[bx] = make_uint2(word[0], word[1]);
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

template <int OUT, bool AWAY>
int launch_decode_t(ug_dxt_t in, const void *src, const OutArgs &o, int w, int h, hipStream_t st)
{
        const int bw = w / 4, bh = h / 4;
        const dim3 block(64, 4), grid((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4));
        if (in == UG_DXT5_YCOCG) {
                hipLaunchKernelGGL((dxt5ycocg_decode_kernel<OUT, AWAY>), grid, block, 0, st, (const uint4 *) src, o, bw, bh);
        } else if (in == UG_DXT1_YUV) {
                hipLaunchKernelGGL((dxt1_decode_kernel<OUT, true, AWAY>), grid, block, 0, st, (const uint2 *) src, o, bw, bh);
        } else {
                hipLaunchKernelGGL((dxt1_decode_kernel<OUT, false, AWAY>), grid, block, 0, st, (const uint2 *) src, o, bw, bh);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
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
        if (!src_dev || !dst_dev || width <= 0 || height <= 0 || (width & 3) || (height & 3) || (15 & (uintptr_t) dst_dev) ||
            ((in == UG_DXT5_YCOCG ? 15 : 7) & (uintptr_t) src_dev) || (height / 4 + 3) / 4 > 65535) {
                ug::set_last_error_msg("ug_hip_dxt_decode: bad size or alignment");
                return UG_HIP_EINVAL;
        }
        if (in != UG_DXT1 && in != UG_DXT1_YUV && in != UG_DXT5_YCOCG) {
                ug::set_last_error_msg("ug_hip_dxt_decode: unknown compressed format");
                return UG_HIP_EUNSUPP;
        }
        if (dst_pitch == 0) {
                dst_pitch = ug::linesize(out, width);
        }
        OutArgs o = { (uint8_t *) dst_dev, dst_pitch, rshift, gshift, bshift };
        hipStream_t st = (hipStream_t) stream;
        switch (out) {
        case UG_PF_RGBA:
                if (dst_pitch & 15) break;
                return launch_decode<UG_PF_RGBA>(in, ties, src_dev, o, width, height, st);
        case UG_PF_RGB:
                if (dst_pitch & 3) break;
                return launch_decode<UG_PF_RGB>(in, ties, src_dev, o, width, height, st);
        case UG_PF_BGR:
                if (dst_pitch & 3) break;
                return launch_decode<UG_PF_BGR>(in, ties, src_dev, o, width, height, st);
        case UG_PF_UYVY:
                if (dst_pitch & 7) break;
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
