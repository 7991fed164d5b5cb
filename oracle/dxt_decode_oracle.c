/*
 * dxt_decode_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU DXT5-YCoCg and DXT1 block decoders, used (a) as the sanity gate for the
 * "parity unpinned" DXT encoder oracle (decode what it produced, check PSNR against
 * the source) and (b) as the oracle of the decompress-side kernels.
 *
 * DXT5-YCoCg follows the reference's CPU decoder cuda_dxt/dxt62tga.c:24-106
 * (fp64; alpha = Y with 8-level interpolation, colour = (Co, Cg, scale) with palette
 * thirds, scale = 1/(31.875*b + 1), R = Y+Co-Cg, G = Y+Cg, B = Y-Co-Cg,
 * clamp(x*255 + 0.5)); output here is RGB order rather than the tool's BGR.
 * DXT1 is standard S3TC 4-colour mode (c0 > c1 is guaranteed by the encoder,
 * compress_dxt1_fp.glsl:116-123); colour expansion /31, /63 as dxt62tga.c:63-68.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static uint8_t clamp8(double s)
{
        int is = (int) (s + 0.5);
        return is > 255 ? 255 : (is < 0 ? 0 : is);
}

static double alpha_decode(double a0, double a1, int idx)
{
        if (a0 > a1) {
                switch (idx) {
                case 0: return a0;
                case 1: return a1;
                default: return ((8 - idx) * a0 + (idx - 1) * a1) / 7.0;
                }
        }
        switch (idx) {
        case 0: return a0;
        case 1: return a1;
        case 6: return 0.0;
        case 7: return 1.0;
        default: return ((6 - idx) * a0 + (idx - 1) * a1) / 5.0;
        }
}

void oracle_dxt5ycocg_decode_rgb(const uint8_t *src, uint8_t *dst, int w, int h)
{
        for (int by = 0; by < (h + 3) / 4; by++) {
                for (int bx = 0; bx < (w + 3) / 4; bx++) {
                        uint64_t ac, cc;
                        memcpy(&ac, src, 8);
                        memcpy(&cc, src + 8, 8);
                        src += 16;
                        const double a0 = (ac & 0xFF) / 255.0, a1 = ((ac >> 8) & 0xFF) / 255.0;
                        double r[4], g[4], b[4];
                        b[0] = (cc & 0x1F) / 31.0;
                        g[0] = ((cc >> 5) & 0x3F) / 63.0;
                        r[0] = ((cc >> 11) & 0x1F) / 31.0;
                        b[1] = ((cc >> 16) & 0x1F) / 31.0;
                        g[1] = ((cc >> 21) & 0x3F) / 63.0;
                        r[1] = ((cc >> 27) & 0x1F) / 31.0;
                        b[2] = (2.0 * b[0] + b[1]) / 3.0; g[2] = (2.0 * g[0] + g[1]) / 3.0; r[2] = (2.0 * r[0] + r[1]) / 3.0;
                        b[3] = (b[0] + 2.0 * b[1]) / 3.0; g[3] = (g[0] + 2.0 * g[1]) / 3.0; r[3] = (r[0] + 2.0 * r[1]) / 3.0;
                        ac >>= 16;
                        cc >>= 32;
                        for (int y = 0; y < 4; y++) {
                                for (int x = 0; x < 4; x++) {
                                        const double a = alpha_decode(a0, a1, ac & 7);
                                        const int ci = cc & 3;
                                        ac >>= 3;
                                        cc >>= 2;
                                        const double scale = 1.0 / (31.875 * b[ci] + 1.0);
                                        const double Co = (r[ci] - 5.01960814E-01) * scale;
                                        const double Cg = (g[ci] - 5.01960814E-01) * scale;
                                        if (4 * by + y >= h || 4 * bx + x >= w) continue; /* texels past the picture are not shown */
                                        uint8_t *o = dst + 3 * ((long) (4 * by + y) * w + 4 * bx + x);
                                        o[0] = clamp8(((a + Co) - Cg) * 255.0);
                                        o[1] = clamp8((a + Cg) * 255.0);
                                        o[2] = clamp8(((a - Co) - Cg) * 255.0);
                                }
                        }
                }
        }
}

void oracle_dxt1_decode_rgb(const uint8_t *src, uint8_t *dst, int w, int h)
{
        for (int by = 0; by < (h + 3) / 4; by++) {
                for (int bx = 0; bx < (w + 3) / 4; bx++) {
                        uint16_t c0, c1;
                        uint32_t idx;
                        memcpy(&c0, src, 2);
                        memcpy(&c1, src + 2, 2);
                        memcpy(&idx, src + 4, 4);
                        src += 8;
                        double p[4][3];
                        p[0][0] = ((c0 >> 11) & 0x1F) / 31.0; p[0][1] = ((c0 >> 5) & 0x3F) / 63.0; p[0][2] = (c0 & 0x1F) / 31.0;
                        p[1][0] = ((c1 >> 11) & 0x1F) / 31.0; p[1][1] = ((c1 >> 5) & 0x3F) / 63.0; p[1][2] = (c1 & 0x1F) / 31.0;
                        for (int k = 0; k < 3; k++) {
                                if (c0 > c1) {
                                        p[2][k] = (2.0 * p[0][k] + p[1][k]) / 3.0;
                                        p[3][k] = (p[0][k] + 2.0 * p[1][k]) / 3.0;
                                } else { /* 3-colour + transparent-black mode */
                                        p[2][k] = (p[0][k] + p[1][k]) / 2.0;
                                        p[3][k] = 0.0;
                                }
                        }
                        for (int i = 0; i < 16; i++) {
                                const int ci = (idx >> (2 * i)) & 3;
                                if (4 * by + i / 4 >= h || 4 * bx + i % 4 >= w) continue;
                                uint8_t *o = dst + 3 * ((long) (4 * by + i / 4) * w + 4 * bx + i % 4);
                                for (int k = 0; k < 3; k++) o[k] = clamp8(p[ci][k] * 255.0);
                        }
                }
        }
}

static inline uint8_t unorm8_out(float x);

/* DXT1_YUV (-c RTDXT:DXT1_YUV): the DXT1 palette holds Y, Cb, Cr; the receiver renders it through
 * dxt_compress/display_dxt1_yuv_fp.glsl:21-32 (fixed-function S3TC fetch, then the colour matrix).  The palette is the
 * DXT1 one above (in double, rounded to fp32 where the sampler hands it to the shader), every shader operation is one fp32
 * operation, the framebuffer write is floorf(clamp01(x) * 255 + 0.5).  GL leaves the S3TC interpolation precision and the
 * unorm conversion to the implementation: parity unpinned, like the rest of the receiver side. */
void oracle_dxt1yuv_decode_rgb(const uint8_t *src, uint8_t *dst, int w, int h)
{
        for (int by = 0; by < (h + 3) / 4; by++) {
                for (int bx = 0; bx < (w + 3) / 4; bx++) {
                        uint16_t c0, c1;
                        uint32_t idx;
                        memcpy(&c0, src, 2);
                        memcpy(&c1, src + 2, 2);
                        memcpy(&idx, src + 4, 4);
                        src += 8;
                        double p[4][3];
                        p[0][0] = ((c0 >> 11) & 0x1F) / 31.0; p[0][1] = ((c0 >> 5) & 0x3F) / 63.0; p[0][2] = (c0 & 0x1F) / 31.0;
                        p[1][0] = ((c1 >> 11) & 0x1F) / 31.0; p[1][1] = ((c1 >> 5) & 0x3F) / 63.0; p[1][2] = (c1 & 0x1F) / 31.0;
                        for (int k = 0; k < 3; k++) {
                                if (c0 > c1) {
                                        p[2][k] = (2.0 * p[0][k] + p[1][k]) / 3.0;
                                        p[3][k] = (p[0][k] + 2.0 * p[1][k]) / 3.0;
                                } else {
                                        p[2][k] = (p[0][k] + p[1][k]) / 2.0;
                                        p[3][k] = 0.0;
                                }
                        }
                        uint8_t pal[4][3];
                        for (int k = 0; k < 4; k++) {
                                const float col0 = (float) p[k][0], col1 = (float) p[k][1], col2 = (float) p[k][2];
                                float t;
                                t = col0 - 0.0625f; const float Y = 1.1643f * t;
                                t = col1 - 0.5f;    const float U = 1.1384f * t;
                                t = col2 - 0.5f;    const float V = 1.1384f * t;
                                float G = 0.39173f * U; G = Y - G; t = 0.81290f * V; G = G - t;
                                float B = 2.017f * U; B = Y + B;
                                float R = 1.5958f * V; R = Y + R;
                                pal[k][0] = unorm8_out(R); pal[k][1] = unorm8_out(G); pal[k][2] = unorm8_out(B);
                        }
                        for (int i = 0; i < 16; i++) {
                                const int ci = (idx >> (2 * i)) & 3;
                                if (4 * by + i / 4 >= h || 4 * bx + i % 4 >= w) continue;
                                memcpy(dst + 3 * ((long) (4 * by + i / 4) * w + 4 * bx + i % 4), pal[ci], 3);
                        }
                }
        }
}

/* ------------------------------------------------------------------------------------------------
 * Generic frame decode used as the oracle of the decompress-side kernels.
 *   in_fmt : ORACLE_OUT_DXT1 / ORACLE_OUT_DXT1_YUV / ORACLE_OUT_DXT5YCOCG (same ids as the encoder side)
 *   out_fmt: OPF_RGB, OPF_BGR, OPF_RGBA (alpha 0xFF, component shifts rs/gs/bs as decoder_t), OPF_UYVY
 * RGB values are the 8-bit results above (cuda_dxt/dxt62tga.c arithmetic).  UYVY follows the reference's
 * RGBA->4:2:2 pass (dxt_compress/rgba_to_yuv422.glsl:27-46) applied to those 8-bit texels, every shader
 * operation one fp32 operation, texel fetch = v / 255.0f, output = floorf(clamp01(x) * 255 + 0.5)
 * (GL's unorm8 conversions are implementation-defined to that extent: parity unpinned).
 * ---------------------------------------------------------------------------------------------- */
/* float -> unorm8 framebuffer write.  GL rounds to nearest and leaves exact .5 ties to the implementation: default here = ties to
 * even, what Mesa llvmpipe does when it runs rgba_to_yuv422.glsl (pinned byte for byte, tests/test_oracle_dxt.py);
 * oracle_set_unorm_ties_even(0) / oracle_set_ties(1) = half up, floor(x * 255 + 0.5) */
static int g_unorm_ties_even = 1;
void oracle_set_unorm_ties_even(int on) { g_unorm_ties_even = on; }
static inline uint8_t unorm8_out(float x)
{
        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        float t = x * 255.0f;
        if (g_unorm_ties_even) {
                return (uint8_t) (int) rintf(t);
        }
        t = t + 0.5f;
        return (uint8_t) (int) t; /* t >= 0: truncation == floor */
}

static void rgb_pair_to_uyvy(const uint8_t *p1, const uint8_t *p2, uint8_t *out)
{
        float c[2][3], yuv[2][3];
        for (int k = 0; k < 3; k++) {
                c[0][k] = (float) p1[k] / 255.0f;
                c[1][k] = (float) p2[k] / 255.0f;
        }
        for (int i = 0; i < 2; i++) {
                const float r = c[i][0], g = c[i][1], b = c[i][2];
                float t;
                t = r * 0.2126f; t = t + g * 0.7152f; t = t + b * 0.0722f; t = t * 0.8588f;
                yuv[i][0] = (float) (1.0 / 16.0) + t;
                t = -r * 0.1145f; t = t - g * 0.3854f; t = t + b * 0.5f; t = t * 0.8784f;
                yuv[i][1] = 0.5f + t;
                t = r * 0.5f; t = t - g * 0.4541f; t = t - b * 0.0458f; t = t * 0.8784f;
                yuv[i][2] = 0.5f + t;
        }
        /* mix(a, b, 0.5) = a * (1 - 0.5) + b * 0.5 */
        const float U = yuv[0][1] * 0.5f + yuv[1][1] * 0.5f;
        const float V = yuv[0][2] * 0.5f + yuv[1][2] * 0.5f;
        out[0] = unorm8_out(U);
        out[1] = unorm8_out(yuv[0][0]);
        out[2] = unorm8_out(V);
        out[3] = unorm8_out(yuv[1][0]);
}

int oracle_dxt_decode(int in_fmt, int out_fmt, const uint8_t *src, uint8_t *dst, int w, int h, long dst_pitch,
                      int rs, int gs, int bs)
{
        /* any size: the stream holds (w+3)/4 x (h+3)/4 blocks (dxt_util.h:59-67), of which the w x h picture is shown
         * (dxt_decoder.c:146-149,368-389: the S3TC texture is created w x h, the upload covers the rounded-up block grid) */
        if (w <= 0 || h <= 0 || (out_fmt == OPF_UYVY && (w & 1))) {
                return -1;
        }
        uint8_t *rgb = (uint8_t *) malloc((size_t) w * h * 3);
        if (!rgb) {
                return -1;
        }
        if (in_fmt == ORACLE_OUT_DXT5YCOCG) {
                oracle_dxt5ycocg_decode_rgb(src, rgb, w, h);
        } else if (in_fmt == ORACLE_OUT_DXT1) {
                oracle_dxt1_decode_rgb(src, rgb, w, h);
        } else if (in_fmt == ORACLE_OUT_DXT1_YUV) {
                oracle_dxt1yuv_decode_rgb(src, rgb, w, h);
        } else {
                free(rgb);
                return -1;
        }
        int rc = 0;
        for (int y = 0; y < h && rc == 0; y++) {
                const uint8_t *s = rgb + (size_t) y * w * 3;
                uint8_t *d = dst + (long) y * dst_pitch;
                switch (out_fmt) {
                case OPF_RGB: memcpy(d, s, (size_t) w * 3); break;
                case OPF_BGR:
                        for (int x = 0; x < w; x++) { d[3 * x] = s[3 * x + 2]; d[3 * x + 1] = s[3 * x + 1]; d[3 * x + 2] = s[3 * x]; }
                        break;
                case OPF_RGBA: {
                        const uint32_t am = 0xFFFFFFFFU ^ (0xFFU << rs) ^ (0xFFU << gs) ^ (0xFFU << bs);
                        for (int x = 0; x < w; x++) {
                                const uint32_t v = am | (uint32_t) s[3 * x] << rs | (uint32_t) s[3 * x + 1] << gs | (uint32_t) s[3 * x + 2] << bs;
                                memcpy(d + 4 * x, &v, 4);
                        }
                        break;
                }
                case OPF_UYVY:
                        for (int x = 0; x < w; x += 2) rgb_pair_to_uyvy(s + 3 * x, s + 3 * x + 3, d + 2 * x);
                        break;
                default: rc = -1;
                }
        }
        free(rgb);
        return rc;
}
