/*
 * pixfmt_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * Plain scalar C restatement of the integer pixel-format line converters and
 * packed->planar converters of UltraGrid that sit on the DXT / JPEG hot path
 * (src/pixfmt_conv.c, src/color_space.c, src/to_planar.c, src/video_codec.c).
 *
 * PINNED: every function here is checked bit-for-bit against the reference's own C
 * compiled from /root/reference (oracle/_ref/libugref.so, recipe in oracle/Makefile)
 * by tests/test_oracle_pixfmt.py, and against the committed fixtures in tests/golden/
 * that were generated from that compiled reference (tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define COMP_BASE 14 /* color_space.h:70-71 (comp_type_t == int32_t) */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------------------------------------------------------------------------- */
/* color_space.c:40-131,149-184 -- Q14 RGB<->YCbCr coefficient tables                   */
/* ---------------------------------------------------------------------------------- */
static double y_limit(int d)    { return d == 0 ? 1.0 : 219. * (1 << (d - 8)) / ((1 << d) - 1); }
static double cbcr_limit(int d) { return d == 0 ? 1.0 : 224. * (1 << (d - 8)) / ((1 << d) - 1); }
static int scaled(double x)     { return (int) ((x * (1 << COMP_BASE)) + (x > 0 ? 1. : -1.) * 0.5); }

int oracle_color_coeffs(int bt601, int depth, int o[14])
{
        if (depth != 0 && depth != 8 && depth != 10 && depth != 12 && depth != 16) {
                return -1;
        }
        const double kr = bt601 ? .299 : .212639;     /* color_space.h:73-76 */
        const double kb = bt601 ? .114 : .072192;
        const double kg = 1. - kr - kb;
        const double D  = 2. * (kr + kg);
        const double E  = 2. * (1. - kr);
        const double yl = y_limit(depth), cl = cbcr_limit(depth);
        const double B  = 1 << COMP_BASE, eps = 0.5;
        o[0]  = (int) (((kr * yl) * B) + eps);                /* y_r  color_space.c:72-74 */
        o[1]  = (int) (((kg * yl) * B) + eps);                /* y_g */
        o[2]  = (int) (((kb * yl) * B) + eps);                /* y_b */
        o[3]  = (int) (((-kr / D * cl) * B) - eps);           /* cb_r */
        o[4]  = (int) (((-kg / D * cl) * B) - eps);           /* cb_g */
        o[5]  = (int) ((((1 - kb) / D * cl) * B) + eps);      /* cb_b */
        o[6]  = (int) ((((1 - kr) / E * cl) * B) - eps);      /* cr_r */
        o[7]  = (int) (((-kg / E * cl) * B) - eps);           /* cr_g */
        o[8]  = (int) (((-kb / E * cl) * B) + eps);           /* cr_b */
        o[9]  = scaled(1. / yl);                              /* y_scale */
        o[10] = scaled((2. * (1. - kr)) / cl);                /* r_cr */
        o[11] = scaled((-kb * (2. * (kr + kg)) / kg) / cl);   /* g_cb */
        o[12] = scaled((-kr * (2. * (1. - kr)) / kg) / cl);   /* g_cr */
        o[13] = scaled((2. * (kr + kg)) / cl);                /* b_cb */
        /* struct color_coeffs stores all but b_cb as `short` (color_space.h:135-148) */
        for (int i = 0; i < 13; i++) {
                o[i] = (short) o[i];
        }
        return 0;
}

struct cfs { int y_r, y_g, y_b, cb_r, cb_g, cb_b, cr_r, cr_g, cr_b, y_scale, r_cr, g_cb, g_cr, b_cb; };
static struct cfs get_cfs(int depth)
{
        int o[14];
        struct cfs c;
        oracle_color_coeffs(0, depth, o); /* CS_DFL -> BT.709 (color_space.c:186-191) */
        memcpy(&c, o, sizeof c);
        return c;
}

/* ---------------------------------------------------------------------------------- */
/* video_codec.c:120-206 (codec_info), :507-538                                         */
/* ---------------------------------------------------------------------------------- */
static int fmt_info(int fmt, int *bs_bytes, int *bs_pix, int *h_align)
{
        switch (fmt) {
        case OPF_RGBA: *bs_bytes = 4;  *bs_pix = 1; *h_align = 1;  return 0;
        case OPF_UYVY:
        case OPF_YUYV: *bs_bytes = 4;  *bs_pix = 2; *h_align = 2;  return 0;
        case OPF_RGB:
        case OPF_BGR:  *bs_bytes = 3;  *bs_pix = 1; *h_align = 1;  return 0;
        case OPF_V210: *bs_bytes = 16; *bs_pix = 6; *h_align = 48; return 0;
        case OPF_RG48: *bs_bytes = 6;  *bs_pix = 1; *h_align = 1;  return 0;
        case OPF_I420: *bs_bytes = 3;  *bs_pix = 2; *h_align = 2;  return 0;
        }
        return -1;
}
int oracle_linesize(int width, int fmt)
{
        int bb, bp, ha;
        if (fmt_info(fmt, &bb, &bp, &ha)) return 0;
        width = (width + ha - 1) / ha * ha;
        return (width + bp - 1) / bp * bb;
}
int oracle_size(int width, int fmt)
{
        int bb, bp, ha;
        if (fmt_info(fmt, &bb, &bp, &ha)) return 0;
        return (width + bp - 1) / bp * bb;
}

/* ---------------------------------------------------------------------------------- */
/* line converters                                                                      */
/* ---------------------------------------------------------------------------------- */
static inline uint32_t ld32(const uint8_t *p) { uint32_t w; memcpy(&w, p, 4); return w; }
static inline void st32(uint8_t *p, uint32_t w) { memcpy(p, &w, 4); }

/* pixfmt_conv.c:86-130 : each 10-bit sample >> 2; 16 B -> 12 B, tail 4 / 8 B */
static void l_v210_to_uyvy(uint8_t *dst, const uint8_t *src, int dst_len)
{
        while (dst_len >= 4) {
                int n = dst_len >= 12 ? 3 : dst_len / 4; /* output words available */
                uint32_t w0 = ld32(src), w1 = ld32(src + 4), w2 = 0, w3 = 0;
                if (n >= 2) w2 = ld32(src + 8);
                if (n >= 3) w3 = ld32(src + 12);
#define S(w, k) ((((w) >> (10 * (k))) & 0x3ffu) >> 2)
                st32(dst, S(w0, 0) | S(w0, 1) << 8 | S(w0, 2) << 16 | S(w1, 0) << 24);
                if (n >= 2) st32(dst + 4, S(w1, 1) | S(w1, 2) << 8 | S(w2, 0) << 16 | S(w2, 1) << 24);
                if (n >= 3) st32(dst + 8, S(w2, 2) | S(w3, 0) << 8 | S(w3, 1) << 16 | S(w3, 2) << 24);
#undef S
                if (n < 3) break;
                src += 16; dst += 12; dst_len -= 12;
        }
}

/* pixfmt_conv.c:136-198 : swap bytes within 16-bit pairs */
static void l_yuyv_swap(uint8_t *dst, const uint8_t *src, int dst_len)
{
        for (int x = 0; x + 4 <= dst_len; x += 4) {
                dst[x] = src[x + 1]; dst[x + 1] = src[x];
                dst[x + 2] = src[x + 3]; dst[x + 3] = src[x + 2];
        }
}

/* pixfmt_conv.c:1065-1094,1102-1108 : Q14 BT.709 limited, clamp [0,255] */
static void l_uyvy_to_rgb(uint8_t *dst, const uint8_t *src, int dst_len)
{
        const struct cfs c = get_cfs(8);
        for (int x = 0; x <= dst_len - 6; x += 6) {
                int y1 = c.y_scale * (src[1] - 16);
                int y2 = c.y_scale * (src[3] - 16);
                int u = src[0] - 128, v = src[2] - 128;
                src += 4;
                *dst++ = clampi((y1 + v * c.r_cr) >> COMP_BASE, 0, 255);
                *dst++ = clampi((y1 + u * c.g_cb + v * c.g_cr) >> COMP_BASE, 0, 255);
                *dst++ = clampi((y1 + u * c.b_cb) >> COMP_BASE, 0, 255);
                *dst++ = clampi((y2 + v * c.r_cr) >> COMP_BASE, 0, 255);
                *dst++ = clampi((y2 + u * c.g_cb + v * c.g_cr) >> COMP_BASE, 0, 255);
                *dst++ = clampi((y2 + u * c.b_cb) >> COMP_BASE, 0, 255);
        }
}

/* pixfmt_conv.c:1137-1163 : double precision, truncation toward zero, clamp, shifts */
static void l_uyvy_to_rgba(uint8_t *dst, const uint8_t *src, int dst_len, int rs, int gs, int bs)
{
        uint32_t am = 0xFFFFFFFFU ^ (0xFFU << rs) ^ (0xFFU << gs) ^ (0xFFU << bs);
        for (int x = 0; x <= dst_len - 8; x += 8) {
                int u = src[0], y1 = src[1], v = src[2], y2 = src[3];
                src += 4;
                for (int k = 0; k < 2; k++) {
                        int y = k ? y2 : y1;
                        int r = 1.164 * (y - 16) + 1.793 * (v - 128);
                        int g = 1.164 * (y - 16) - 0.534 * (v - 128) - 0.213 * (u - 128);
                        int b = 1.164 * (y - 16) + 2.115 * (u - 128);
                        r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                        st32(dst, am | (uint32_t) r << rs | (uint32_t) g << gs | (uint32_t) b << bs);
                        dst += 4;
                }
        }
}

/* pixfmt_conv.c:1008-1053 : Q14; chroma = ((cb1+cb2)/2 >> 14) + 128 with C `/` truncation */
static void l_to_uyvy(uint8_t *dst, const uint8_t *src, int dst_len, int ro, int go, int bo, int ps)
{
        const struct cfs c = get_cfs(8);
        const int count = (dst_len + 3) / 4;
        for (int x = 0; x < count; x++) {
                int r = src[ro], g = src[go], b = src[bo];
                src += ps;
                int y1 = ((r * c.y_r + g * c.y_g + b * c.y_b) >> COMP_BASE) + 16;
                int u = r * c.cb_r + g * c.cb_g + b * c.cb_b;
                int v = r * c.cr_r + g * c.cr_g + b * c.cr_b;
                r = src[ro]; g = src[go]; b = src[bo];
                src += ps;
                int y2 = ((r * c.y_r + g * c.y_g + b * c.y_b) >> COMP_BASE) + 16;
                u += r * c.cb_r + g * c.cb_g + b * c.cb_b;
                v += r * c.cr_r + g * c.cr_g + b * c.cr_b;
                u = ((u / 2) >> COMP_BASE) + 128;
                v = ((v / 2) >> COMP_BASE) + 128;
                st32(dst, ((uint32_t) (y2 & 0xFF) << 24) | ((v & 0xFF) << 16) | ((y1 & 0xFF) << 8) | (u & 0xFF));
                dst += 4;
        }
}

/* pixfmt_conv.c:2884-2940 (8-bit out) and :2942-3002 (16-bit out) */
static void l_v210_to_rgb(uint8_t *dst, const uint8_t *src, int dst_len, int out16)
{
        const int idepth = out16 ? 10 : 8;
        const struct cfs c = get_cfs(idepth);
        const int y_shift = 1 << (idepth - 4), c_shift = 1 << (idepth - 1);
        const int sh = out16 ? COMP_BASE - 6 : COMP_BASE;
        const int lo = out16 ? 1 << 8 : 1, hi = out16 ? (255 << 8) - 1 : 254; /* CLAMP_FULL color_space.h:96-98 */
        const int drop = out16 ? 0 : 2;
        const int obl = out16 ? 36 : 18;
        uint16_t *d16 = (uint16_t *) (void *) dst;
        for (int x = 0; x < dst_len; x += obl) {
                uint32_t w[4];
                for (int i = 0; i < 4; i++) w[i] = ld32(src + 4 * i);
                src += 16;
#define S(wd, k) ((int) ((((wd) >> (10 * (k))) & 0x3ffu) >> drop))
                int Y[6] = { S(w[0], 1), S(w[1], 0), S(w[1], 2), S(w[2], 1), S(w[3], 0), S(w[3], 2) };
                int U[3] = { S(w[0], 0), S(w[1], 1), S(w[2], 2) };
                int V[3] = { S(w[0], 2), S(w[2], 0), S(w[3], 1) };
#undef S
                for (int i = 0; i < 6; i++) {
                        int u = U[i / 2] - c_shift, v = V[i / 2] - c_shift;
                        int y = c.y_scale * (Y[i] - y_shift);
                        int r = clampi((y + v * c.r_cr) >> sh, lo, hi);
                        int g = clampi((y + u * c.g_cb + v * c.g_cr) >> sh, lo, hi);
                        int b = clampi((y + u * c.b_cb) >> sh, lo, hi);
                        if (out16) { *d16++ = r; *d16++ = g; *d16++ = b; }
                        else       { *dst++ = r; *dst++ = g; *dst++ = b; }
                }
        }
}

/* pixfmt_conv.c:866-900 */
static void l_rgba_to_rgb(uint8_t *dst, const uint8_t *src, int dst_len)
{
        for (int x = 0; x <= dst_len - 3; x += 3) {
                *dst++ = src[0]; *dst++ = src[1]; *dst++ = src[2];
                src += 4;
        }
}
/* pixfmt_conv.c:944-990 */
static void l_rgb_to_rgba(uint8_t *dst, const uint8_t *src, int dst_len, int rs, int gs, int bs)
{
        uint32_t am = 0xFFFFFFFFU ^ (0xFFU << rs) ^ (0xFFU << gs) ^ (0xFFU << bs);
        for (int x = 0; x <= dst_len - 4; x += 4) {
                uint32_t r = src[0], g = src[1], b = src[2];
                src += 3;
                st32(dst, am | r << rs | g << gs | b << bs);
                dst += 4;
        }
}
/* pixfmt_conv.c:538-589 : default shifts = memcpy (alpha preserved), else alpha = 0xFF */
static void l_rgba_shift(uint8_t *dst, const uint8_t *src, int len, int rs, int gs, int bs)
{
        if (rs == 0 && gs == 8 && bs == 16) { memcpy(dst, src, len); return; }
        uint32_t am = 0xFFFFFFFFU ^ (0xFFU << rs) ^ (0xFFU << gs) ^ (0xFFU << bs);
        while (len >= 4) {
                uint32_t t = ld32(src);
                st32(dst, am | (t & 0xff) << rs | ((t >> 8) & 0xff) << gs | ((t >> 16) & 0xff) << bs);
                src += 4; dst += 4; len -= 4;
        }
}
/* pixfmt_conv.c:732-753 (and :2520-2527 for BGR->RGB = shifts 16,8,0) */
static void l_rgb_shift(uint8_t *dst, const uint8_t *src, int dst_len, int rs, int gs, int bs)
{
        if (rs == 0 && gs == 8 && bs == 16) { memcpy(dst, src, dst_len); return; }
        for (int x = 0; x <= dst_len - 3; x += 3) {
                uint32_t o = (uint32_t) src[0] << rs | (uint32_t) src[1] << gs | (uint32_t) src[2] << bs;
                src += 3;
                *dst++ = o & 0xff; *dst++ = (o >> 8) & 0xff; *dst++ = (o >> 16) & 0xff;
        }
}
/* pixfmt_conv.c:2581-2607 : one 32-bit word per 3 source bytes, sample << 2, pad 0 */
static void l_uyvy_to_v210(uint8_t *dst, const uint8_t *src, int dst_len)
{
        while (dst_len >= 4) {
                uint32_t a = src[0], b = src[1], c = src[2];
                src += 3;
                st32(dst, ((a << 2) & 0x3ff) | ((b << 2) & 0x3ff) << 10 | ((c << 2) & 0x3ff) << 20);
                dst += 4; dst_len -= 4;
        }
}

/* pixfmt_conv.c:3041-3125 (decoders[] + get_decoder_from_to) */
int oracle_convert_line(int in, int out, uint8_t *dst, const uint8_t *src, int dst_len,
                        int rs, int gs, int bs)
{
        if (in == out && out != OPF_RGBA && out != OPF_RGB) { memcpy(dst, src, dst_len); return 0; }
#define P(a, b) ((a) * 16 + (b))
        switch (P(in, out)) {
        case P(OPF_V210, OPF_UYVY): l_v210_to_uyvy(dst, src, dst_len); return 0;
        case P(OPF_YUYV, OPF_UYVY):
        case P(OPF_UYVY, OPF_YUYV): l_yuyv_swap(dst, src, dst_len); return 0;
        case P(OPF_UYVY, OPF_RGB):  l_uyvy_to_rgb(dst, src, dst_len); return 0;
        case P(OPF_UYVY, OPF_RGBA): l_uyvy_to_rgba(dst, src, dst_len, rs, gs, bs); return 0;
        case P(OPF_RGB,  OPF_UYVY): l_to_uyvy(dst, src, dst_len, 0, 1, 2, 3); return 0;
        case P(OPF_BGR,  OPF_UYVY): l_to_uyvy(dst, src, dst_len, 2, 1, 0, 3); return 0;
        case P(OPF_RGBA, OPF_UYVY): l_to_uyvy(dst, src, dst_len, 0, 1, 2, 4); return 0;
        case P(OPF_RG48, OPF_UYVY): l_to_uyvy(dst, src, dst_len, 1, 3, 5, 6); return 0;
        case P(OPF_V210, OPF_RGB):  l_v210_to_rgb(dst, src, dst_len, 0); return 0;
        case P(OPF_V210, OPF_RG48): l_v210_to_rgb(dst, src, dst_len, 1); return 0;
        case P(OPF_RGBA, OPF_RGB):  l_rgba_to_rgb(dst, src, dst_len); return 0;
        case P(OPF_RGB,  OPF_RGBA): l_rgb_to_rgba(dst, src, dst_len, rs, gs, bs); return 0;
        case P(OPF_RGBA, OPF_RGBA): l_rgba_shift(dst, src, dst_len, rs, gs, bs); return 0;
        case P(OPF_RGB,  OPF_RGB):  l_rgb_shift(dst, src, dst_len, rs, gs, bs); return 0;
        case P(OPF_BGR,  OPF_RGB):  l_rgb_shift(dst, src, dst_len, 16, 8, 0); return 0;
        case P(OPF_UYVY, OPF_V210): l_uyvy_to_v210(dst, src, dst_len); return 0;
        }
#undef P
        return -1;
}

/* testcard_common.c:121-129 / tools/convert.cpp:43-48 line loop */
int oracle_convert_frame(int in, int out, uint8_t *dst, const uint8_t *src, int width,
                         int height, int rs, int gs, int bs)
{
        const int sls = oracle_linesize(width, in), dls = oracle_linesize(width, out);
        const int dsz = oracle_size(width, out);
        for (int y = 0; y < height; y++) {
                if (oracle_convert_line(in, out, dst + (long) y * dls, src + (long) y * sls, dsz,
                                        rs, gs, bs)) {
                        return -1;
                }
        }
        return 0;
}

/* to_planar.c:343-378 */
void oracle_uyvy_to_i420(uint8_t *yp, int y_ls, uint8_t *up, int u_ls, uint8_t *vp, int v_ls,
                         const uint8_t *src, int width, int height)
{
        const long sls = oracle_linesize(width, OPF_UYVY);
        for (int i = 0; i < (height + 1) / 2; i++) {
                const uint8_t *in1 = src + 2L * i * sls, *in2 = in1 + sls;
                uint8_t *y1 = yp + 2L * i * y_ls, *y2 = y1 + y_ls;
                uint8_t *u = up + (long) i * u_ls, *v = vp + (long) i * v_ls;
                if (2 * i + 1 == height) { y2 = y1; in2 = in1; }
                for (int j = 0; j < width / 2; j++) {
                        *u++  = (in1[0] + in2[0] + 1) / 2;
                        *y1++ = in1[1]; *y2++ = in2[1];
                        *v++  = (in1[2] + in2[2] + 1) / 2;
                        *y1++ = in1[3]; *y2++ = in2[3];
                        in1 += 4; in2 += 4;
                }
                if (width % 2 == 1) {
                        *u++  = (in1[0] + in2[0] + 1) / 2;
                        *y1++ = in1[1]; *y2++ = in2[1];
                        *v++  = (in1[2] + in2[2] + 1) / 2;
                }
        }
}

/* One v210 group (16 bytes) -> six luma and six chroma (Cb Cr Cb Cr Cb Cr) 10-bit samples; sample n of the group's
 * twelve (U Y V Y ...) sits in word n / 3 at bit 10 * (n % 3) (the table of to_planar.c:92-103). */
static void v210_group(const uint8_t *p, uint32_t luma[6], uint32_t chroma[6])
{
        for (int s = 0; s < 6; s++) {
                luma[s] = (ld32(p + 4 * ((2 * s + 1) / 3)) >> (10 * ((2 * s + 1) % 3))) & 0x3ffu;
                chroma[s] = (ld32(p + 4 * ((2 * s) / 3)) >> (10 * ((2 * s) % 3))) & 0x3ffu;
        }
}

/* to_planar.c:64-155 for any geometry, the writes in the reference's order (they overlap when a line size is shorter than
 * roundup6(width) samples, so the order is part of the result):
 *   line pairs with more than two lines left (:85-86): (width + 5) / 6 whole groups, each group even line, odd line, chroma (:113-132);
 *   the last one or two lines (:87-89): width / 6 groups, an odd last line converted against itself (:80-83, the odd luma
 *   line goes to a scratch buffer), then the width % 6 samples behind them copied (:139-150) from `dst - out_linesize[]`, which on
 *   a uint16_t pointer is TWO lines above -- the luma of both lines from line y - 2, the chroma from chroma line y / 2 - 2.
 * y_ls / uv_ls in bytes.  With width % 6 != 0 and fewer than 5 lines the reference reads in front of the planes: not restated
 * (returns without writing the margin). */
void oracle_v210_to_p010le(uint16_t *yp, int y_ls, uint16_t *uvp, int uv_ls,
                           const uint8_t *src, int width, int height)
{
        const long sls = oracle_linesize(width, OPF_V210);
        const long lw = y_ls / 2, lc = uv_ls / 2;
        for (int y = 0; y < height; y += 2) {
                const int left = height - y;
                const uint8_t *s0 = src + y * sls, *s1 = left == 1 ? s0 : s0 + sls;
                uint16_t *d0 = yp + y * lw, *d1 = yp + (y + 1) * lw, *dc = uvp + (y / 2) * lc;
                const int groups = left > 2 ? (width + 5) / 6 : width / 6;
                for (int g = 0; g < groups; g++) {
                        uint32_t ya[6], yb[6], ca[6], cb[6];
                        v210_group(s0 + 16 * g, ya, ca);
                        v210_group(s1 + 16 * g, yb, cb);
                        for (int i = 0; i < 6; i++) d0[6 * g + i] = (uint16_t) (ya[i] << 6);
                        if (left != 1) for (int i = 0; i < 6; i++) d1[6 * g + i] = (uint16_t) (yb[i] << 6);
                        for (int i = 0; i < 6; i++) dc[6 * g + i] = (uint16_t) (((ca[i] + cb[i]) / 2) << 6);
                }
                if (left <= 2 && width % 6 != 0 && height >= 5) {
                        for (int x = 6 * groups; x < width; x++) d0[x] = d0[x - 2 * lw];
                        if (left == 2) for (int x = 6 * groups; x < width; x++) d1[x] = d0[x - 2 * lw];
                        for (int x = 6 * groups; x < width; x++) dc[x] = dc[x - 2 * lc];
                }
        }
}

/* vc_deinterlace (src/video_codec.c:597-664) as the reference's x86-64 build computes it (its SSE2 bodies, :624-720; the plain C loop at
 * :606-613 is never compiled where __SSE2__ is defined): IN PLACE, 16-byte column by 16-byte column, down the lines with pavgb
 * ((a + b + 1) >> 1):
 *     x0 = line 0, x1 = line 1;  for (j = 0; j < lines - 4; j += 2) {
 *         x2 = line j+2;  x0 = avg(avg(x0, x2), x1);  x1 = line j+3;  line j+1 = x0;  x0 = avg(avg(x0, x1), x2);  line j+2 = x0;  }
 * -- a recursive blend: every output feeds the next one; line 0 and the last two or three lines are left as they are.  A column whose 16
 * bytes reach past the end of the line (src_linesize % 16 != 0) takes its last bytes from the BEGINNING of the next line, which column 0
 * has filtered already (the columns are processed one after the other): restated literally, in that order.
 * Valid for src_linesize >= 16: the bytes of a column are then 16 different memory columns and may be walked one after the other.  Below
 * that the reference's 16-byte vectors overlap themselves from line to line (and are stored past the frame below 6 bytes); that is not
 * restated -- oracle/pyoracle.py refuses it, the library returns UG_HIP_EINVAL (tests/test_deinterlace.py). */
void oracle_deinterlace_blend(uint8_t *src, long src_linesize, int lines)
{
        for (long i = 0; i < src_linesize; i += 16) {
                for (int b = 0; b < 16; b++) {
                        uint8_t *col = src + i + b;
                        unsigned x0 = col[0], x1 = col[src_linesize], x2;
                        for (int j = 0; j < lines - 4; j += 2) {
                                x2 = col[(long) (j + 2) * src_linesize];
                                x0 = (x0 + x2 + 1) >> 1;
                                x0 = (x0 + x1 + 1) >> 1;
                                x1 = col[(long) (j + 3) * src_linesize];
                                col[(long) (j + 1) * src_linesize] = (uint8_t) x0;
                                x0 = (x0 + x1 + 1) >> 1;
                                x0 = (x0 + x2 + 1) >> 1;
                                col[(long) (j + 2) * src_linesize] = (uint8_t) x0;
                        }
                }
        }
}
