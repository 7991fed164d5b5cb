/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * CPU specification of the JPEG 8x8 forward-DCT + quantisation stage.
 *
 * PARITY UNPINNED.  UltraGrid does not contain this code: `-c gpujpeg` / `-c jpeg`
 * call the external library CESNET/GPUJPEG (configure.ac:2631-2675 requires
 * `libgpujpeg >= 0.14.0`; ext-deps/bootstrap_gpujpeg.sh clones an unpinned HEAD; call
 * sites src/video_compress/gpujpeg.cpp:279-353,617-631).  The library is neither
 * vendored under /root/reference nor installed in this image, and the only reference
 * test at that boundary (test/gpujpeg_test.cpp:68-106: flat 127 frame, |diff| <= 1 after
 * a full encode/decode round trip) pins no coefficient.  This file therefore *defines*
 * the stage from the published algorithms:
 *
 *   - samples are level-shifted by -128 (ITU-T T.81 A.3.1);
 *   - 2-D DCT-II by the Arai-Agui-Nakajima (AAN) factorisation in fp32, rows then
 *     columns, 5 multiplies + 29 adds per 1-D transform (Arai, Agui, Nakajima, Trans.
 *     IEICE E-71(11), 1988; the same factorisation GPUJPEG and IJG's float DCT use),
 *     every operation one IEEE binary32 operation, no FMA (-ffp-contract=off);
 *   - the AAN output is scaled by 8*aan[u]*aan[v]; that factor is folded into the fp32
 *     reciprocal quantiser  div[i] = (float)(1.0 / (q[i] * aan[row] * aan[col] * 8.0));
 *   - quantised value = (int16) rintf(coef * div)   (round-half-to-even);
 *   - quantiser tables: T.81 Annex K.1 (luma) / K.2 (chroma) scaled by the IJG quality
 *     rule  s = q < 50 ? 5000/q : 200 - 2q ;  t = clamp((base*s + 50)/100, 1, 255);
 *   - coefficients are emitted in zig-zag order (T.81 Figure A.6).
 *
 * What CAN be pinned is pinned -- to IJG's float DCT pipeline as the image's libjpeg-turbo 2.1.2 executes it (tests/libjpeg_float.py
 * drives the library through ctypes with dct_method = JDCT_FLOAT):
 *   - tests/test_oracle_jpeg.py::test_fdct_is_ijg_float_dct_bit_for_bit: the forward DCT below gives the same BITS as jpeg_fdct_float
 *     (jfdctflt.c, exported symbol) on the level-shifted samples of every block; the quality rule equals jpeg_quality_scaling +
 *     jpeg_add_quant_table's baseline clamp for q = 1 .. 100;
 *   - ::test_plane_to_scan_equals_libjpeg_turbo_float_pipeline: a grey plane compressed by libjpeg-turbo (float DCT + its float
 *     quantiser, standard Huffman tables) and this file's coefficients coded by the test writer give the same entropy-coded bytes -- i.e.
 *     the same quantised coefficients, rounding of the reciprocal quantiser included (libjpeg-turbo's SSE2 quantiser converts with
 *     cvtps2dq, round-to-nearest-even = rintf; IJG's plain C would round (int)(x + 16384.5) - 16384, different on exact ties) -- for 240
 *     cases, picture sizes that are no multiple of 8 and noise at q = 100 included;
 *   - tests/test_gpu_jpeg.py::test_streams_equal_libjpeg_turbo_with_its_float_dct: the PRODUCT's RGB 4:4:4, UYVY 4:2:2 and 4:2:0 streams
 *     against libjpeg-turbo on the same samples (the 4:2:x planes through jpeg_write_raw_data), same entropy-coded bytes.
 * That ties the restatement, and the product, to a published executable implementation of the formulation restated here; towards
 * UltraGrid's libgpujpeg itself the stage stays unpinned (one known freedom: blocks that lie wholly outside the picture -- MCU padding --
 * are DCTs of the replicated edge here, "dummy" blocks (neighbour's DC, no AC) in libjpeg; they are never displayed).
 *
 * Tolerance contract (tests/test_jpeg_*.py):
 *   (a) HIP kernel vs this file: unquantised fp32 coefficients identical (0 ULP; the
 *       north-star bound is 1 ULP) and quantised int16 output identical -- same op order,
 *       no contraction on either side;
 *   (b) this file vs an fp64 scipy.fft.dctn reference: unquantised coefficients within
 *       2e-3 absolute (|coef| <= 1.6e4, i.e. fp32 round-off of the butterfly);
 *       |quantised - round(DCT64/q)| <= 1 on every coefficient, != 0 on < 0.1 % of them, and
 *       only where the exact quotient lies within 1e-3 of a rounding tie (multiply-by-
 *       reciprocal cannot preserve exact .5 ties, which are common for DC = sum/8).
 */
#include <math.h>
#include <stdint.h>

#include "oracle.h"

const uint8_t oracle_jpeg_zigzag[64] = {
         0,  1,  8, 16,  9,  2,  3, 10,
        17, 24, 32, 25, 18, 11,  4,  5,
        12, 19, 26, 33, 40, 48, 41, 34,
        27, 20, 13,  6,  7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36,
        29, 22, 15, 23, 30, 37, 44, 51,
        58, 59, 52, 45, 38, 31, 39, 46,
        53, 60, 61, 54, 47, 55, 62, 63,
};

/* T.81 Annex K, Tables K.1 and K.2, natural (row-major) order */
static const uint8_t k1_luma[64] = {
        16, 11, 10, 16,  24,  40,  51,  61,
        12, 12, 14, 19,  26,  58,  60,  55,
        14, 13, 16, 24,  40,  57,  69,  56,
        14, 17, 22, 29,  51,  87,  80,  62,
        18, 22, 37, 56,  68, 109, 103,  77,
        24, 35, 55, 64,  81, 104, 113,  92,
        49, 64, 78, 87, 103, 121, 120, 101,
        72, 92, 95, 98, 112, 100, 103,  99,
};
static const uint8_t k2_chroma[64] = {
        17, 18, 24, 47, 99, 99, 99, 99,
        18, 21, 26, 66, 99, 99, 99, 99,
        24, 26, 56, 99, 99, 99, 99, 99,
        47, 66, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
};

void oracle_jpeg_qtable(int quality, int comp, uint8_t table[64])
{
        if (quality < 1) quality = 1;
        if (quality > 100) quality = 100;
        const int s = quality < 50 ? 5000 / quality : 200 - quality * 2;
        const uint8_t *base = comp == 0 ? k1_luma : k2_chroma;
        for (int i = 0; i < 64; i++) {
                int t = (base[i] * s + 50) / 100;
                table[i] = t < 1 ? 1 : (t > 255 ? 255 : t);
        }
}

/* AAN post-scale: aan[0] = 1, aan[k] = cos(k*pi/16) * sqrt(2) */
static const double aan_scale[8] = {
        1.0, 1.387039845, 1.306562965, 1.175875602,
        1.0, 0.785694958, 0.541196100, 0.275899379,
};

void oracle_jpeg_divisors(const uint8_t q[64], float div[64])
{
        for (int r = 0; r < 8; r++) {
                for (int c = 0; c < 8; c++) {
                        div[8 * r + c] =
                            (float) (1.0 / ((double) q[8 * r + c] * aan_scale[r] * aan_scale[c] * 8.0));
                }
        }
}

/* one 1-D AAN pass over 8 values with stride `st` (in place) */
static void aan_1d(float *d, int st)
{
        const float c4 = 0.707106781f, c6 = 0.382683433f, c2mc6 = 0.541196100f, c2pc6 = 1.306562965f;
        float t0 = d[0 * st] + d[7 * st], t7 = d[0 * st] - d[7 * st];
        float t1 = d[1 * st] + d[6 * st], t6 = d[1 * st] - d[6 * st];
        float t2 = d[2 * st] + d[5 * st], t5 = d[2 * st] - d[5 * st];
        float t3 = d[3 * st] + d[4 * st], t4 = d[3 * st] - d[4 * st];
        /* even part */
        float t10 = t0 + t3, t13 = t0 - t3;
        float t11 = t1 + t2, t12 = t1 - t2;
        d[0 * st] = t10 + t11;
        d[4 * st] = t10 - t11;
        float z1 = t12 + t13;
        z1 = z1 * c4;
        d[2 * st] = t13 + z1;
        d[6 * st] = t13 - z1;
        /* odd part */
        t10 = t4 + t5;
        t11 = t5 + t6;
        t12 = t6 + t7;
        float z5 = t10 - t12;
        z5 = z5 * c6;
        float z2 = c2mc6 * t10;
        z2 = z2 + z5;
        float z4 = c2pc6 * t12;
        z4 = z4 + z5;
        float z3 = t11 * c4;
        float z11 = t7 + z3, z13 = t7 - z3;
        d[5 * st] = z13 + z2;
        d[3 * st] = z13 - z2;
        d[1 * st] = z11 + z4;
        d[7 * st] = z11 - z4;
}

void oracle_jpeg_fdct_quant_plane(const uint8_t *plane, int ls, int width, int height,
                                  int blocks_w, int blocks_h, const float div[64],
                                  int16_t *out, float *coef_out)
{
        for (int by = 0; by < blocks_h; by++) {
                for (int bx = 0; bx < blocks_w; bx++) {
                        float blk[64];
                        for (int r = 0; r < 8; r++) {
                                int y = 8 * by + r;
                                if (y > height - 1) y = height - 1;
                                for (int c = 0; c < 8; c++) {
                                        int x = 8 * bx + c;
                                        if (x > width - 1) x = width - 1;
                                        blk[8 * r + c] = (float) ((int) plane[(long) y * ls + x] - 128);
                                }
                        }
                        for (int r = 0; r < 8; r++) aan_1d(blk + 8 * r, 1);
                        for (int c = 0; c < 8; c++) aan_1d(blk + c, 8);
                        long b = (long) by * blocks_w + bx;
                        if (coef_out) {
                                for (int i = 0; i < 64; i++) coef_out[64 * b + i] = blk[i];
                        }
                        for (int k = 0; k < 64; k++) {
                                int i = oracle_jpeg_zigzag[k];
                                float q = blk[i] * div[i];
                                out[64 * b + k] = (int16_t) rintf(q);
                        }
                }
        }
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * The encoder's colour stage (color_space_internal of src/video_compress/gpujpeg.cpp:303-305,398-405; GPUJPEG's preprocessor, whose source is
 * not in the reference tree: UNPINNED, like the FDCT).  Restated from the published definitions: luma weights Kr, Kb of BT.601 (0.299,
 * 0.114) and BT.709 (0.2126, 0.0722); E'Cb = (B' - Y') / (2 (1 - Kb)), E'Cr = (R' - Y') / (2 (1 - Kr)); 8-bit limited range
 * 16 + 219 E'Y, 128 + 224 E'C; "256 levels" (JFIF) 255 E'Y, 128 + 255 E'C.  cs: 1 = full-range R'G'B', 2 = BT.601 limited,
 * 3 = BT.601 256 levels, 4 = BT.709 limited (UG_JPEG_CS_* of include/ug_mi355x.h).
 * ------------------------------------------------------------------------------------------------------------------------------- */
static void cs_from_rgb(int cs, double t[3][4])
{
        const int bt709 = cs == 4, full = cs == 3;
        const double kr = bt709 ? 0.2126 : 0.299, kb = bt709 ? 0.0722 : 0.114, kg = 1.0 - kr - kb;
        const double ys = (full ? 255.0 : 219.0) / 255.0, cs_ = (full ? 255.0 : 224.0) / 255.0;
        const double y[3] = { kr, kg, kb }, cb[3] = { -kr / (2 * (1 - kb)), -kg / (2 * (1 - kb)), 0.5 }, cr[3] = { 0.5, -kg / (2 * (1 - kr)), -kb / (2 * (1 - kr)) };
        for (int i = 0; i < 3; i++) {
                t[0][i] = ys * y[i];
                t[1][i] = cs_ * cb[i];
                t[2][i] = cs_ * cr[i];
        }
        t[0][3] = full ? 0.0 : 16.0;
        t[1][3] = t[2][3] = 128.0;
}

int oracle_jpeg_colour_matrix(int cs_in, int cs_out, float m[12])
{
        if (cs_in < 1 || cs_in > 4 || cs_out < 1 || cs_out > 4) return -1;
        double to_rgb[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } }, from_rgb[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } };
        if (cs_in != 1) { /* invert the affine map RGB -> cs_in */
                double a[3][4];
                cs_from_rgb(cs_in, a);
                const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                                   a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
                for (int i = 0; i < 3; i++) {
                        for (int j = 0; j < 3; j++) {
                                const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
                                to_rgb[i][j] = (a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]) / det;
                        }
                }
                for (int i = 0; i < 3; i++) to_rgb[i][3] = -(to_rgb[i][0] * a[0][3] + to_rgb[i][1] * a[1][3] + to_rgb[i][2] * a[2][3]);
        }
        if (cs_out != 1) cs_from_rgb(cs_out, from_rgb);
        for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 4; j++) {
                        double v = j == 3 ? from_rgb[i][3] : 0.0;
                        for (int k = 0; k < 3; k++) v += from_rgb[i][k] * to_rgb[k][j];
                        m[4 * i + j] = (float) v;
                }
        }
        return 0;
}

static float cs_row(const float *m, float a, float b, float c)
{
        float t = m[0] * a;
        t = t + m[1] * b;
        t = t + m[2] * c;
        return t + m[3];
}
static uint8_t cs_code(float v) { return (uint8_t) rintf(fminf(255.0f, fmaxf(0.0f, v))); }

/* fmt 0: packed 3 bytes per pixel; 1: UYVY (every pixel with its pair's chroma; the pair's two chroma results averaged a * 0.5 + b * 0.5);
 * 2: UYVY -> packed 3 bytes per pixel (4:2:2 -> 4:4:4, every pixel with its pair's chroma; cs_in == cs_out: the samples as they are; lines of (width + 1) / 2
 *    pairs) -- what a 4:4:4 stream from a 4:2:2 source codes (gpujpeg.cpp:297-302 with subsampling=444 on UYVY input) */
int oracle_jpeg_colour_convert(int fmt, int cs_in, int cs_out, const uint8_t *src, uint8_t *dst, int width, int height)
{
        float m[12];
        if (oracle_jpeg_colour_matrix(cs_in, cs_out, m) || (fmt == 1 && (width & 1))) return -1;
        if (fmt == 2) {
                const long pairs = (width + 1) / 2;
                for (long y = 0; y < height; y++) {
                        for (long x = 0; x < width; x++) {
                                const uint8_t *p = src + 4 * (y * pairs + x / 2);
                                uint8_t *o = dst + 3 * (y * width + x);
                                const float a = p[1 + 2 * (x & 1)], b = p[0], c = p[2];
                                if (cs_in == cs_out) {
                                        o[0] = p[1 + 2 * (x & 1)]; o[1] = p[0]; o[2] = p[2];
                                } else {
                                        o[0] = cs_code(cs_row(m, a, b, c));
                                        o[1] = cs_code(cs_row(m + 4, a, b, c));
                                        o[2] = cs_code(cs_row(m + 8, a, b, c));
                                }
                        }
                }
                return 0;
        }
        for (long i = 0; i < (long) width * height / (fmt ? 2 : 1); i++) {
                if (fmt == 0) {
                        const float a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
                        dst[3 * i] = cs_code(cs_row(m, a, b, c));
                        dst[3 * i + 1] = cs_code(cs_row(m + 4, a, b, c));
                        dst[3 * i + 2] = cs_code(cs_row(m + 8, a, b, c));
                } else {
                        const float u = src[4 * i], y0 = src[4 * i + 1], v = src[4 * i + 2], y1 = src[4 * i + 3];
                        const float cb0 = cs_row(m + 4, y0, u, v), cb1 = cs_row(m + 4, y1, u, v), cr0 = cs_row(m + 8, y0, u, v), cr1 = cs_row(m + 8, y1, u, v);
                        const float cb = cb0 * 0.5f + cb1 * 0.5f, cr = cr0 * 0.5f + cr1 * 0.5f;
                        dst[4 * i] = cs_code(cb);
                        dst[4 * i + 1] = cs_code(cs_row(m, y0, u, v));
                        dst[4 * i + 2] = cs_code(cr);
                        dst[4 * i + 3] = cs_code(cs_row(m, y1, u, v));
                }
        }
        return 0;
}
