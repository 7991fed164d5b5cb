/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatements of the UltraGrid hot path (pixel-format conversion, DXT1 /
 * DXT5-YCoCg block encode, JPEG FDCT+quantise).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or load this; the product
 * library (ultragrid_amd/csrc) never does.
 */
#ifndef UG_ORACLE_H
#define UG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DXT (dxt_oracle.c) ---- */
enum {
        ORACLE_IN_RGB      = 0, /* 3 B/px */
        ORACLE_IN_RGBA     = 1, /* 4 B/px, alpha ignored */
        ORACLE_IN_YUV444   = 2, /* packed Y,U,V 3 B/px -> YUV->RGB (cuda_yuv_to_dxt*) */
        ORACLE_IN_UYVY     = 3, /* 4:2:2, chroma replicated, YUV->RGB */
        ORACLE_IN_UYVY_RAW = 4, /* 4:2:2, chroma replicated, NO colour conversion (DXT1_YUV) */
        ORACLE_IN_V210     = 5, /* 10-bit 4:2:2, samples >>2, then as UYVY */
};
enum {
        ORACLE_OUT_DXT1      = 1,
        ORACLE_OUT_DXT1_YUV  = 2, /* decode side only: DXT1 blocks holding Y,Cb,Cr (encode = ORACLE_IN_UYVY_RAW -> DXT1) */
        ORACLE_OUT_DXT5YCOCG = 6,
};

void oracle_dxt5ycocg_encode_block(const float rgb[16][3], uint32_t out[4]);
void oracle_dxt1_encode_block(const float rgb[16][3], uint32_t out[2]);
/* h < 0: source read bottom-up. pitch = source line stride in bytes. 0 ok, -1 bad args */
/* GLSL's implementation-defined choices (round() ties, dot(vec3) order, unorm8 ties on the decode side):
 * oracle_set_ties(0) = "ties even", Mesa's choices = the default, pinned to the reference's shaders executed on llvmpipe;
 * oracle_set_ties(1) = "ties away", the text of the reference's CUDA port.  The three single knobs remain for the tests that
 * take them apart. */
void oracle_set_ties(int away);
void oracle_set_round_half_even(int on);
void oracle_set_dot3_reverse(int on);
void oracle_set_unorm_ties_even(int on); /* decode side: float -> unorm8 ties (dxt_decode_oracle.c) */
int  oracle_dxt_encode(int in_fmt, int out_fmt, const uint8_t *src, uint8_t *dst,
                       int w, int h, long pitch);
/* row bands over nthreads OpenMP threads (0 = all cores); cpu_baseline timing only */
int  oracle_dxt_encode_mt(int in_fmt, int out_fmt, const uint8_t *src, uint8_t *dst,
                          int w, int h, long pitch, int nthreads);
void oracle_yuv422_to_yuv444(const uint8_t *src, uint8_t *dst, long pix_count);

/* ---- DXT decode (dxt_decode_oracle.c) ---- */
void oracle_dxt5ycocg_decode_rgb(const uint8_t *src, uint8_t *dst_rgb, int w, int h);
void oracle_dxt1_decode_rgb(const uint8_t *src, uint8_t *dst_rgb, int w, int h);
void oracle_dxt1yuv_decode_rgb(const uint8_t *src, uint8_t *dst_rgb, int w, int h);
/* in_fmt ORACLE_OUT_*; out_fmt OPF_RGB / OPF_BGR / OPF_RGBA / OPF_UYVY; 0 ok, -1 bad args */
int  oracle_dxt_decode(int in_fmt, int out_fmt, const uint8_t *src, uint8_t *dst, int w, int h, long dst_pitch,
                       int rs, int gs, int bs);

/* ---- pixfmt (pixfmt_oracle.c) ---- */
enum {
        OPF_RGBA = 1, OPF_UYVY = 2, OPF_YUYV = 3, OPF_RGB = 4, OPF_BGR = 5,
        OPF_V210 = 6, OPF_RG48 = 7, OPF_I420 = 8,
};
/* bytes per line as vc_get_linesize (video_codec.c:507-521) */
int  oracle_linesize(int width, int fmt);
/* bytes written for `width` px as vc_get_size (video_codec.c:530-538) */
int  oracle_size(int width, int fmt);
/* one line, decoder_t semantics (pixfmt_conv.h:87-88); returns 0, or -1 if no such pair */
int  oracle_convert_line(int in_fmt, int out_fmt, uint8_t *dst, const uint8_t *src,
                         int dst_len, int rshift, int gshift, int bshift);
/* whole frame by the testcard_convert_buffer line loop (testcard_common.c:121-129) */
int  oracle_convert_frame(int in_fmt, int out_fmt, uint8_t *dst, const uint8_t *src,
                          int width, int height, int rshift, int gshift, int bshift);
/* to_planar.c:343-378 */
/* vc_deinterlace, video_codec.c:597-664 (SSE2 bodies): in place */
void oracle_deinterlace_blend(uint8_t *src, long src_linesize, int lines);
void oracle_uyvy_to_i420(uint8_t *y, int y_ls, uint8_t *u, int u_ls, uint8_t *v, int v_ls,
                         const uint8_t *src, int width, int height);
/* to_planar.c:64-155 */
void oracle_v210_to_p010le(uint16_t *y, int y_ls, uint16_t *uv, int uv_ls,
                           const uint8_t *src, int width, int height);
/* color_space.c:149-184 : Q14 coefficients, cs 709 (default) / 601, depth 0/8/10/12/16.
 * out[14] = y_r y_g y_b cb_r cb_g cb_b cr_r cr_g cr_b y_scale r_cr g_cb g_cr b_cb */
int  oracle_color_coeffs(int bt601, int depth, int out[14]);

/* ---- JPEG FDCT + quantise (jpeg_oracle.c) ---- */
/* quality 1..100 -> 8-bit quant tables in NATURAL order (T.81 Annex K scaled with the
 * IJG formula).  comp 0 = luma, 1 = chroma. */
void oracle_jpeg_qtable(int quality, int comp, uint8_t table[64]);
/* fp32 reciprocal divisors 1/(q * aan[r] * aan[c] * 8), natural order */
void oracle_jpeg_divisors(const uint8_t qtable[64], float div[64]);
/* one plane: 8-bit samples -> level shift (-128) -> AAN float FDCT -> coef*div -> rintf ->
 * int16, zig-zag order within a block, blocks in raster order (blocks_w x blocks_h of
 * them; blocks_w*8 >= width, blocks_h*8 >= height; samples past the right/bottom edge
 * replicate the last column/row).  If coef_out != NULL also stores the unquantised
 * (AAN-scaled) fp32 coefficients, natural order, 64 per block. */
void oracle_jpeg_fdct_quant_plane(const uint8_t *plane, int ls, int width, int height,
                                  int blocks_w, int blocks_h, const float div[64],
                                  int16_t *out, float *coef_out);
/* zig-zag scan: zz[k] = natural index of the k-th coefficient (T.81 Figure A.6) */
extern const uint8_t oracle_jpeg_zigzag[64];

#ifdef __cplusplus
}
#endif
/* jpeg_decode_oracle.c: baseline JPEG -> component planes (T.81 Huffman decoding + libjpeg's jidctint "islow" IDCT).  info[12] =
 * width, height, components, h0, v0, h1, v1, h2, v2, restart interval, Adobe transform (-1 = no marker), scans.  planes[c]: MCU-padded
 * plane of component c ((mcu_w * 8 * h_c) x (mcu_h * 8 * v_c)), pitch[c] bytes per line; planes == NULL: headers only. */
int oracle_jpeg_decode(const uint8_t *data, long len, int info[12], uint8_t *planes[3], const int pitch[3]);

#endif
