// ug_common.h -- internal helpers shared by the HIP translation units of libug_mi355x.so
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ug_mi355x.h"

namespace ug {

// thread-local copy of the last HIP error text (cuda_wrapper.cu:115-118 keeps the same state)
void set_last_error(hipError_t e, const char *what);
void set_last_error_msg(const char *msg);

#define UG_HIP_TRY(expr)                                                     \
        do {                                                                 \
                hipError_t e_ = (expr);                                      \
                if (e_ != hipSuccess) {                                      \
                        ug::set_last_error(e_, #expr);                       \
                        return UG_HIP_ERUNTIME;                              \
                }                                                            \
        } while (0)

// check the asynchronous launch error right after a <<<>>> launch
#define UG_HIP_LAUNCH_CHECK()                                                \
        do {                                                                 \
                hipError_t e_ = hipGetLastError();                           \
                if (e_ != hipSuccess) {                                      \
                        ug::set_last_error(e_, "kernel launch");             \
                        return UG_HIP_ERUNTIME;                              \
                }                                                            \
        } while (0)

static inline int linesize(ug_pixfmt_t f, int width)
{
        // video_codec.c:120-206 (block bytes / pixels, h_align) and :507-521
        int bb, bp, ha;
        switch (f) {
        case UG_PF_RGBA: bb = 4; bp = 1; ha = 1; break;
        case UG_PF_UYVY:
        case UG_PF_UYVY_RAW:
        case UG_PF_YUYV: bb = 4; bp = 2; ha = 2; break;
        case UG_PF_RGB:
        case UG_PF_BGR:
        case UG_PF_YUV444: bb = 3; bp = 1; ha = 1; break;
        case UG_PF_DVS10:
        case UG_PF_V210: bb = 16; bp = 6; ha = 48; break;
        case UG_PF_R10K: bb = 4; bp = 1; ha = 64; break;
        case UG_PF_R12L: bb = 36; bp = 8; ha = 8; break;
        case UG_PF_Y216: bb = 8; bp = 2; ha = 2; break;
        case UG_PF_Y416: bb = 8; bp = 1; ha = 1; break;
        case UG_PF_VUYA: bb = 4; bp = 1; ha = 1; break;
        case UG_PF_RG48: bb = 6; bp = 1; ha = 1; break;
        default: return 0;
        }
        width = (width + ha - 1) / ha * ha;
        return (width + bp - 1) / bp * bb;
}

// jpeg_fdct.hip: FDCT+quantise of one 8-bit component, planar (xstride 1) or packed (xstride = bytes per pixel)
int jpeg_fdct_quant_strided(const void *plane, int pitch, int xstride, int width, int height, int blocks_w, int blocks_h,
                            const float *div, int16_t *out, float *coef, ug_hip_stream_t stream);

int jpeg_fdct_quant_rgb444(const void *src, int pitch, int width, int height, int blocks_w, int blocks_h, const float *div,
                           int16_t *out_r, int16_t *out_g, int16_t *out_b, ug_hip_stream_t stream);

// get_color_coeffs(cs, depth) (color_space.c:149-184) as constants: [cs - 1][slot], cs 1 = BT.601, 2 = BT.709; slot 0 = full range, then
// 8, 10, 12, 16 bit limited range; 14 ints in the field order of struct color_coeffs.  Defined in lavc_conv.hip, checked against the compiled
// reference through ug_hip_color_coeffs (tests/test_lavc_conv.py).
extern const int kColorCoeffs[2][5][14];
static inline int color_depth_slot(int depth) { return depth == 0 ? 0 : depth == 8 ? 1 : depth == 10 ? 2 : depth == 12 ? 3 : depth == 16 ? 4 : -1; }

// pixfmt_ext.hip: the pairs of decoders[] outside pixfmt.hip's core
int pixfmt_ext_supported(ug_pixfmt_t in, ug_pixfmt_t out);
int pixfmt_ext_convert(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height, int src_pitch, int dst_pitch, int dst_len,
                       int rshift, int gshift, int bshift, hipStream_t st);

} // namespace ug
