/* ug_codec_map.h -- UltraGrid codec_t (src/types.h:62-112) <-> ug_pixfmt_t (include/ug_mi355x.h).
 * The kernel library keeps its own numbering so that it does not depend on the order of the
 * reference's enum; this is the only place where the two meet. */
#ifndef UG_CODEC_MAP_H
#define UG_CODEC_MAP_H

#include "types.h"
#include "../../include/ug_mi355x.h"

static inline ug_pixfmt_t ug_pixfmt_from_codec(codec_t c)
{
        switch (c) {
        case RGBA: return UG_PF_RGBA;
        case UYVY: return UG_PF_UYVY;
        case YUYV: return UG_PF_YUYV;
        case RGB:  return UG_PF_RGB;
        case BGR:  return UG_PF_BGR;
        case v210: return UG_PF_V210;
        case RG48: return UG_PF_RG48;
        case I420: return UG_PF_I420;
        case R10k: return UG_PF_R10K;
        case R12L: return UG_PF_R12L;
        case Y216: return UG_PF_Y216;
        case Y416: return UG_PF_Y416;
        case VUYA: return UG_PF_VUYA;
        case DVS10: return UG_PF_DVS10;
        default:   return UG_PF_NONE;
        }
}

static inline codec_t ug_codec_from_pixfmt(ug_pixfmt_t f)
{
        switch (f) {
        case UG_PF_RGBA: return RGBA;
        case UG_PF_UYVY: return UYVY;
        case UG_PF_YUYV: return YUYV;
        case UG_PF_RGB:  return RGB;
        case UG_PF_BGR:  return BGR;
        case UG_PF_V210: return v210;
        case UG_PF_RG48: return RG48;
        case UG_PF_I420: return I420;
        case UG_PF_R10K: return R10k;
        case UG_PF_R12L: return R12L;
        case UG_PF_Y216: return Y216;
        case UG_PF_Y416: return Y416;
        case UG_PF_VUYA: return VUYA;
        case UG_PF_DVS10: return DVS10;
        default:         return VIDEO_CODEC_NONE;
        }
}
#endif
