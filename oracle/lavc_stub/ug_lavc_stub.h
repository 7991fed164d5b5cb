/* TEST INFRASTRUCTURE ONLY.  A stand-in for the few libavutil / libavcodec declarations that UltraGrid's
 * src/libavcodec/{to,from}_lavc_vid_conv.c and utils.c name, so that those reference files -- whose converters are plain C
 * loops over AVFrame::data / linesize -- can be compiled here (FFmpeg's headers are not in this image) and run as the
 * oracle for the lavc conversions (oracle/Makefile `ref_lavc`).  Nothing of FFmpeg is reproduced beyond names: the enum
 * values are arbitrary, AVFrame holds only the fields those files touch, and the three av_frame_* functions are
 * implemented in lavc_stub.c with malloc.  Both sides of every call (the reference code and the test driver) are compiled
 * against this same header, so the layout need not match the real library. */
#ifndef UG_LAVC_STUB_H
#define UG_LAVC_STUB_H
#include <stddef.h>
#include <stdint.h>

#define AV_VERSION_INT(a, b, c) ((a) << 16 | (b) << 8 | (c))
#define LIBAVUTIL_VERSION_MAJOR 59
#define LIBAVUTIL_VERSION_INT AV_VERSION_INT(59, 8, 100)
#define LIBAVCODEC_VERSION_MAJOR 63
#define LIBAVCODEC_VERSION_INT AV_VERSION_INT(63, 0, 100)
#define AV_NUM_DATA_POINTERS 8

enum AVPixelFormat {
        AV_PIX_FMT_NONE = -1,
        AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUYV422, AV_PIX_FMT_RGB24, AV_PIX_FMT_BGR24, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P,
        AV_PIX_FMT_YUVJ420P, AV_PIX_FMT_YUVJ422P, AV_PIX_FMT_YUVJ444P, AV_PIX_FMT_UYVY422, AV_PIX_FMT_NV12, AV_PIX_FMT_RGBA,
        AV_PIX_FMT_BGRA, AV_PIX_FMT_BGR0, AV_PIX_FMT_RGB48LE, AV_PIX_FMT_YUV420P10LE, AV_PIX_FMT_YUV422P10LE,
        AV_PIX_FMT_YUV444P10LE, AV_PIX_FMT_YUV422P12LE, AV_PIX_FMT_YUV444P12LE, AV_PIX_FMT_YUV422P16LE, AV_PIX_FMT_YUV444P16LE,
        AV_PIX_FMT_GBRP, AV_PIX_FMT_GBRAP, AV_PIX_FMT_GBRP10LE, AV_PIX_FMT_GBRP12LE, AV_PIX_FMT_GBRP16LE, AV_PIX_FMT_P010LE,
        AV_PIX_FMT_P210LE, AV_PIX_FMT_AYUV64LE, AV_PIX_FMT_VUYA, AV_PIX_FMT_VUYX, AV_PIX_FMT_X2RGB10LE, AV_PIX_FMT_XV30,
        AV_PIX_FMT_XV36, AV_PIX_FMT_Y210, AV_PIX_FMT_Y212, AV_PIX_FMT_VDPAU, AV_PIX_FMT_VAAPI, AV_PIX_FMT_VULKAN,
        AV_PIX_FMT_DRM_PRIME, AV_PIX_FMT_VIDEOTOOLBOX, AV_PIX_FMT_CUDA, AV_PIX_FMT_QSV,
        AV_PIX_FMT_NB
};
#define AV_PIX_FMT_AYUV64 AV_PIX_FMT_AYUV64LE
#define AV_PIX_FMT_FLAG_HWACCEL (1 << 3)
#define AV_PIX_FMT_FLAG_RGB (1 << 5)

enum AVColorSpace { AVCOL_SPC_RGB = 0, AVCOL_SPC_BT709 = 1, AVCOL_SPC_UNSPECIFIED = 2, AVCOL_SPC_BT470BG = 5, AVCOL_SPC_SMPTE170M = 6, AVCOL_SPC_SMPTE240M = 7 };
enum AVColorRange { AVCOL_RANGE_UNSPECIFIED = 0, AVCOL_RANGE_MPEG = 1, AVCOL_RANGE_JPEG = 2 };
enum AVCodecID { AV_CODEC_ID_NONE = 0 };
enum AVSampleFormat { AV_SAMPLE_FMT_NONE = -1 };

typedef struct AVComponentDescriptor { int plane, step, offset, shift, depth; } AVComponentDescriptor;
typedef struct AVPixFmtDescriptor {
        const char *name;
        uint8_t nb_components, log2_chroma_w, log2_chroma_h;
        uint64_t flags;
        AVComponentDescriptor comp[4];
} AVPixFmtDescriptor;

typedef struct AVFrame {
        uint8_t *data[AV_NUM_DATA_POINTERS];
        int linesize[AV_NUM_DATA_POINTERS];
        int width, height, format;
        int64_t pts;
        enum AVColorSpace colorspace;
        enum AVColorRange color_range;
        void *opaque;
        void *stub_buf[AV_NUM_DATA_POINTERS];
} AVFrame;
typedef struct AVCodecContext AVCodecContext;
typedef struct AVCodec AVCodec;

typedef struct AVDRMObjectDescriptor { int fd; size_t size; uint64_t format_modifier; } AVDRMObjectDescriptor;
typedef struct AVDRMPlaneDescriptor { int object_index; ptrdiff_t offset, pitch; } AVDRMPlaneDescriptor;
typedef struct AVDRMLayerDescriptor { uint32_t format; int nb_planes; AVDRMPlaneDescriptor planes[4]; } AVDRMLayerDescriptor;
typedef struct AVDRMFrameDescriptor { int nb_objects; AVDRMObjectDescriptor objects[4]; int nb_layers; AVDRMLayerDescriptor layers[4]; } AVDRMFrameDescriptor;

#ifdef __cplusplus
extern "C" {
#endif
AVFrame *av_frame_alloc(void);
void av_frame_free(AVFrame **f);
int av_frame_get_buffer(AVFrame *f, int align);
int av_frame_copy_props(AVFrame *dst, const AVFrame *src);
AVFrame *av_frame_clone(const AVFrame *src);
int av_frame_make_writable(AVFrame *f);
const char *av_get_pix_fmt_name(enum AVPixelFormat f);
const AVPixFmtDescriptor *av_pix_fmt_desc_get(enum AVPixelFormat f);
const char *av_color_space_name(enum AVColorSpace s);
#ifdef __cplusplus
}
#endif
#endif
