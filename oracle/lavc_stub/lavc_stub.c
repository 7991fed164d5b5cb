/* TEST INFRASTRUCTURE ONLY -- see ug_lavc_stub.h.  Frame allocation and the pixel-format descriptors the reference's lavc
 * conversion files ask for, plus the four UltraGrid symbols they import from files that need the real FFmpeg
 * (lavc_common.c) or the application (host.cpp, tv.c).  The descriptor facts (planes, sample size, chroma shifts) are
 * the published layouts of the named FFmpeg pixel formats. */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ug_lavc_stub.h"
#include "types.h" /* struct pixfmt_desc */

struct fmt_info {
        enum AVPixelFormat f;
        const char *name;
        int planes;       /* data[] entries */
        int bytes[4];     /* bytes per line = bytes[i] * ceil(width >> (i ? cw : 0)) */
        int cw, ch;       /* log2 chroma subsampling */
        int depth, rgb, comps;
};
static const struct fmt_info k_fmts[] = {
        { AV_PIX_FMT_YUV420P, "yuv420p", 3, { 1, 1, 1 }, 1, 1, 8, 0, 3 },
        { AV_PIX_FMT_YUVJ420P, "yuvj420p", 3, { 1, 1, 1 }, 1, 1, 8, 0, 3 },
        { AV_PIX_FMT_YUV422P, "yuv422p", 3, { 1, 1, 1 }, 1, 0, 8, 0, 3 },
        { AV_PIX_FMT_YUVJ422P, "yuvj422p", 3, { 1, 1, 1 }, 1, 0, 8, 0, 3 },
        { AV_PIX_FMT_YUV444P, "yuv444p", 3, { 1, 1, 1 }, 0, 0, 8, 0, 3 },
        { AV_PIX_FMT_YUVJ444P, "yuvj444p", 3, { 1, 1, 1 }, 0, 0, 8, 0, 3 },
        { AV_PIX_FMT_YUV420P10LE, "yuv420p10le", 3, { 2, 2, 2 }, 1, 1, 10, 0, 3 },
        { AV_PIX_FMT_YUV422P10LE, "yuv422p10le", 3, { 2, 2, 2 }, 1, 0, 10, 0, 3 },
        { AV_PIX_FMT_YUV444P10LE, "yuv444p10le", 3, { 2, 2, 2 }, 0, 0, 10, 0, 3 },
        { AV_PIX_FMT_YUV422P12LE, "yuv422p12le", 3, { 2, 2, 2 }, 1, 0, 12, 0, 3 },
        { AV_PIX_FMT_YUV444P12LE, "yuv444p12le", 3, { 2, 2, 2 }, 0, 0, 12, 0, 3 },
        { AV_PIX_FMT_YUV422P16LE, "yuv422p16le", 3, { 2, 2, 2 }, 1, 0, 16, 0, 3 },
        { AV_PIX_FMT_YUV444P16LE, "yuv444p16le", 3, { 2, 2, 2 }, 0, 0, 16, 0, 3 },
        { AV_PIX_FMT_NV12, "nv12", 2, { 1, 2 }, 1, 1, 8, 0, 3 },
        { AV_PIX_FMT_P010LE, "p010le", 2, { 2, 4 }, 1, 1, 10, 0, 3 },
        { AV_PIX_FMT_P210LE, "p210le", 2, { 2, 4 }, 1, 0, 10, 0, 3 },
        { AV_PIX_FMT_GBRP, "gbrp", 3, { 1, 1, 1 }, 0, 0, 8, 1, 3 },
        { AV_PIX_FMT_GBRAP, "gbrap", 4, { 1, 1, 1, 1 }, 0, 0, 8, 1, 4 },
        { AV_PIX_FMT_GBRP10LE, "gbrp10le", 3, { 2, 2, 2 }, 0, 0, 10, 1, 3 },
        { AV_PIX_FMT_GBRP12LE, "gbrp12le", 3, { 2, 2, 2 }, 0, 0, 12, 1, 3 },
        { AV_PIX_FMT_GBRP16LE, "gbrp16le", 3, { 2, 2, 2 }, 0, 0, 16, 1, 3 },
        { AV_PIX_FMT_RGB24, "rgb24", 1, { 3 }, 0, 0, 8, 1, 3 },
        { AV_PIX_FMT_BGR24, "bgr24", 1, { 3 }, 0, 0, 8, 1, 3 },
        { AV_PIX_FMT_RGBA, "rgba", 1, { 4 }, 0, 0, 8, 1, 4 },
        { AV_PIX_FMT_BGRA, "bgra", 1, { 4 }, 0, 0, 8, 1, 4 },
        { AV_PIX_FMT_BGR0, "bgr0", 1, { 4 }, 0, 0, 8, 1, 3 },
        { AV_PIX_FMT_RGB48LE, "rgb48le", 1, { 6 }, 0, 0, 16, 1, 3 },
        { AV_PIX_FMT_X2RGB10LE, "x2rgb10le", 1, { 4 }, 0, 0, 10, 1, 3 },
        { AV_PIX_FMT_UYVY422, "uyvy422", 1, { 2 }, 1, 0, 8, 0, 3 },
        { AV_PIX_FMT_YUYV422, "yuyv422", 1, { 2 }, 1, 0, 8, 0, 3 },
        { AV_PIX_FMT_Y210, "y210le", 1, { 4 }, 1, 0, 10, 0, 3 },
        { AV_PIX_FMT_Y212, "y212le", 1, { 4 }, 1, 0, 12, 0, 3 },
        { AV_PIX_FMT_VUYA, "vuya", 1, { 4 }, 0, 0, 8, 0, 4 },
        { AV_PIX_FMT_VUYX, "vuyx", 1, { 4 }, 0, 0, 8, 0, 3 },
        { AV_PIX_FMT_XV30, "xv30le", 1, { 4 }, 0, 0, 10, 0, 3 },
        { AV_PIX_FMT_XV36, "xv36le", 1, { 8 }, 0, 0, 12, 0, 3 },
        { AV_PIX_FMT_AYUV64LE, "ayuv64le", 1, { 8 }, 0, 0, 16, 0, 4 },
};
#define N_FMTS (sizeof k_fmts / sizeof k_fmts[0])

static const struct fmt_info *info(enum AVPixelFormat f)
{
        for (size_t i = 0; i < N_FMTS; i++) {
                if (k_fmts[i].f == f) return &k_fmts[i];
        }
        return NULL;
}

AVFrame *av_frame_alloc(void) { return calloc(1, sizeof(AVFrame)); }

void av_frame_free(AVFrame **f)
{
        if (!f || !*f) return;
        for (int i = 0; i < AV_NUM_DATA_POINTERS; i++) free((*f)->stub_buf[i]);
        free(*f);
        *f = NULL;
}

int av_frame_get_buffer(AVFrame *f, int align)
{
        (void) align;
        const struct fmt_info *fi = info((enum AVPixelFormat) f->format);
        if (!fi) return -1;
        for (int i = 0; i < fi->planes; i++) {
                /* packed 4:2:2 formats store bytes[0] per PIXEL of an even-rounded line */
                int w = i ? (f->width + (1 << fi->cw) - 1) >> fi->cw : f->width;
                if (fi->planes == 1 && fi->cw) w = (f->width + 1) / 2 * 2;
                const int h = i ? (f->height + (1 << fi->ch) - 1) >> fi->ch : f->height;
                f->linesize[i] = (w * fi->bytes[i] + 63) / 64 * 64; /* padded lines, as FFmpeg's allocator makes them */
                f->stub_buf[i] = calloc((size_t) f->linesize[i] * (h + 1) + 64, 1);
                f->data[i] = f->stub_buf[i];
        }
        return 0;
}

int av_frame_copy_props(AVFrame *dst, const AVFrame *src)
{
        dst->pts = src->pts;
        dst->colorspace = src->colorspace;
        dst->color_range = src->color_range;
        return 0;
}
AVFrame *av_frame_clone(const AVFrame *src) { (void) src; return NULL; }
int av_frame_make_writable(AVFrame *f) { (void) f; return 0; }

const char *av_get_pix_fmt_name(enum AVPixelFormat f)
{
        const struct fmt_info *fi = info(f);
        return fi ? fi->name : NULL;
}

const AVPixFmtDescriptor *av_pix_fmt_desc_get(enum AVPixelFormat f)
{
        static AVPixFmtDescriptor d[N_FMTS];
        static AVPixFmtDescriptor hw = { "hw", 0, 0, 0, AV_PIX_FMT_FLAG_HWACCEL, { { 0 } } };
        for (size_t i = 0; i < N_FMTS; i++) {
                if (k_fmts[i].f != f) continue;
                d[i].name = k_fmts[i].name;
                d[i].nb_components = k_fmts[i].comps;
                d[i].log2_chroma_w = k_fmts[i].cw;
                d[i].log2_chroma_h = k_fmts[i].ch;
                d[i].flags = k_fmts[i].rgb ? AV_PIX_FMT_FLAG_RGB : 0;
                for (int c = 0; c < 4; c++) d[i].comp[c].depth = k_fmts[i].depth;
                return &d[i];
        }
        return f == AV_PIX_FMT_NONE ? NULL : &hw;
}

const char *av_color_space_name(enum AVColorSpace s) { return s == AVCOL_SPC_BT709 ? "bt709" : "other"; }

/* ---- UltraGrid symbols whose home files cannot be compiled here ---- */
int av_pixfmt_get_subsampling(enum AVPixelFormat fmt) /* lavc_common.c:211-223 needs only the descriptor */
{
        const AVPixFmtDescriptor *pd = av_pix_fmt_desc_get(fmt);
        if (pd->log2_chroma_w == 0 && pd->log2_chroma_h == 0) return 4440;
        if (pd->log2_chroma_w == 1 && pd->log2_chroma_h == 0) return 4220;
        if (pd->log2_chroma_w == 1 && pd->log2_chroma_h == 1) return 4200;
        return 0;
}
struct pixfmt_desc av_pixfmt_get_desc(enum AVPixelFormat pixfmt)
{
        struct pixfmt_desc ret = { 0 };
        const AVPixFmtDescriptor *avd = av_pix_fmt_desc_get(pixfmt);
        ret.depth = avd->comp[0].depth;
        ret.rgb = avd->flags & AV_PIX_FMT_FLAG_RGB;
        ret.subsampling = av_pixfmt_get_subsampling(pixfmt);
        return ret;
}
void print_libav_error(int verbosity, const char *msg, int rc) { (void) verbosity; fprintf(stderr, "%s: %d\n", msg, rc); }
bool cuda_devices_explicit = false;
long long get_time_in_ns(void)
{
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

/* ---- driver helpers for the ctypes side (tests) ---- */
AVFrame *ug_stub_frame_new(int format, int width, int height)
{
        AVFrame *f = av_frame_alloc();
        f->format = format;
        f->width = width;
        f->height = height;
        if (av_frame_get_buffer(f, 0)) {
                av_frame_free(&f);
                return NULL;
        }
        return f;
}
int ug_stub_pixfmt_by_name(const char *name)
{
        for (size_t i = 0; i < N_FMTS; i++) {
                if (!strcmp(k_fmts[i].name, name)) return k_fmts[i].f;
        }
        return AV_PIX_FMT_NONE;
}
int ug_stub_plane_rows(int format, int plane, int height)
{
        const struct fmt_info *fi = info((enum AVPixelFormat) format);
        if (!fi || plane >= fi->planes) return 0;
        return plane ? (height + (1 << fi->ch) - 1) >> fi->ch : height;
}
