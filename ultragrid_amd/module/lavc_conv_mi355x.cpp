/**
 * lavc_conv_mi355x.cpp -- the GPU hook of UltraGrid's lavc pixel-format conversions, for MI355X.
 *
 * The reference reserves six functions for a GPU implementation and ships them as stubs that return NULL:
 *     to_lavc_vid_conv_cuda_init / to_lavc_vid_conv_cuda / to_lavc_vid_conv_cuda_destroy     (src/libavcodec/to_lavc_vid_conv_cuda.h:55-66)
 *     get_av_to_uv_cuda_conversion / av_to_uv_convert_cuda / av_to_uv_conversion_cuda_destroy (src/libavcodec/from_lavc_vid_conv_cuda.h:55-66)
 * This file defines exactly those symbols on top of libug_mi355x.so (ug_hip_uv_to_av / ug_hip_av_to_uv), so building UltraGrid with
 * HAVE_LAVC_CUDA_CONV and this object instead of the two *_cuda.cu files gives to_lavc_vid_conv.c:1901-1906,2002-2004 and
 * from_lavc_vid_conv.c:2284-2295,2666-2670 a working device path.  The reference's callers hand over host memory (the capture buffer, the
 * encoder's / decoder's AVFrame), so each call uploads, converts and downloads on one stream; the conversions themselves are the kernels of
 * csrc/lavc_conv.hip, byte-identical to the CPU converters the hook bypasses.
 *
 * An unsupported pair returns NULL from the init / get function -- the reference then logs it and takes its CPU path.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {
#include <libavutil/frame.h>
#include <libavutil/pixdesc.h>
}

#include "libavcodec/from_lavc_vid_conv_cuda.h"
#include "libavcodec/to_lavc_vid_conv.h" // get_av_pixfmt_details
#include "libavcodec/to_lavc_vid_conv_cuda.h"
#include "video_codec.h"

#include "../../include/ug_mi355x.h"

namespace {

int plane_rows(const AVPixFmtDescriptor *d, int plane, int height)
{
        return plane == 0 ? height : (height + (1 << d->log2_chroma_h) - 1) >> d->log2_chroma_h;
}

struct device_frame {
        void *data[4] = {};
        size_t size[4] = {};
        int planes = 0;
        bool ensure(const AVFrame *f, const AVPixFmtDescriptor *d)
        {
                planes = 0;
                for (int i = 0; i < 4 && f->data[i] != nullptr; i++) {
                        const size_t need = (size_t) f->linesize[i] * plane_rows(d, i, f->height);
                        if (size[i] < need) {
                                ug_hip_free(data[i]);
                                data[i] = nullptr;
                                // one spare line: some converters read whole pixel groups past the end of the last line (FFmpeg pads its buffers too)
                                if (ug_hip_malloc(&data[i], need + (size_t) f->linesize[i]) != UG_HIP_SUCCESS) {
                                        return false;
                                }
                                size[i] = need;
                        }
                        planes++;
                }
                return planes > 0;
        }
        void release()
        {
                for (auto &p : data) {
                        ug_hip_free(p);
                        p = nullptr;
                }
        }
        ug_av_frame view(const AVFrame *f) const
        {
                ug_av_frame v = {};
                for (int i = 0; i < planes; i++) {
                        v.data[i] = data[i];
                        v.linesize[i] = f->linesize[i];
                }
                v.width = f->width;
                v.height = f->height;
                v.colorspace = f->colorspace;
                v.color_range = f->color_range;
                return v;
        }
};

} // namespace

struct to_lavc_vid_conv_cuda {
        AVFrame *out_frame = nullptr;
        const char *uv = nullptr, *av = nullptr;
        const AVPixFmtDescriptor *desc = nullptr;
        void *in_dev = nullptr;
        size_t in_size = 0;
        device_frame dev;
        ug_hip_stream_t stream = nullptr;
};

struct av_to_uv_convert_cuda {
        const char *uv = nullptr, *av = nullptr;
        const AVPixFmtDescriptor *desc = nullptr;
        codec_t out_codec = VIDEO_CODEC_NONE;
        device_frame dev;
        void *dst_dev = nullptr;
        size_t dst_size = 0;
        ug_hip_stream_t stream = nullptr;
};

extern "C" {

struct to_lavc_vid_conv_cuda *
to_lavc_vid_conv_cuda_init(codec_t in_pixfmt, int width, int height, enum AVPixelFormat out_pixfmt)
{
        const char *uv = get_codec_name(in_pixfmt), *av = av_get_pix_fmt_name(out_pixfmt);
        int ndev = 0;
        if (uv == nullptr || av == nullptr || !ug_hip_uv_to_av_supported(uv, av) || ug_hip_device_count(&ndev) != UG_HIP_SUCCESS || ndev < 1) {
                return nullptr;
        }
        auto *s = new struct to_lavc_vid_conv_cuda();
        s->uv = uv;
        s->av = av;
        s->desc = av_pix_fmt_desc_get(out_pixfmt);
        s->out_frame = av_frame_alloc();
        s->out_frame->pts = -1;
        s->out_frame->format = out_pixfmt;
        s->out_frame->width = width;
        s->out_frame->height = height;
        get_av_pixfmt_details(out_pixfmt, &s->out_frame->colorspace, &s->out_frame->color_range);
        s->in_size = (size_t) vc_get_linesize(width, in_pixfmt) * height;
        if (av_frame_get_buffer(s->out_frame, 0) != 0 || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS ||
            ug_hip_malloc(&s->in_dev, s->in_size + MAX_PADDING) != UG_HIP_SUCCESS || !s->dev.ensure(s->out_frame, s->desc)) {
                to_lavc_vid_conv_cuda_destroy(&s);
                return nullptr;
        }
        // some converters read what the planes already hold (r10k_to_yuv420p10le) or leave padding untouched: start the device planes
        // with the content of the frame the CPU path would start from
        for (int i = 0; i < s->dev.planes; i++) {
                ug_hip_memcpy(s->dev.data[i], s->out_frame->data[i], (size_t) s->out_frame->linesize[i] * plane_rows(s->desc, i, height),
                              UG_HIP_MEMCPY_HOST_TO_DEVICE);
        }
        // Some rows depend on the geometry (v210 -> p010le needs width % 6 == 0 and an even height on the device, the reference's
        // CPU function has a ragged-edge path): a dry run of the conversion on the device buffers decides.  If it is refused the hook
        // declines here -- NULL makes to_lavc_vid_conv_init() set up its own CPU conversion (to_lavc_vid_conv.c:1901-1906) -- instead
        // of initialising and then returning NULL for every frame.
        const ug_av_frame probe = s->dev.view(s->out_frame);
        if (ug_hip_uv_to_av(s->uv, s->av, s->in_dev, &probe, s->stream) != UG_HIP_SUCCESS || ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                if (getenv("UG_MI355X_VERBOSE") != nullptr) {
                        fprintf(stderr, "[lavc_conv_mi355x] %s -> %s at %dx%d is left to the CPU path: %s\n", uv, av, width, height, ug_hip_last_error_string());
                }
                to_lavc_vid_conv_cuda_destroy(&s);
                return nullptr;
        }
        for (int i = 0; i < s->dev.planes; i++) { // the dry run wrote the planes: start again from the CPU path's initial frame
                ug_hip_memcpy(s->dev.data[i], s->out_frame->data[i], (size_t) s->out_frame->linesize[i] * plane_rows(s->desc, i, height),
                              UG_HIP_MEMCPY_HOST_TO_DEVICE);
        }
        return s;
}

struct AVFrame *
to_lavc_vid_conv_cuda(struct to_lavc_vid_conv_cuda *s, const char *in_data)
{
        if (ug_hip_memcpy_async(s->in_dev, in_data, s->in_size, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) != UG_HIP_SUCCESS) {
                return nullptr;
        }
        if (getenv("UG_MI355X_VERBOSE") != nullptr) {
                fprintf(stderr, "[lavc_conv_mi355x] to_lavc %s -> %s on the device\n", s->uv, s->av);
        }
        const ug_av_frame out = s->dev.view(s->out_frame);
        if (ug_hip_uv_to_av(s->uv, s->av, s->in_dev, &out, s->stream) != UG_HIP_SUCCESS) {
                fprintf(stderr, "[lavc_conv_mi355x] %s -> %s: %s\n", s->uv, s->av, ug_hip_last_error_string());
                return nullptr;
        }
        for (int i = 0; i < s->dev.planes; i++) {
                const size_t n = (size_t) s->out_frame->linesize[i] * plane_rows(s->desc, i, s->out_frame->height);
                if (ug_hip_memcpy_async(s->out_frame->data[i], s->dev.data[i], n, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) != UG_HIP_SUCCESS) {
                        return nullptr;
                }
        }
        return ug_hip_stream_sync(s->stream) == UG_HIP_SUCCESS ? s->out_frame : nullptr;
}

void
to_lavc_vid_conv_cuda_destroy(struct to_lavc_vid_conv_cuda **state)
{
        if (state == nullptr || *state == nullptr) {
                return;
        }
        struct to_lavc_vid_conv_cuda *s = *state;
        s->dev.release();
        ug_hip_free(s->in_dev);
        if (s->stream != nullptr) {
                ug_hip_stream_destroy(s->stream);
        }
        av_frame_free(&s->out_frame);
        delete s;
        *state = nullptr;
}

struct av_to_uv_convert_cuda *
get_av_to_uv_cuda_conversion(enum AVPixelFormat av_codec, codec_t uv_codec)
{
        const char *uv = get_codec_name(uv_codec), *av = av_get_pix_fmt_name(av_codec);
        int ndev = 0;
        if (uv == nullptr || av == nullptr || !ug_hip_av_to_uv_supported(av, uv) || ug_hip_device_count(&ndev) != UG_HIP_SUCCESS || ndev < 1) {
                return nullptr;
        }
        auto *s = new struct av_to_uv_convert_cuda();
        s->uv = uv;
        s->av = av;
        s->desc = av_pix_fmt_desc_get(av_codec);
        s->out_codec = uv_codec;
        if (ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS) {
                delete s;
                return nullptr;
        }
        return s;
}

void
av_to_uv_convert_cuda(struct av_to_uv_convert_cuda *s, char *__restrict dst_buffer, struct AVFrame *__restrict in_frame, int width, int height,
                      int pitch, const int *__restrict rgb_shift)
{
        // The caller's buffer holds `height` lines `pitch` apart, but its LAST line may be only vc_get_linesize() long (a display
        // pitch larger than the line, video_display.h): move pitch * (height - 1) + linesize bytes, never pitch * height.
        const size_t linesize = (size_t) vc_get_linesize(width, s->out_codec);
        const size_t dst_need = (size_t) pitch * (height > 0 ? height - 1 : 0) + ((size_t) pitch < linesize ? (size_t) pitch : linesize);
        if (s->dst_size < dst_need) {
                ug_hip_free(s->dst_dev);
                s->dst_dev = nullptr;
                s->dst_size = 0;
                if (ug_hip_malloc(&s->dst_dev, dst_need + (size_t) pitch + MAX_PADDING) != UG_HIP_SUCCESS) {
                        return;
                }
                s->dst_size = dst_need;
        }
        if (!s->dev.ensure(in_frame, s->desc)) {
                return;
        }
        bool ok = true;
        for (int i = 0; i < s->dev.planes && ok; i++) {
                const size_t n = (size_t) in_frame->linesize[i] * plane_rows(s->desc, i, in_frame->height);
                ok = ug_hip_memcpy_async(s->dev.data[i], in_frame->data[i], n, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) == UG_HIP_SUCCESS;
        }
        // what the converters leave untouched (odd last line, line padding) stays as the caller's buffer has it
        ok = ok && ug_hip_memcpy_async(s->dst_dev, dst_buffer, dst_need, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) == UG_HIP_SUCCESS;
        if (!ok) { // a failed upload must not deliver a stale or garbage frame: leave the caller's buffer as it is
                fprintf(stderr, "[lavc_conv_mi355x] %s -> %s: upload failed: %s\n", s->av, s->uv, ug_hip_last_error_string());
                ug_hip_stream_sync(s->stream);
                return;
        }
        if (getenv("UG_MI355X_VERBOSE") != nullptr) {
                fprintf(stderr, "[lavc_conv_mi355x] from_lavc %s -> %s on the device\n", s->av, s->uv);
        }
        const ug_av_frame in = s->dev.view(in_frame);
        if (ug_hip_av_to_uv(s->av, s->uv, s->dst_dev, pitch, &in, rgb_shift, s->stream) != UG_HIP_SUCCESS) {
                fprintf(stderr, "[lavc_conv_mi355x] %s -> %s: %s\n", s->av, s->uv, ug_hip_last_error_string());
                ug_hip_stream_sync(s->stream);
                return;
        }
        if (ug_hip_memcpy_async(dst_buffer, s->dst_dev, dst_need, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) != UG_HIP_SUCCESS ||
            ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                fprintf(stderr, "[lavc_conv_mi355x] %s -> %s: download failed: %s\n", s->av, s->uv, ug_hip_last_error_string());
        }
}

void
av_to_uv_conversion_cuda_destroy(struct av_to_uv_convert_cuda **state)
{
        if (state == nullptr || *state == nullptr) {
                return;
        }
        struct av_to_uv_convert_cuda *s = *state;
        s->dev.release();
        ug_hip_free(s->dst_dev);
        if (s->stream != nullptr) {
                ug_hip_stream_destroy(s->stream);
        }
        delete s;
        *state = nullptr;
}

} // extern "C"
