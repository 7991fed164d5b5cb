/**
 * @file vdecompress_jpeg_mi355x.c
 * UltraGrid video_decompress module "jpeg_mi355x": JPEG -> UYVY / RGB / RGBA / I420 on an MI355X through libug_mi355x.so
 * (include/ug_mi355x.h: ug_hip_jpeg_decoder_*).  Receiver-side counterpart of vcompress_jpeg_mi355x.cpp; plain C, the callback set and
 * conventions of the reference's GPUJPEG decompress module (src/video_decompress/gpujpeg.c): out_codec VIDEO_CODEC_NONE = probe of the
 * stream's internal pixel format (:202-266), the output codecs and priorities of :355-366, a display pitch different from the line size
 * served by one 2-D copy (the reference loops over the lines on the CPU, :296-319), corrupted frames not accepted (:322-341).
 * The GPU: --param mi355x-device=<n>[:<n>...] or -D (the reference's module takes cuda_devices[0], :162), the states of a process in turn
 * (mi355x_receiver.h).
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "debug.h"
#include "lib_common.h"
#include "types.h"
#include "video_codec.h"
#include "video_decompress.h"

#include "mi355x_receiver.h"

#define MOD_NAME "[JPEG MI355X dec] "

struct state_decompress_jpeg_mi355x {
        struct video_desc    desc;
        int                  rshift, gshift, bshift, pitch;
        codec_t              out_codec;
        ug_pixfmt_t          out_fmt;
        int                  device; ///< --param mi355x-device / -D (mi355x_receiver.h)
        ug_hip_stream_t      stream;
        ug_hip_jpeg_decoder *dec;
        void                *dev_out;
        size_t               out_len;
};

static void jpeg_mi355x_decompress_done(void *state);
static unsigned jpeg_mi355x_state_count; // the states of this process take the listed devices in turn

static void *jpeg_mi355x_decompress_init(void)
{
        struct state_decompress_jpeg_mi355x *s = calloc(1, sizeof *s);
        if (s == NULL) {
                return NULL;
        }
        s->device = mi355x_next_state_device(&jpeg_mi355x_state_count, MOD_NAME);
        if (s->device < 0 || ug_hip_set_device(s->device) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS ||
            ug_hip_jpeg_decoder_create(&s->dec) != UG_HIP_SUCCESS) {
                if (s->device >= 0) MSG(ERROR, "cannot set up the decoder on HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                s->device = s->device < 0 ? 0 : s->device;
                jpeg_mi355x_decompress_done(s); // releases whichever of the two was made
                return NULL;
        }
        return s;
}

static int jpeg_mi355x_decompress_reconfigure(void *state, struct video_desc desc, int rshift, int gshift, int bshift, int pitch,
                                              codec_t out_codec)
{
        struct state_decompress_jpeg_mi355x *s = state;
        if (desc.color_spec != JPEG) {
                MSG(ERROR, "Wrong compression to decompress: %s\n", get_codec_name(desc.color_spec));
                return false;
        }
        switch (out_codec) {
        case VIDEO_CODEC_NONE: s->out_fmt = UG_PF_NONE; break; // probe
        case RGBA: s->out_fmt = UG_PF_RGBA; break;
        case RGB:  s->out_fmt = UG_PF_RGB; break;
        case UYVY: s->out_fmt = UG_PF_UYVY; break;
        case I420: s->out_fmt = UG_PF_I420; break;
        default:
                MSG(ERROR, "Unsupported output codec: %s\n", get_codec_name(out_codec));
                return false;
        }
        if (out_codec != VIDEO_CODEC_NONE && out_codec != I420 && pitch < vc_get_linesize(desc.width, out_codec)) {
                MSG(ERROR, "pitch %d is shorter than a line of %u %s pixels\n", pitch, desc.width, get_codec_name(out_codec));
                return false;
        }
        ug_hip_set_device(s->device);
        if (s->dev_out) {
                ug_hip_free(s->dev_out);
                s->dev_out = NULL;
        }
        s->desc = desc;
        s->rshift = rshift; s->gshift = gshift; s->bshift = bshift;
        s->pitch = pitch;
        s->out_codec = out_codec;
        if (out_codec != VIDEO_CODEC_NONE) {
                s->out_len = out_codec == I420 ? (size_t) desc.width * desc.height + 2 * (size_t) ((desc.width + 1) / 2) * ((desc.height + 1) / 2)
                                               : (size_t) vc_get_linesize(desc.width, out_codec) * desc.height;
                if (ug_hip_malloc(&s->dev_out, s->out_len + 64) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "Could not allocate the device output buffer: %s\n", ug_hip_last_error_string());
                        return false;
                }
        }
        return true;
}

/// gpujpeg_probe_internal_codec (gpujpeg.c:202-266): depth 8, RGB or not, subsampling in the 4xxx notation
static decompress_status probe_internal_codec(unsigned char *buffer, size_t len, struct pixfmt_desc *internal_prop)
{
        int w = 0, h = 0, sub = 0, rgb = 0;
        if (ug_hip_jpeg_read_info(buffer, len, &w, &h, &sub, &rgb, NULL) != UG_HIP_SUCCESS) {
                MSG(WARNING, "probe - cannot get image info!\n");
                return DECODER_NO_FRAME;
        }
        if (internal_prop != NULL) {
                internal_prop->depth = 8;
                internal_prop->rgb = rgb != 0;
                internal_prop->subsampling = sub == 444 ? 4440 : (sub == 422 ? 4220 : (sub == 420 ? 4200 : 4000));
        }
        return DECODER_GOT_CODEC;
}

static decompress_status jpeg_mi355x_decompress(void *state, unsigned char *dst, unsigned char *buffer, unsigned int src_len, int frame_seq,
                                                struct video_frame_callbacks *callbacks, struct pixfmt_desc *internal_prop)
{
        struct state_decompress_jpeg_mi355x *s = state;
        (void) frame_seq, (void) callbacks;
        if (s->out_codec == VIDEO_CODEC_NONE) {
                return probe_internal_codec(buffer, src_len, internal_prop);
        }
        int w = 0, h = 0;
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS || ug_hip_jpeg_read_info(buffer, src_len, &w, &h, NULL, NULL, NULL) != UG_HIP_SUCCESS ||
            (unsigned) w != s->desc.width || (unsigned) h != s->desc.height) {
                MSG(ERROR, "not a JPEG frame of the configured size %ux%u\n", s->desc.width, s->desc.height);
                return DECODER_NO_FRAME;
        }
        // _sized: the decoder checks the frame header IT parses against the size dev_out was allocated for (a stream with one scan per component
        // is parsed a second time, further than ug_hip_jpeg_read_info looks)
        if (ug_hip_jpeg_decoder_decode_sized(s->dec, buffer, src_len, (int) s->desc.width, (int) s->desc.height, s->out_fmt, s->dev_out, 0, s->rshift,
                                             s->gshift, s->bshift, s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "decode failed: %s\n", ug_hip_last_error_string());
                ug_hip_stream_sync(s->stream);
                return DECODER_NO_FRAME;
        }
        bool ok;
        if (s->out_codec == I420) { // three planes back to back; `pitch` has no meaning for it and is not used
                ok = ug_hip_memcpy_async(dst, s->dev_out, s->out_len, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) == UG_HIP_SUCCESS;
        } else { // a display pitch that differs from the packed line size (gpujpeg.c:296-319 loops over the lines on the CPU there): one 2-D copy
                ok = mi355x_download_picture(dst, (size_t) s->pitch, s->dev_out, (size_t) vc_get_linesize(s->desc.width, s->out_codec), s->desc.height,
                                             s->stream) == UG_HIP_SUCCESS;
        }
        if (ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS || !ok) {
                MSG(ERROR, "download failed: %s\n", ug_hip_last_error_string());
                return DECODER_NO_FRAME;
        }
        return DECODER_GOT_FRAME;
}

static int jpeg_mi355x_decompress_get_property(void *state, int property, void *val, size_t *len)
{
        (void) state;
        if (property == DECOMPRESS_PROPERTY_ACCEPTS_CORRUPTED_FRAME && *len >= sizeof(int)) {
                *(int *) val = false; // as gpujpeg.c:328-333
                *len = sizeof(int);
                return true;
        }
        return false;
}

static void jpeg_mi355x_decompress_done(void *state)
{
        struct state_decompress_jpeg_mi355x *s = state;
        ug_hip_set_device(s->device);
        if (s->dev_out) ug_hip_free(s->dev_out);
        if (s->dec) ug_hip_jpeg_decoder_destroy(s->dec);
        if (s->stream) ug_hip_stream_destroy(s->stream);
        free(s);
}

static int jpeg_mi355x_decompress_get_priority(codec_t compression, struct pixfmt_desc internal, codec_t ugc)
{
        (void) internal;
        if (compression != JPEG) {
                return -1;
        }
        if (ugc == VIDEO_CODEC_NONE) {
                return VDEC_PRIO_PROBE_HI; // gpujpeg.c:360-362
        }
        if (ugc == I420 || ugc == RGB || ugc == RGBA || ugc == UYVY) {
                return VDEC_PRIO_PREFERRED;
        }
        return VDEC_PRIO_NA;
}

static const struct video_decompress_info jpeg_mi355x_dec_info = {
        jpeg_mi355x_decompress_init,
        jpeg_mi355x_decompress_reconfigure,
        jpeg_mi355x_decompress,
        jpeg_mi355x_decompress_get_property,
        jpeg_mi355x_decompress_done,
        jpeg_mi355x_decompress_get_priority,
};

REGISTER_MODULE(jpeg_mi355x, &jpeg_mi355x_dec_info, LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION);
// In a build without GPUJPEG (config.h: HAVE_GPUJPEG undefined) the module also answers to the name of the one it stands in for
// (gpujpeg.c:378), so that `--param decompress=gpujpeg` -- what the reference's unit test sets, test/gpujpeg_test.cpp:64 -- finds it.
#ifndef HAVE_GPUJPEG
REGISTER_MODULE(gpujpeg, &jpeg_mi355x_dec_info, LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION);
#endif
