/**
 * @file vdecompress_dxt_mi355x.c
 * UltraGrid video_decompress module "dxt_mi355x": DXT1 / DXT1_YUV / DXT5-YCoCg -> RGBA / RGB / UYVY on an MI355X through
 * libug_mi355x.so (include/ug_mi355x.h: ug_hip_dxt_decode).  Receiver-side counterpart of
 * vcompress_dxt_mi355x.cpp; plain C like the reference's decompress modules (video_decompress.h:74-171), same
 * callback set and conventions as src/video_decompress/dxt_glsl.c:69-249 (init / reconfigure with shifts+pitch /
 * decompress / get_property / done / priority 500 for DXT1 + DXT5 to RGBA or UYVY).
 *
 * DXT1_YUV (dxt_glsl.c:83-84) goes through the display matrix of display_dxt1_yuv_fp.glsl on the device.
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

#include "../../include/ug_mi355x.h"

#define MOD_NAME "[DXT MI355X dec] "

struct state_decompress_dxt_mi355x {
        struct video_desc desc;
        int               rshift, gshift, bshift, pitch;
        codec_t           out_codec;
        ug_dxt_t          in_fmt;
        ug_pixfmt_t       out_fmt;
        ug_hip_stream_t   stream;
        void             *dev_in, *dev_out;
        size_t            in_len, out_len;
        bool              configured;
};

static void *dxt_mi355x_decompress_init(void)
{
        struct state_decompress_dxt_mi355x *s = calloc(1, sizeof *s);
        if (ug_hip_set_device(0) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot use HIP device 0: %s\n", ug_hip_last_error_string());
                free(s);
                return NULL;
        }
        return s;
}

static void release_buffers(struct state_decompress_dxt_mi355x *s)
{
        if (s->dev_in) ug_hip_free(s->dev_in);
        if (s->dev_out) ug_hip_free(s->dev_out);
        s->dev_in = s->dev_out = NULL;
        s->configured = false;
}

static int dxt_mi355x_decompress_reconfigure(void *state, struct video_desc desc, int rshift, int gshift, int bshift,
                                             int pitch, codec_t out_codec)
{
        struct state_decompress_dxt_mi355x *s = state;
        release_buffers(s);
        if (desc.color_spec == DXT5) {
                s->in_fmt = UG_DXT5_YCOCG;
        } else if (desc.color_spec == DXT1) {
                s->in_fmt = UG_DXT1;
        } else if (desc.color_spec == DXT1_YUV) {
                s->in_fmt = UG_DXT1_YUV;
        } else {
                MSG(ERROR, "Wrong compression to decompress: %s\n", get_codec_name(desc.color_spec));
                return false;
        }
        switch (out_codec) {
        case RGBA: s->out_fmt = UG_PF_RGBA; break;
        case RGB:  s->out_fmt = UG_PF_RGB; break;
        case UYVY: s->out_fmt = UG_PF_UYVY; break;
        default:
                MSG(ERROR, "Unsupported output codec: %s\n", get_codec_name(out_codec));
                return false;
        }
        if (desc.width % 4 != 0 || desc.height % 4 != 0) {
                MSG(ERROR, "Frame size %ux%u is not a multiple of the 4x4 block\n", desc.width, desc.height);
                return false;
        }
        s->desc = desc;
        s->rshift = rshift; s->gshift = gshift; s->bshift = bshift;
        s->pitch = pitch;
        s->out_codec = out_codec;
        s->in_len = ug_hip_dxt_size(s->in_fmt, (int) desc.width, (int) desc.height);
        s->out_len = (size_t) vc_get_linesize(desc.width, out_codec) * desc.height;
        if (ug_hip_set_device(0) != UG_HIP_SUCCESS || ug_hip_malloc(&s->dev_in, s->in_len) != UG_HIP_SUCCESS ||
            ug_hip_malloc(&s->dev_out, s->out_len) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Could not allocate device buffers: %s\n", ug_hip_last_error_string());
                release_buffers(s);
                return false;
        }
        // a short (corrupted) frame uploads only the bytes that arrived and decodes the whole block grid: what did not arrive must not be
        // whatever the allocation held -- start from all-zero blocks (black), later frames leave the previous picture there
        if (ug_hip_memset_async(s->dev_in, 0, s->in_len, s->stream) != UG_HIP_SUCCESS || ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Could not clear the device input buffer: %s\n", ug_hip_last_error_string());
                release_buffers(s);
                return false;
        }
        s->configured = true;
        return true;
}

static decompress_status dxt_mi355x_decompress(void *state, unsigned char *dst, unsigned char *buffer, unsigned int src_len,
                                               int frame_seq, struct video_frame_callbacks *callbacks,
                                               struct pixfmt_desc *internal_prop)
{
        struct state_decompress_dxt_mi355x *s = state;
        (void) frame_seq, (void) callbacks, (void) internal_prop;
        if (!s->configured) {
                MSG(ERROR, "DXT decoder not configured!\n");
                return DECODER_NO_FRAME;
        }
        // accepts corrupted (short) frames, property below: decode the blocks that arrived
        const size_t n = src_len < s->in_len ? src_len : s->in_len;
        const int linesize = vc_get_linesize(s->desc.width, s->out_codec);
        if (ug_hip_set_device(0) != UG_HIP_SUCCESS ||
            ug_hip_memcpy_async(s->dev_in, buffer, n, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) != UG_HIP_SUCCESS ||
            ug_hip_dxt_decode(s->in_fmt, s->out_fmt, s->dev_in, s->dev_out, (int) s->desc.width, (int) s->desc.height, 0,
                              s->rshift, s->gshift, s->bshift, s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "decode failed: %s\n", ug_hip_last_error_string());
                return DECODER_NO_FRAME;
        }
        if (s->pitch == linesize) {
                if (ug_hip_memcpy_async(dst, s->dev_out, s->out_len, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) != UG_HIP_SUCCESS) {
                        return DECODER_NO_FRAME;
                }
        } else { // display pitch differs from the packed line size (dxt_glsl.c:163-186 does a CPU line loop here)
                for (unsigned i = 0; i < s->desc.height; i++) {
                        if (ug_hip_memcpy_async(dst + (size_t) i * s->pitch, (char *) s->dev_out + (size_t) i * linesize,
                                                linesize, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) != UG_HIP_SUCCESS) {
                                return DECODER_NO_FRAME;
                        }
                }
        }
        if (ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "stream sync failed: %s\n", ug_hip_last_error_string());
                return DECODER_NO_FRAME;
        }
        return DECODER_GOT_FRAME;
}

static int dxt_mi355x_decompress_get_property(void *state, int property, void *val, size_t *len)
{
        (void) state;
        if (property == DECOMPRESS_PROPERTY_ACCEPTS_CORRUPTED_FRAME && *len >= sizeof(int)) {
                *(int *) val = true;
                *len = sizeof(int);
                return true;
        }
        return false;
}

static void dxt_mi355x_decompress_done(void *state)
{
        struct state_decompress_dxt_mi355x *s = state;
        ug_hip_set_device(0);
        release_buffers(s);
        if (s->stream) ug_hip_stream_destroy(s->stream);
        free(s);
}

static int dxt_mi355x_decompress_get_priority(codec_t compression, struct pixfmt_desc internal, codec_t ugc)
{
        (void) internal;
        if (compression != DXT1 && compression != DXT1_YUV && compression != DXT5) { /* dxt_glsl.c:230 */
                return -1;
        }
        if (ugc != RGBA && ugc != RGB && ugc != UYVY) {
                return -1;
        }
        return 500; // same rank as dxt_glsl (video_decompress/dxt_glsl.c:228-237); force with --param decompress=dxt_mi355x
}

static const struct video_decompress_info dxt_mi355x_dec_info = {
        dxt_mi355x_decompress_init,
        dxt_mi355x_decompress_reconfigure,
        dxt_mi355x_decompress,
        dxt_mi355x_decompress_get_property,
        dxt_mi355x_decompress_done,
        dxt_mi355x_decompress_get_priority,
};

REGISTER_MODULE(dxt_mi355x, &dxt_mi355x_dec_info, LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION);
