/**
 * @file vdecompress_jpeg_to_dxt_mi355x.c
 * UltraGrid video_decompress module "jpeg_to_dxt_mi355x": JPEG -> DXT1 / DXT5-YCoCg on an MI355X, for receivers that display compressed
 * textures.  Counterpart of the reference's transcoder src/video_decompress/gpujpeg_to_dxt.cpp: the JPEG frame is decoded to packed RGB
 * in device memory (:136-140) and block-compressed from there with the image flipped vertically (negative height, :142-148), only the
 * DXT frame crosses PCIe on the way back (:150-156).  Same callback set and priorities (:368-373: JPEG -> DXT1 | DXT5 at 900).
 *
 * What differs, on purpose:
 *  - one device (HIP device 0), synchronous: the reference round-robins frames over its CUDA devices and returns frame N while N+1.. are
 *    in flight; a decode + encode is well under a millisecond here, so nothing is gained by delaying the frame;
 *  - the block encoder follows the CUDA kernels' rounding (cuda_dxt.cu, UG_DXT_TIES_AWAY), since that is what the reference runs on
 *    this path; UG_MI355X_DXT_TIES=even selects the GLSL shaders' rounding instead.
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

#define MOD_NAME "[JPEG to DXT MI355X] "

struct state_decompress_jpeg_to_dxt_mi355x {
        struct video_desc    desc;
        codec_t              out_codec;
        int                  ties;
        ug_hip_stream_t      stream;
        ug_hip_jpeg_decoder *dec;
        void                *dev_rgb, *dev_dxt;
        size_t               dxt_len;
};

static void jpeg_to_dxt_mi355x_decompress_done(void *state);

static void *jpeg_to_dxt_mi355x_decompress_init(void)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = calloc(1, sizeof *s);
        if (s == NULL) {
                return NULL;
        }
        const char *ties = getenv("UG_MI355X_DXT_TIES");
        s->ties = ties != NULL && strcmp(ties, "even") == 0 ? UG_DXT_TIES_EVEN : UG_DXT_TIES_AWAY;
        if (ug_hip_set_device(0) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS ||
            ug_hip_jpeg_decoder_create(&s->dec) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot set up the decoder on HIP device 0: %s\n", ug_hip_last_error_string());
                jpeg_to_dxt_mi355x_decompress_done(s); // releases whichever of the two was made
                return NULL;
        }
        return s;
}

static void free_buffers(struct state_decompress_jpeg_to_dxt_mi355x *s)
{
        if (s->dev_rgb) ug_hip_free(s->dev_rgb);
        if (s->dev_dxt) ug_hip_free(s->dev_dxt);
        s->dev_rgb = s->dev_dxt = NULL;
}

static int jpeg_to_dxt_mi355x_decompress_reconfigure(void *state, struct video_desc desc, int rshift, int gshift, int bshift, int pitch,
                                                     codec_t out_codec)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = state;
        (void) rshift, (void) gshift, (void) bshift;
        if (desc.color_spec != JPEG || (out_codec != DXT1 && out_codec != DXT5)) {
                MSG(ERROR, "only JPEG -> DXT1 / DXT5, not %s -> %s\n", get_codec_name(desc.color_spec), get_codec_name(out_codec));
                return false;
        }
        if (desc.width % 4 != 0 || desc.height % 4 != 0) { // the block encoder's requirement (cuda_dxt.cu:745)
                MSG(ERROR, "the frame size must be a multiple of 4 in both directions, not %ux%u\n", desc.width, desc.height);
                return false;
        }
        const int ppb = out_codec == DXT1 ? 2 : 1; // pixels per byte (gpujpeg_to_dxt.cpp:239-243)
        if (pitch != (int) desc.width / ppb) {
                MSG(ERROR, "a DXT frame has no other pitch than width / %d\n", ppb);
                return false;
        }
        ug_hip_set_device(0);
        free_buffers(s);
        s->desc = desc;
        s->out_codec = out_codec;
        s->dxt_len = (size_t) desc.width * desc.height / ppb;
        if (ug_hip_malloc(&s->dev_rgb, (size_t) desc.width * desc.height * 3 + 64) != UG_HIP_SUCCESS ||
            ug_hip_malloc(&s->dev_dxt, s->dxt_len + 64) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Could not allocate the device buffers: %s\n", ug_hip_last_error_string());
                free_buffers(s);
                return false;
        }
        return true;
}

static decompress_status jpeg_to_dxt_mi355x_decompress(void *state, unsigned char *dst, unsigned char *buffer, unsigned int src_len,
                                                       int frame_seq, struct video_frame_callbacks *callbacks,
                                                       struct pixfmt_desc *internal_prop)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = state;
        (void) frame_seq, (void) callbacks, (void) internal_prop;
        int w = 0, h = 0;
        if (ug_hip_set_device(0) != UG_HIP_SUCCESS || ug_hip_jpeg_read_info(buffer, src_len, &w, &h, NULL, NULL, NULL) != UG_HIP_SUCCESS ||
            (unsigned) w != s->desc.width || (unsigned) h != s->desc.height) {
                MSG(ERROR, "not a JPEG frame of the configured size %ux%u\n", s->desc.width, s->desc.height);
                return DECODER_NO_FRAME;
        }
        // _sized: the decoder checks the frame header IT parses against the size the buffers were allocated for
        bool ok = ug_hip_jpeg_decoder_decode_sized(s->dec, buffer, src_len, (int) s->desc.width, (int) s->desc.height, UG_PF_RGB, s->dev_rgb, 3 * w, 0, 8, 16,
                                                   s->stream) == UG_HIP_SUCCESS &&
                  ug_hip_dxt_encode_batch_ex(UG_PF_RGB, s->out_codec == DXT1 ? UG_DXT1 : UG_DXT5_YCOCG, s->dev_rgb, s->dev_dxt, w, -h, 3 * w, 1, 0, 0,
                                             s->ties, s->stream) == UG_HIP_SUCCESS &&
                  ug_hip_memcpy_async(dst, s->dev_dxt, s->dxt_len, UG_HIP_MEMCPY_DEVICE_TO_HOST, s->stream) == UG_HIP_SUCCESS;
        if (!ok) MSG(ERROR, "transcoding failed: %s\n", ug_hip_last_error_string());
        if (ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS || !ok) {
                return DECODER_NO_FRAME;
        }
        return DECODER_GOT_FRAME;
}

static int jpeg_to_dxt_mi355x_decompress_get_property(void *state, int property, void *val, size_t *len)
{
        (void) state, (void) property, (void) val, (void) len;
        return false; // gpujpeg_to_dxt.cpp:346-353
}

static void jpeg_to_dxt_mi355x_decompress_done(void *state)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = state;
        if (!s) return;
        ug_hip_set_device(0);
        free_buffers(s);
        if (s->dec) ug_hip_jpeg_decoder_destroy(s->dec);
        if (s->stream) ug_hip_stream_destroy(s->stream);
        free(s);
}

static int jpeg_to_dxt_mi355x_decompress_get_priority(codec_t compression, struct pixfmt_desc internal, codec_t ugc)
{
        (void) internal;
        return compression == JPEG && (ugc == DXT1 || ugc == DXT5) ? 900 : -1;
}

static const struct video_decompress_info jpeg_to_dxt_mi355x_info = {
        jpeg_to_dxt_mi355x_decompress_init,
        jpeg_to_dxt_mi355x_decompress_reconfigure,
        jpeg_to_dxt_mi355x_decompress,
        jpeg_to_dxt_mi355x_decompress_get_property,
        jpeg_to_dxt_mi355x_decompress_done,
        jpeg_to_dxt_mi355x_decompress_get_priority,
};

REGISTER_MODULE(jpeg_to_dxt_mi355x, &jpeg_to_dxt_mi355x_info, LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION);
