/**
 * @file vdecompress_dxt_mi355x.c
 * UltraGrid video_decompress module "dxt_mi355x": DXT1 / DXT1_YUV / DXT5-YCoCg -> RGBA / RGB / UYVY on an MI355X through
 * libug_mi355x.so (include/ug_mi355x.h: ug_hip_dxt_decode).  Receiver-side counterpart of
 * vcompress_dxt_mi355x.cpp; plain C like the reference's decompress modules (video_decompress.h:74-171), same
 * callback set and conventions as src/video_decompress/dxt_glsl.c:69-249 (init / reconfigure with shifts+pitch /
 * decompress / get_property / done / priority 500 for DXT1 + DXT5 to RGBA or UYVY).
 *
 * DXT1_YUV (dxt_glsl.c:83-84) goes through the display matrix of display_dxt1_yuv_fp.glsl on the device.
 *
 * Any frame size (dxt_glsl.c:95-98 -> dxt_decoder.c:146-149).  --param mi355x-device / -D pick the GPU, --param mi355x-bands=<k> the number of
 * row bands a frame is pipelined in (mi355x_receiver.h); a display pitch that differs from the line size costs nothing (one 2-D copy per band).
 */
#include <pthread.h>
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

#define MOD_NAME "[DXT MI355X dec] "
#define MI355X_MAX_BANDS 16

struct state_decompress_dxt_mi355x {
        struct video_desc desc;
        int               rshift, gshift, bshift, pitch;
        codec_t           out_codec;
        ug_dxt_t          in_fmt;
        ug_pixfmt_t       out_fmt;
        int               device; ///< --param mi355x-device / -D (mi355x_receiver.h)
        int               bands_param; ///< --param mi355x-bands (MI355X_AUTO_BANDS: chosen at reconfigure by the size of the frame)
        int               bands;
        ug_hip_stream_t   stream;
        void             *dev_in, *dev_out;
        size_t            in_len, out_len;
        bool              configured;
        // the band pipeline's second thread (see decode_in_bands): downloads band k while the caller's thread uploads band k + 1
        ug_hip_stream_t   down;                 ///< the downloader's stream
        ug_hip_event_t    band_done[MI355X_MAX_BANDS]; ///< recorded behind the decoder of band k
        pthread_t         downloader;
        bool              has_downloader;
        pthread_mutex_t   lock;
        pthread_cond_t    cv;
        struct band_job {
                unsigned char *dst;
                int            r0[MI355X_MAX_BANDS + 1]; ///< band k = lines r0[k] .. r0[k + 1]
                int            bands;
        } job;
        int               issued;               ///< bands whose decoder is queued (caller -> downloader); -1 = no frame in work
        int               downloaded;           ///< bands copied out (downloader -> caller)
        bool              failed, quit;
};

static unsigned dxt_mi355x_state_count; // the states of this process take the listed devices in turn

static void *dxt_mi355x_decompress_init(void)
{
        struct state_decompress_dxt_mi355x *s = calloc(1, sizeof *s);
        if (s == NULL) {
                return NULL;
        }
        s->device = mi355x_next_state_device(&dxt_mi355x_state_count, MOD_NAME);
        s->bands_param = mi355x_receiver_bands(MI355X_AUTO_BANDS);
        pthread_mutex_init(&s->lock, NULL);
        pthread_cond_init(&s->cv, NULL);
        if (s->device < 0 || ug_hip_set_device(s->device) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS) {
                if (s->device >= 0) MSG(ERROR, "cannot use HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                pthread_cond_destroy(&s->cv);
                pthread_mutex_destroy(&s->lock);
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
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot use HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                return false;
        }
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
        // Any frame size, as dxt_decoder_create takes it (dxt_glsl.c:95-98 -> dxt_decoder.c:146-149: the stream holds whole blocks, the
        // picture is width x height); a 4:2:2 line is made of pixel pairs (rgba_to_yuv422.glsl renders width / 2 texels)
        if (out_codec == UYVY && desc.width % 2 != 0) {
                MSG(ERROR, "A UYVY picture %u pixels wide is not made of pixel pairs\n", desc.width);
                return false;
        }
        if (pitch < vc_get_linesize(desc.width, out_codec)) {
                MSG(ERROR, "pitch %d is shorter than a line of %u %s pixels\n", pitch, desc.width, get_codec_name(out_codec));
                return false;
        }
        s->desc = desc;
        s->rshift = rshift; s->gshift = gshift; s->bshift = bshift;
        s->pitch = pitch;
        s->out_codec = out_codec;
        s->in_len = ug_hip_dxt_size(s->in_fmt, (int) desc.width, (int) desc.height); // dxt_get_size: (w+3)/4 x (h+3)/4 blocks
        s->out_len = (size_t) vc_get_linesize(desc.width, out_codec) * desc.height;
        // How many bands pay (profiles/r06_receiver.txt, fps against the frame's two copies alone one after the other): 8K DXT5 -> UYVY 0.96 of it on
        // one thread, 1.10 with 2 bands, 1.15-1.16 with 4 or 8; 4K DXT5 -> UYVY 0.93 / 1.00 / 0.94-0.99 / 0.82-0.95; below that the hand-over
        // between the two threads costs more than the overlap gives.
        const size_t traffic = s->in_len + s->out_len;
        s->bands = s->bands_param != MI355X_AUTO_BANDS ? s->bands_param : (traffic >= ((size_t) 64 << 20) ? 4 : (traffic >= ((size_t) 16 << 20) ? 2 : 1));
        if (s->in_len == 0 || ug_hip_malloc(&s->dev_in, s->in_len) != UG_HIP_SUCCESS || ug_hip_malloc(&s->dev_out, s->out_len + 64) != UG_HIP_SUCCESS) {
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

/// The band pipeline (--param mi355x-bands=<k>, k > 1).  A copy from or to PAGEABLE host memory -- what decompress() is handed -- blocks the thread
/// that issues it until the bytes are across (hipMemcpyAsync returns after 0.596 of 0.597 ms for a 4K RGBA picture, profiles/r06_copy_probe.txt),
/// so one thread alone never has an upload and a download in flight together, whatever streams it uses: cut into bands on one thread a frame only
/// got slower (profiles/r06_receiver.txt).  Hence two threads: the caller's uploads band k + 1 and queues its decoder while this one waits (on its
/// own stream, through an event) for the decoder of band k and copies that band out.  DXT blocks do not see their neighbours: the bytes are those
/// of the whole-frame call.
static void *downloader_thread(void *arg)
{
        struct state_decompress_dxt_mi355x *s = arg;
        pthread_mutex_lock(&s->lock);
        for (;;) {
                while (!s->quit && (s->issued < 0 || s->downloaded >= s->issued)) pthread_cond_wait(&s->cv, &s->lock);
                if (s->quit) {
                        break;
                }
                const int k = s->downloaded;
                const struct band_job job = s->job;
                pthread_mutex_unlock(&s->lock);
                const size_t ls = (size_t) vc_get_linesize(s->desc.width, s->out_codec);
                const bool ok = ug_hip_set_device(s->device) == UG_HIP_SUCCESS && ug_hip_stream_wait_event(s->down, s->band_done[k]) == UG_HIP_SUCCESS &&
                                mi355x_download_picture(job.dst + (size_t) job.r0[k] * s->pitch, (size_t) s->pitch, (char *) s->dev_out + (size_t) job.r0[k] * ls, ls,
                                                        (size_t) (job.r0[k + 1] - job.r0[k]), s->down) == UG_HIP_SUCCESS &&
                                (k + 1 < job.bands || ug_hip_stream_sync(s->down) == UG_HIP_SUCCESS); // (the last band: the picture is complete when this returns)
                pthread_mutex_lock(&s->lock);
                if (!ok) s->failed = true;
                s->downloaded = k + 1;
                pthread_cond_broadcast(&s->cv);
        }
        pthread_mutex_unlock(&s->lock);
        return NULL;
}

static bool start_downloader(struct state_decompress_dxt_mi355x *s)
{
        if (s->has_downloader) {
                return true;
        }
        // (what a failed earlier attempt left behind is kept and used: done() releases it)
        if (s->down == NULL && ug_hip_stream_create(&s->down) != UG_HIP_SUCCESS) {
                return false;
        }
        for (int k = 0; k < MI355X_MAX_BANDS; k++) {
                if (s->band_done[k] == NULL && ug_hip_event_create(&s->band_done[k]) != UG_HIP_SUCCESS) return false;
        }
        s->issued = -1;
        if (pthread_create(&s->downloader, NULL, downloader_thread, s) != 0) {
                return false;
        }
        s->has_downloader = true;
        return true;
}

/// Band edges lie on multiples of 16 lines (4 block rows): every band's block and line addresses stay 16-byte aligned wherever the output's line
/// size is a multiple of 16; pictures whose lines are not (a width that is not a multiple of 4) are not cut, nor are small ones.
static bool decode_in_bands(struct state_decompress_dxt_mi355x *s, unsigned char *dst, const unsigned char *buffer, size_t n)
{
        const int w = (int) s->desc.width, h = (int) s->desc.height;
        const size_t linesize = (size_t) vc_get_linesize(s->desc.width, s->out_codec);
        const size_t block_row = s->in_len / (size_t) ((h + 3) / 4); // bytes of one row of blocks
        const bool cuttable = (linesize & 15) == 0 && h >= 64 && s->bands > 1;
        if (!cuttable || !start_downloader(s)) { // one after the other, on the caller's thread and the state's stream
                const bool ok = ug_hip_memcpy_async(s->dev_in, buffer, n, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) == UG_HIP_SUCCESS &&
                                ug_hip_dxt_decode(s->in_fmt, s->out_fmt, s->dev_in, s->dev_out, w, h, 0, s->rshift, s->gshift, s->bshift, s->stream) == UG_HIP_SUCCESS &&
                                mi355x_download_picture(dst, (size_t) s->pitch, s->dev_out, linesize, (size_t) h, s->stream) == UG_HIP_SUCCESS;
                if (!ok) MSG(ERROR, "decode failed: %s\n", ug_hip_last_error_string());
                return ok;
        }
        struct band_job job = { .dst = dst, .bands = 0 };
        for (int k = 0, r0 = 0; k < s->bands && r0 < h; k++) {
                const int cut = (int) (((long) h * (k + 1) / s->bands + 15) / 16 * 16);
                const int r1 = k == s->bands - 1 || cut > h ? h : cut;
                if (r1 > r0) {
                        job.r0[job.bands++] = r0;
                        r0 = r1;
                }
        }
        job.r0[job.bands] = h;
        pthread_mutex_lock(&s->lock);
        s->job = job;
        s->issued = 0;
        s->downloaded = 0;
        s->failed = false;
        pthread_mutex_unlock(&s->lock);
        bool ok = true;
        for (int k = 0; k < job.bands && ok; k++) {
                const int r0 = job.r0[k], r1 = job.r0[k + 1];
                const size_t off = (size_t) (r0 / 4) * block_row, end = (size_t) ((r1 + 3) / 4) * block_row;
                char *const blocks = (char *) s->dev_in + off;
                if (off < n) { // (a short frame: the bands past its end keep what the buffer held -- the previous picture, or black)
                        ok = ug_hip_memcpy_async(blocks, buffer + off, (end < n ? end : n) - off, UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) == UG_HIP_SUCCESS;
                }
                ok = ok && ug_hip_dxt_decode(s->in_fmt, s->out_fmt, blocks, (char *) s->dev_out + (size_t) r0 * linesize, w, r1 - r0, 0, s->rshift, s->gshift, s->bshift,
                                             s->stream) == UG_HIP_SUCCESS &&
                     ug_hip_event_record(s->band_done[k], s->stream) == UG_HIP_SUCCESS;
                if (ok) {
                        pthread_mutex_lock(&s->lock);
                        s->issued = k + 1;
                        pthread_cond_broadcast(&s->cv);
                        pthread_mutex_unlock(&s->lock);
                }
        }
        if (!ok) MSG(ERROR, "decode failed: %s\n", ug_hip_last_error_string());
        // wait for the bands that were handed over (all of them, or those before the failure): the downloader must be idle before dst goes back
        pthread_mutex_lock(&s->lock);
        while (s->downloaded < s->issued) pthread_cond_wait(&s->cv, &s->lock);
        ok = ok && !s->failed;
        s->issued = -1;
        pthread_mutex_unlock(&s->lock);
        return ok;
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
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot use HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                return DECODER_NO_FRAME;
        }
        // accepts corrupted (short) frames, property below: decode the blocks that arrived
        const size_t n = src_len < s->in_len ? src_len : s->in_len;
        const bool ok = decode_in_bands(s, dst, buffer, n);
        if (ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "stream sync failed: %s\n", ug_hip_last_error_string());
                return DECODER_NO_FRAME;
        }
        return ok ? DECODER_GOT_FRAME : DECODER_NO_FRAME;
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
        ug_hip_set_device(s->device);
        if (s->has_downloader) {
                pthread_mutex_lock(&s->lock);
                s->quit = true;
                pthread_cond_broadcast(&s->cv);
                pthread_mutex_unlock(&s->lock);
                pthread_join(s->downloader, NULL);
        }
        for (int k = 0; k < MI355X_MAX_BANDS; k++) ug_hip_event_destroy(s->band_done[k]);
        if (s->down) ug_hip_stream_destroy(s->down);
        release_buffers(s);
        if (s->stream) ug_hip_stream_destroy(s->stream);
        pthread_cond_destroy(&s->cv);
        pthread_mutex_destroy(&s->lock);
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
