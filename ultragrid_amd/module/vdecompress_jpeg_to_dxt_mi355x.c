/**
 * @file vdecompress_jpeg_to_dxt_mi355x.c
 * UltraGrid video_decompress module "jpeg_to_dxt_mi355x": JPEG -> DXT1 / DXT5-YCoCg on an MI355X, for receivers that display compressed
 * textures.  Counterpart of the reference's transcoder src/video_decompress/gpujpeg_to_dxt.cpp: the JPEG frame is decoded to packed RGB
 * in device memory (:136-140) and block-compressed from there with the image flipped vertically (negative height, :142-148), only the
 * DXT frame crosses PCIe on the way back (:150-156).  Same callback set and priorities (:368-373: JPEG -> DXT1 | DXT5 at 900).
 *
 * Devices and the frame rotation (gpujpeg_to_dxt.cpp:187-212,305-328): one worker thread per listed device (--param mi355x-device=<n>[:<n>...],
 * or -D, mi355x_receiver.h; a device listed twice gets two workers), frame k goes to worker k mod N, and the call that hands in frame k takes
 * frame k - (N - 1) out: N - 1 frames of delay, the frames in order -- the reference's scheme.  With the default single device N = 1: nothing
 * is delayed and no thread is made; the frame is transcoded on the caller's thread.  Two differences from the reference, on purpose:
 *  - the first N - 1 calls return DECODER_NO_FRAME (there is no picture to show yet); the reference returns DECODER_GOT_FRAME without having
 *    written the buffer (:311-314);
 *  - a worker keeps its result in device memory and the CALLING thread downloads it straight into the buffer it was given (one PCIe copy,
 *    overlapping the kernels of the frame just handed to the next worker); the reference downloads into a message and memcpy()s that.
 *
 * Any frame size (the DXT frame holds (w+3)/4 x (h+3)/4 blocks, dxt_util.h:59-67; the reference's CUDA encoder takes multiples of 4 only).
 * The block encoder follows the CUDA kernels' rounding (cuda_dxt.cu, UG_DXT_TIES_AWAY), since that is what the reference runs on this path;
 * UG_MI355X_DXT_TIES=even selects the GLSL shaders' rounding instead.
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

#define MOD_NAME "[JPEG to DXT MI355X] "

enum job_kind { JOB_NONE, JOB_FRAME, JOB_RECONFIGURE, JOB_QUIT };

/// one per listed device: its decoder, its device buffers, and -- when there are several -- its thread with a one-deep queue either way
/// (synchronized_queue<msg *, 1> m_in / m_out of gpujpeg_to_dxt.cpp:64-67)
struct transcoder {
        struct state_decompress_jpeg_to_dxt_mi355x *parent;
        int                  device;
        ug_hip_stream_t      stream;
        ug_hip_jpeg_decoder *dec;
        void                *dev_rgb, *dev_dxt;
        // the thread and its two slots
        pthread_t            thread;
        bool                 has_thread;
        pthread_mutex_t      lock;
        pthread_cond_t       cv;
        enum job_kind        in;          ///< JOB_NONE = the slot is free
        unsigned char       *in_data;     ///< the worker's own copy of the JPEG frame: the caller reuses its buffer after decompress() returns (:305-306)
        size_t               in_len, in_cap;
        bool                 has_out, out_ok;
};

struct state_decompress_jpeg_to_dxt_mi355x {
        struct video_desc    desc;
        codec_t              out_codec;
        int                  ties;
        size_t               dxt_len;
        int                  n;           ///< workers = listed devices
        struct transcoder    t[MI355X_MAX_DEVICES];
        unsigned             next;        ///< worker that takes the next frame (`free` of gpujpeg_to_dxt.cpp:108)
        unsigned             in_flight;   ///< frames handed in and not yet taken out (`occupied_count`)
};

static void jpeg_to_dxt_mi355x_decompress_done(void *state);

static void free_buffers(struct transcoder *t)
{
        if (t->dev_rgb) ug_hip_free(t->dev_rgb);
        if (t->dev_dxt) ug_hip_free(t->dev_dxt);
        t->dev_rgb = t->dev_dxt = NULL;
}

/// (re)allocates the device buffers of one worker for the parent's geometry; on the thread that owns the worker's device context
static bool transcoder_reconfigure(struct transcoder *t)
{
        const struct state_decompress_jpeg_to_dxt_mi355x *s = t->parent;
        if (ug_hip_set_device(t->device) != UG_HIP_SUCCESS) {
                return false;
        }
        free_buffers(t);
        if (ug_hip_malloc(&t->dev_rgb, (size_t) s->desc.width * s->desc.height * 3 + 64) != UG_HIP_SUCCESS ||
            ug_hip_malloc(&t->dev_dxt, s->dxt_len + 64) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Could not allocate the device buffers on device %d: %s\n", t->device, ug_hip_last_error_string());
                free_buffers(t);
                return false;
        }
        return true;
}

/// JPEG bytes -> DXT blocks in the worker's device memory (kernels finished when this returns)
static bool transcoder_frame(struct transcoder *t, const unsigned char *jpeg, size_t len)
{
        const struct state_decompress_jpeg_to_dxt_mi355x *s = t->parent;
        const int w = (int) s->desc.width, h = (int) s->desc.height;
        int fw = 0, fh = 0;
        if (ug_hip_set_device(t->device) != UG_HIP_SUCCESS || ug_hip_jpeg_read_info(jpeg, len, &fw, &fh, NULL, NULL, NULL) != UG_HIP_SUCCESS || fw != w || fh != h) {
                MSG(ERROR, "not a JPEG frame of the configured size %dx%d\n", w, h);
                return false;
        }
        // _sized: the decoder checks the frame header IT parses against the size the buffers were allocated for
        const bool ok = ug_hip_jpeg_decoder_decode_sized(t->dec, jpeg, len, w, h, UG_PF_RGB, t->dev_rgb, 3 * w, 0, 8, 16, t->stream) == UG_HIP_SUCCESS &&
                        ug_hip_dxt_encode_batch_ex(UG_PF_RGB, s->out_codec == DXT1 ? UG_DXT1 : UG_DXT5_YCOCG, t->dev_rgb, t->dev_dxt, w, -h, 3 * w, 1, 0, 0, s->ties,
                                                   t->stream) == UG_HIP_SUCCESS;
        if (!ok) MSG(ERROR, "transcoding failed: %s\n", ug_hip_last_error_string());
        return ug_hip_stream_sync(t->stream) == UG_HIP_SUCCESS && ok;
}

static void *transcoder_thread(void *arg)
{
        struct transcoder *t = arg;
        for (;;) {
                pthread_mutex_lock(&t->lock);
                while (t->in == JOB_NONE) pthread_cond_wait(&t->cv, &t->lock);
                const enum job_kind job = t->in;
                pthread_mutex_unlock(&t->lock);
                if (job == JOB_QUIT) {
                        break;
                }
                // (the in slot stays taken while the job runs: its data is read in place; the producer hands in the next job only after it
                // has taken this one's result, which is the reference's rotation)
                const bool ok = job == JOB_FRAME ? transcoder_frame(t, t->in_data, t->in_len) : transcoder_reconfigure(t);
                pthread_mutex_lock(&t->lock);
                t->in = JOB_NONE;
                t->has_out = true;
                t->out_ok = ok;
                pthread_cond_broadcast(&t->cv);
                pthread_mutex_unlock(&t->lock);
        }
        return NULL;
}

static void transcoder_push(struct transcoder *t, enum job_kind job, const unsigned char *data, size_t len)
{
        pthread_mutex_lock(&t->lock);
        while (t->in != JOB_NONE || t->has_out) pthread_cond_wait(&t->cv, &t->lock); // (never waits in the rotation: the result was taken before)
        if (job == JOB_FRAME) {
                if (len > t->in_cap) {
                        free(t->in_data);
                        t->in_data = malloc(len + len / 4);
                        t->in_cap = t->in_data ? len + len / 4 : 0;
                }
                if (t->in_data != NULL) memcpy(t->in_data, data, len);
                t->in_len = t->in_data != NULL ? len : 0; // (no memory: the frame fails in the decoder, which refuses an empty stream)
        }
        t->in = job;
        pthread_cond_broadcast(&t->cv);
        pthread_mutex_unlock(&t->lock);
}

static bool transcoder_pop(struct transcoder *t)
{
        pthread_mutex_lock(&t->lock);
        while (!t->has_out) pthread_cond_wait(&t->cv, &t->lock);
        t->has_out = false;
        const bool ok = t->out_ok;
        pthread_cond_broadcast(&t->cv);
        pthread_mutex_unlock(&t->lock);
        return ok;
}

static void *jpeg_to_dxt_mi355x_decompress_init(void)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = calloc(1, sizeof *s);
        if (s == NULL) {
                return NULL;
        }
        const char *ties = getenv("UG_MI355X_DXT_TIES");
        s->ties = ties != NULL && strcmp(ties, "even") == 0 ? UG_DXT_TIES_EVEN : UG_DXT_TIES_AWAY;
        int devs[MI355X_MAX_DEVICES];
        bool bad = false;
        const int n = mi355x_receiver_devices(devs, MI355X_MAX_DEVICES, &bad);
        if (bad) {
                MSG(ERROR, "--param " MI355X_DEVICE_PARAM "=%s: expected <n>[:<n>...]\n", get_commandline_param(MI355X_DEVICE_PARAM));
                free(s);
                return NULL;
        }
        for (int i = 0; i < n; i++) {
                struct transcoder *t = &s->t[i];
                t->parent = s;
                t->device = devs[i];
                pthread_mutex_init(&t->lock, NULL);
                pthread_cond_init(&t->cv, NULL);
                s->n = i + 1;
                if (ug_hip_set_device(t->device) != UG_HIP_SUCCESS || ug_hip_stream_create(&t->stream) != UG_HIP_SUCCESS ||
                    ug_hip_jpeg_decoder_create(&t->dec) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "cannot set up the decoder on HIP device %d: %s\n", t->device, ug_hip_last_error_string());
                        jpeg_to_dxt_mi355x_decompress_done(s); // releases what was made so far
                        return NULL;
                }
        }
        for (int i = 0; i < n && n > 1; i++) {
                if (pthread_create(&s->t[i].thread, NULL, transcoder_thread, &s->t[i]) != 0) {
                        MSG(ERROR, "cannot start a worker thread\n");
                        jpeg_to_dxt_mi355x_decompress_done(s);
                        return NULL;
                }
                s->t[i].has_thread = true;
        }
        if (n > 1) MSG(NOTICE, "%d workers: frames come out %d frame(s) after they went in\n", n, n - 1);
        return s;
}

/// drops the frames that are still on their way (flush() of gpujpeg_to_dxt.cpp:206-218)
static void flush(struct state_decompress_jpeg_to_dxt_mi355x *s)
{
        while (s->in_flight > 0) {
                (void) transcoder_pop(&s->t[(s->next + (unsigned) s->n - s->in_flight) % (unsigned) s->n]);
                s->in_flight--;
        }
        s->next = 0;
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
        const int ppb = out_codec == DXT1 ? 2 : 1; // pixels per byte (gpujpeg_to_dxt.cpp:239-243)
        if (pitch != (int) desc.width / ppb && pitch != vc_get_linesize(desc.width, out_codec)) { // (an odd width: the reference's assert knows width / ppb only, :244)
                MSG(ERROR, "a DXT frame has no other pitch than width / %d\n", ppb);
                return false;
        }
        flush(s);
        s->desc = desc;
        s->out_codec = out_codec;
        s->dxt_len = ug_hip_dxt_size(out_codec == DXT1 ? UG_DXT1 : UG_DXT5_YCOCG, (int) desc.width, (int) desc.height); // whole blocks (dxt_util.h:59-67)
        if (s->dxt_len == 0) {
                MSG(ERROR, "not a frame size: %ux%u\n", desc.width, desc.height);
                return false;
        }
        bool ok = true;
        for (int i = 0; i < s->n; i++) { // every worker, one after the other, as :252-266
                if (s->n == 1) {
                        ok = transcoder_reconfigure(&s->t[i]);
                } else {
                        transcoder_push(&s->t[i], JOB_RECONFIGURE, NULL, 0);
                        ok = transcoder_pop(&s->t[i]) && ok;
                }
        }
        return ok;
}

static decompress_status jpeg_to_dxt_mi355x_decompress(void *state, unsigned char *dst, unsigned char *buffer, unsigned int src_len,
                                                       int frame_seq, struct video_frame_callbacks *callbacks,
                                                       struct pixfmt_desc *internal_prop)
{
        struct state_decompress_jpeg_to_dxt_mi355x *s = state;
        (void) frame_seq, (void) callbacks, (void) internal_prop;
        struct transcoder *out = &s->t[0];
        bool ok;
        if (s->n == 1) { // one device: here and now, on the caller's thread
                ok = transcoder_frame(out, buffer, src_len);
        } else {
                transcoder_push(&s->t[s->next], JOB_FRAME, buffer, src_len);
                s->next = (s->next + 1) % (unsigned) s->n;
                if (s->in_flight < (unsigned) s->n - 1) { // the pipeline is filling (:308-311)
                        s->in_flight++;
                        return DECODER_NO_FRAME;
                }
                out = &s->t[s->next]; // the oldest frame on its way: the worker that takes the NEXT one (:313-319)
                ok = transcoder_pop(out);
        }
        if (!ok) {
                return DECODER_NO_FRAME;
        }
        // the result lies in the worker's device memory: straight into the caller's buffer (the worker is idle until the next call hands it a frame)
        if (ug_hip_set_device(out->device) != UG_HIP_SUCCESS ||
            ug_hip_memcpy_async(dst, out->dev_dxt, s->dxt_len, UG_HIP_MEMCPY_DEVICE_TO_HOST, out->stream) != UG_HIP_SUCCESS ||
            ug_hip_stream_sync(out->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "download failed: %s\n", ug_hip_last_error_string());
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
        flush(s);
        for (int i = 0; i < s->n; i++) {
                struct transcoder *t = &s->t[i];
                if (t->has_thread) {
                        transcoder_push(t, JOB_QUIT, NULL, 0);
                        pthread_join(t->thread, NULL);
                }
                ug_hip_set_device(t->device);
                free_buffers(t);
                if (t->dec) ug_hip_jpeg_decoder_destroy(t->dec);
                if (t->stream) ug_hip_stream_destroy(t->stream);
                free(t->in_data);
                pthread_cond_destroy(&t->cv);
                pthread_mutex_destroy(&t->lock);
        }
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
