/**
 * @file mi355x_receiver.h
 * What the three MI355X video_decompress modules (dxt_mi355x, jpeg_mi355x, jpeg_to_dxt_mi355x) share: which GPU a decompress state
 * runs on, and how a decoded picture gets back into the caller's buffer.  Plain C.
 *
 * Device choice.  The reference's receivers take the devices of `-D / --cuda-device <i>[,<i>...]` (host.cpp:177-179 cuda_devices[],
 * used by src/video_decompress/gpujpeg.c:162 and gpujpeg_to_dxt.cpp:187-212); here, in this order:
 *   --param mi355x-device=<n>[:<n>...]   (':' or '+' between the numbers: ',' separates --param entries, host.cpp:1098-1100)
 *   -D <n>[,<n>...]                      when given explicitly (cuda_devices_explicit)
 *   device 0.
 * The states of one process take the listed devices in turn (an atomic counter per module): the receiver makes one decompress state per
 * tile (decompress_init_multi, rtp/video_decoders.cpp:590-612) and decodes the tiles of a frame side by side, so with two devices listed
 * a two-tile stream uses both.  The JPEG -> DXT transcoder instead rotates FRAMES over the whole list, as the reference's does.
 *
 * Download.  A display pitch that differs from the packed line size is served by ONE 2-D copy (ug_hip_memcpy_2d_async): the reference does
 * a CPU memcpy loop there (dxt_glsl.c:163-186, gpujpeg.c:305-315); one copy call per line through the runtime measures 1-3 GB/s
 * (profiles/r06_copy_probe.txt) against 55 GB/s for the 2-D copy.
 */
#ifndef MI355X_RECEIVER_H
#define MI355X_RECEIVER_H

#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

#include "debug.h"
#include "host.h"

#include "../../include/ug_mi355x.h"

#define MI355X_DEVICE_PARAM "mi355x-device"
#define MI355X_BANDS_PARAM  "mi355x-bands"
#define MI355X_MAX_DEVICES  64
#define MI355X_AUTO_BANDS 0 /* no --param mi355x-bands: the module chooses by the size of the frame */

ADD_TO_PARAM(MI355X_DEVICE_PARAM, "* " MI355X_DEVICE_PARAM "=<n>[:<n>...]\n"
                                  "  GPU(s) of the MI355X decompress modules (dxt_mi355x, jpeg_mi355x: the states take them in turn;\n"
                                  "  jpeg_to_dxt_mi355x: frames rotate over them, each listed device adds one frame of delay). Default: -D, else 0.\n");
ADD_TO_PARAM(MI355X_BANDS_PARAM, "* " MI355X_BANDS_PARAM "=<k>\n"
                                 "  dxt_mi355x: a frame is uploaded, decoded and downloaded as k row bands, upload and download on two threads\n"
                                 "  (default: by frame size -- 1 below 16 MiB in + out, 2 up to 64 MiB (4K), 4 above (8K); 1 = one after the other on the caller's thread).\n");

/* the reference's own device list (host.cpp:177-179); weak: a host that does not have it (the test harnesses) simply has no -D */
extern unsigned int cuda_devices[] __attribute__((weak));
extern unsigned int cuda_devices_count __attribute__((weak));
extern bool cuda_devices_explicit __attribute__((weak));

/** "<n>[:<n>...]" ('+' and ',' are taken too) -> devs[]; returns the count, 0 for an empty / malformed / too long list */
static inline int mi355x_parse_device_list(const char *s, int *devs, int max)
{
        int n = 0;
        if (s == NULL || *s == '\0') {
                return 0;
        }
        for (const char *p = s;;) {
                char *end = NULL;
                const long v = strtol(p, &end, 10);
                if (end == p || v < 0 || v >= 1024 || n == max) {
                        return 0;
                }
                devs[n++] = (int) v;
                if (*end == '\0') {
                        return n;
                }
                if (*end != ':' && *end != '+' && *end != ',') {
                        return 0;
                }
                p = end + 1;
        }
}

/** the device list of this process for the decompress modules (see the file comment); always >= 1 entry; *bad = the parameter was given but is not a list */
static inline int mi355x_receiver_devices(int *devs, int max, bool *bad)
{
        const char *p = get_commandline_param(MI355X_DEVICE_PARAM);
        if (bad) *bad = false;
        if (p != NULL) {
                const int n = mi355x_parse_device_list(p, devs, max);
                if (n > 0) {
                        return n;
                }
                if (bad) *bad = true;
        } else if (&cuda_devices_explicit != NULL && &cuda_devices_count != NULL && cuda_devices_explicit && cuda_devices_count > 0) {
                int n = 0;
                for (unsigned i = 0; i < cuda_devices_count && n < max; i++) devs[n++] = (int) cuda_devices[i];
                return n;
        }
        devs[0] = 0;
        return 1;
}

/** the device of the next decompress state of a module: the listed devices in turn; -1 (and a message) for a bad list */
static inline int mi355x_next_state_device(unsigned *counter, const char *mod_name)
{
        int devs[MI355X_MAX_DEVICES];
        bool bad = false;
        const int n = mi355x_receiver_devices(devs, MI355X_MAX_DEVICES, &bad);
        if (bad) {
                log_msg(LOG_LEVEL_ERROR, "%s--param " MI355X_DEVICE_PARAM "=%s: expected <n>[:<n>...]\n", mod_name, get_commandline_param(MI355X_DEVICE_PARAM));
                return -1;
        }
        return devs[__atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED) % (unsigned) n];
}

/** --param mi355x-bands=<k>, 1..16 (default `dflt`, e.g. MI355X_AUTO_BANDS) */
static inline int mi355x_receiver_bands(int dflt)
{
        const char *p = get_commandline_param(MI355X_BANDS_PARAM);
        if (p == NULL) {
                return dflt;
        }
        const int k = atoi(p);
        return k < 1 ? 1 : (k > 16 ? 16 : k);
}

/** device picture (lines `linesize` apart) -> the caller's buffer with its display pitch: one contiguous copy, or one 2-D copy */
static inline int mi355x_download_picture(void *dst, size_t pitch, const void *src_dev, size_t linesize, size_t rows, ug_hip_stream_t stream)
{
        if (pitch == linesize) {
                return ug_hip_memcpy_async(dst, src_dev, linesize * rows, UG_HIP_MEMCPY_DEVICE_TO_HOST, stream);
        }
        return ug_hip_memcpy_2d_async(dst, pitch, src_dev, linesize, linesize, rows, UG_HIP_MEMCPY_DEVICE_TO_HOST, stream);
}

#endif /* MI355X_RECEIVER_H */
