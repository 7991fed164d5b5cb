/* TEST INFRASTRUCTURE ONLY.  The two to_planar.h functions the reference's unit tests call (test/codec_conversions_test.cpp: uyvy_to_i420
 * through testcard_convert_buffer, y216_to_p010le directly), under the reference's own names and signature (void f(struct to_planar_data)),
 * implemented over the product's C ABI: the frame goes to the MI355X, ug_hip_to_planar() converts it, the planes come back.  Linked into
 * the test executable these definitions take the place of the ones in libugref.so, so the reference's tests run unmodified against the GPU
 * implementation.  This is also the binding INTEGRATION.md describes for callers that hold host memory. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "to_planar.h"
#include "video_codec.h"

#include "../../include/ug_mi355x.h"

static int gpu_calls;
int to_planar_gpu_shim_calls(void) { return gpu_calls; }

static void run_on_gpu(const char *name, codec_t in_codec, struct to_planar_data d, int planes, const size_t plane_rows[])
{
        const size_t in_len = (size_t) vc_get_linesize(d.width, in_codec) * d.height;
        void *dev_in = NULL, *dev_out[4] = { NULL, NULL, NULL, NULL };
        struct ug_to_planar_data g = { .width = d.width, .height = d.height };
        int ok = ug_hip_malloc(&dev_in, in_len + 64) == UG_HIP_SUCCESS && ug_hip_memcpy(dev_in, d.in_data, in_len, UG_HIP_MEMCPY_HOST_TO_DEVICE) == UG_HIP_SUCCESS;
        for (int p = 0; p < planes && ok; p++) { // the planes go up first: bytes the converter does not write stay what they were
                const size_t len = (size_t) d.out_linesize[p] * plane_rows[p];
                ok = ug_hip_malloc(&dev_out[p], len + 64) == UG_HIP_SUCCESS && ug_hip_memcpy(dev_out[p], d.out_data[p], len, UG_HIP_MEMCPY_HOST_TO_DEVICE) == UG_HIP_SUCCESS;
                g.out_data[p] = dev_out[p];
                g.out_linesize[p] = d.out_linesize[p];
        }
        g.in_data = dev_in;
        ok = ok && ug_hip_to_planar(name, &g, NULL) == UG_HIP_SUCCESS && ug_hip_stream_sync(NULL) == UG_HIP_SUCCESS;
        for (int p = 0; p < planes && ok; p++) {
                ok = ug_hip_memcpy(d.out_data[p], dev_out[p], (size_t) d.out_linesize[p] * plane_rows[p], UG_HIP_MEMCPY_DEVICE_TO_HOST) == UG_HIP_SUCCESS;
        }
        if (!ok) {
                fprintf(stderr, "to_planar_gpu_shim: %s %dx%d failed: %s\n", name, d.width, d.height, ug_hip_last_error_string());
                abort();
        }
        for (int p = 0; p < planes; p++) ug_hip_free(dev_out[p]);
        ug_hip_free(dev_in);
        gpu_calls++;
}

void uyvy_to_i420(struct to_planar_data d)
{
        const size_t rows[3] = { (size_t) d.height, ((size_t) d.height + 1) / 2, ((size_t) d.height + 1) / 2 };
        run_on_gpu("uyvy_to_i420", UYVY, d, 3, rows);
}

void y216_to_p010le(struct to_planar_data d)
{
        const size_t rows[2] = { (size_t) d.height, ((size_t) d.height + 1) / 2 };
        run_on_gpu("y216_to_p010le", Y216, d, 2, rows);
}
