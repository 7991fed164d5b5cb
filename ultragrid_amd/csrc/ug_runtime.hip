// ug_runtime.hip -- runtime shim of libug_mi355x.so: device select, device / pinned-host
// allocation, copies, streams, last-error text.  Own entry points for what UltraGrid's
// modules get from src/cuda_wrapper.h:50-76 (that shim is CUDA-only and is not hipified).
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "ug_common.h"

namespace ug {
static thread_local char g_err[256] = "no error";
void set_last_error(hipError_t e, const char *what)
{
        snprintf(g_err, sizeof g_err, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
}
void set_last_error_msg(const char *msg)
{
        snprintf(g_err, sizeof g_err, "%s", msg);
}
} // namespace ug

extern "C" {

int ug_hip_abi_version(void) { return UG_HIP_ABI_VERSION; }

const char *ug_hip_last_error_string(void) { return ug::g_err; }

int ug_hip_device_count(int *count)
{
        if (!count) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipGetDeviceCount(count));
        return UG_HIP_SUCCESS;
}

int ug_hip_pointer_is_device(const void *ptr)
{
        hipPointerAttribute_t a;
        if (!ptr || hipPointerGetAttributes(&a, ptr) != hipSuccess) {
                (void) hipGetLastError(); // plain malloc memory is "invalid value" to the runtime: not an error here
                return 0;
        }
        return a.type == hipMemoryTypeDevice ? 1 : 0;
}

int ug_hip_pointer_device(const void *ptr)
{
        hipPointerAttribute_t a;
        if (!ptr || hipPointerGetAttributes(&a, ptr) != hipSuccess) {
                (void) hipGetLastError();
                return -1;
        }
        return a.type == hipMemoryTypeDevice ? a.device : -1;
}

int ug_hip_set_device(int index)
{
        UG_HIP_TRY(hipSetDevice(index));
        return UG_HIP_SUCCESS;
}

int ug_hip_malloc(void **buffer, size_t size)
{
        if (!buffer) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipMalloc(buffer, size));
        return UG_HIP_SUCCESS;
}

int ug_hip_free(void *buffer)
{
        UG_HIP_TRY(hipFree(buffer));
        return UG_HIP_SUCCESS;
}

int ug_hip_malloc_host(void **buffer, size_t size)
{
        if (!buffer) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipHostMalloc(buffer, size, hipHostMallocDefault));
        return UG_HIP_SUCCESS;
}

int ug_hip_free_host(void *buffer)
{
        UG_HIP_TRY(hipHostFree(buffer));
        return UG_HIP_SUCCESS;
}

static hipMemcpyKind kind_of(int kind)
{
        switch (kind) {
        case UG_HIP_MEMCPY_HOST_TO_DEVICE: return hipMemcpyHostToDevice;
        case UG_HIP_MEMCPY_DEVICE_TO_HOST: return hipMemcpyDeviceToHost;
        case UG_HIP_MEMCPY_DEVICE_TO_DEVICE: return hipMemcpyDeviceToDevice;
        }
        return hipMemcpyDefault;
}

int ug_hip_memcpy(void *dst, const void *src, size_t count, int kind)
{
        UG_HIP_TRY(hipMemcpy(dst, src, count, kind_of(kind)));
        return UG_HIP_SUCCESS;
}

int ug_hip_memcpy_async(void *dst, const void *src, size_t count, int kind, ug_hip_stream_t stream)
{
        UG_HIP_TRY(hipMemcpyAsync(dst, src, count, kind_of(kind), (hipStream_t) stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_memset_async(void *dst, int value, size_t count, ug_hip_stream_t stream)
{
        UG_HIP_TRY(hipMemsetAsync(dst, value, count, (hipStream_t) stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_stream_create(ug_hip_stream_t *stream)
{
        if (!stream) return UG_HIP_EINVAL;
        hipStream_t s;
        UG_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = (ug_hip_stream_t) s;
        return UG_HIP_SUCCESS;
}

int ug_hip_stream_destroy(ug_hip_stream_t stream)
{
        UG_HIP_TRY(hipStreamDestroy((hipStream_t) stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_stream_sync(ug_hip_stream_t stream)
{
        UG_HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
        return UG_HIP_SUCCESS;
}

// ---- copy lanes: ONE upload stream and ONE download stream per device, shared by every caller of this process ----
// Several frames in flight, each on its own stream with its own H2D -> kernels -> D2H chain, make the copy engines serve two uploads
// (or two downloads) at once, each at half rate, and every frame's kernel starts late.  With all uploads of a device queued on one
// stream and all downloads on another -- events tie a frame's stages together -- each copy runs at the full rate of its direction and
// the kernels of frame k overlap the upload of k + 1 and the download of k - 1.  Measured PCIe-inclusive, 8K UYVY -> DXT5, two frames
// in flight: 651 -> 780 fps; 8K v210: 447 -> 593 (profiles/r03_e2e_sweep.txt).  UG_MI355X_COPY_LANES=0 puts the copies back on the
// caller's stream (A/B).
namespace {
constexpr int kMaxDevices = 64;
struct Lanes {
        hipStream_t up = nullptr, down = nullptr;
};
Lanes g_lanes[kMaxDevices];
std::mutex g_lanes_lock;
bool lanes_enabled()
{
        static const bool on = !(getenv("UG_MI355X_COPY_LANES") && getenv("UG_MI355X_COPY_LANES")[0] == '0');
        return on;
}
// the calling thread's current device must be `device`
hipError_t lanes_of(int device, Lanes &out)
{
        if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(g_lanes_lock);
        Lanes &l = g_lanes[device];
        if (l.up == nullptr) {
                hipError_t e = hipStreamCreateWithFlags(&l.up, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.down, hipStreamNonBlocking);
                if (e != hipSuccess) {
                        l.up = l.down = nullptr;
                        return e;
                }
        }
        out = l;
        return hipSuccess;
}
// `to` waits for everything queued on `from` so far
hipError_t chain(hipStream_t from, hipStream_t to)
{
        hipEvent_t ev;
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        e = hipEventRecord(ev, from);
        if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
        (void) hipEventDestroy(ev); // released by the runtime once it has completed
        return e;
}
} // namespace

int ug_hip_upload_ordered(int device, void *dst_dev, const void *src, size_t count, int kind, ug_hip_stream_t then_stream)
{
        if (!lanes_enabled()) return ug_hip_memcpy_async(dst_dev, src, count, kind, then_stream);
        Lanes l;
        UG_HIP_TRY(lanes_of(device, l));
        UG_HIP_TRY(chain((hipStream_t) then_stream, l.up)); // the destination may still be read by what the caller queued before (its previous frame)
        UG_HIP_TRY(hipMemcpyAsync(dst_dev, src, count, kind_of(kind), l.up));
        UG_HIP_TRY(chain(l.up, (hipStream_t) then_stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_download_ordered(int device, void *dst_host, const void *src_dev, size_t count, ug_hip_stream_t after_stream)
{
        if (!lanes_enabled()) return ug_hip_memcpy_async(dst_host, src_dev, count, UG_HIP_MEMCPY_DEVICE_TO_HOST, after_stream);
        Lanes l;
        UG_HIP_TRY(lanes_of(device, l));
        UG_HIP_TRY(chain((hipStream_t) after_stream, l.down));
        UG_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, count, hipMemcpyDeviceToHost, l.down));
        UG_HIP_TRY(chain(l.down, (hipStream_t) after_stream)); // ug_hip_stream_sync(after_stream) now also waits for the download
        return UG_HIP_SUCCESS;
}

int ug_hip_linesize(ug_pixfmt_t fmt, int width) { return ug::linesize(fmt, width); }

} // extern "C"
