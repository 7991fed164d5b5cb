// ug_runtime.hip -- runtime shim of libug_mi355x.so: device select, device / pinned-host
// allocation, copies, streams, last-error text.  Own entry points for what UltraGrid's
// modules get from src/cuda_wrapper.h:50-76 (that shim is CUDA-only and is not hipified).
#include <string.h>

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

int ug_hip_linesize(ug_pixfmt_t fmt, int width) { return ug::linesize(fmt, width); }

} // extern "C"
