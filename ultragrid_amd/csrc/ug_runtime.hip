// ug_runtime.hip -- runtime shim of libug_mi355x.so: device select, device / pinned-host
// allocation, copies, streams, last-error text.  Own entry points for what UltraGrid's
// modules get from src/cuda_wrapper.h:50-76 (that shim is CUDA-only and is not hipified).
#include <sched.h>
#include <stdio.h>
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

} // extern "C"
static hipMemcpyKind kind_of(int kind)
{
        switch (kind) {
        case UG_HIP_MEMCPY_HOST_TO_DEVICE: return hipMemcpyHostToDevice;
        case UG_HIP_MEMCPY_DEVICE_TO_HOST: return hipMemcpyDeviceToHost;
        case UG_HIP_MEMCPY_DEVICE_TO_DEVICE: return hipMemcpyDeviceToDevice;
        }
        return hipMemcpyDefault;
}
extern "C" {

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

// `rows` lines of `width_bytes` each, `spitch` / `dpitch` bytes apart: what a display pitch that differs from the packed line size needs
// (video_decompress/dxt_glsl.c:163-186 and gpujpeg.c:305-315 do a CPU line loop there; one copy per LINE through the runtime measures
// 1-3 GB/s, one 2-D copy the full rate of the link: profiles/r06_copy_probe.txt)
static bool copy_2d_ok(const void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, const char *who)
{
        // (hipMemcpy2D's own limits are the pitches of its 2-D descriptor; the sizes here are those of one picture: ug::dims_ok's range)
        if (!dst || !src || width_bytes == 0 || rows == 0 || width_bytes > dpitch || width_bytes > spitch || rows > 65536 ||
            dpitch > (size_t) 0x7fffffff || spitch > (size_t) 0x7fffffff) {
                ug::set_last_error_msg(who);
                return false;
        }
        return true;
}
int ug_hip_memcpy_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, int kind, ug_hip_stream_t stream)
{
        if (!copy_2d_ok(dst, dpitch, src, spitch, width_bytes, rows, "ug_hip_memcpy_2d_async: bad geometry (0 < width_bytes <= both pitches, 0 < rows <= 65536)")) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, kind_of(kind), (hipStream_t) stream));
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

// Events: what lets one host thread queue work on a stream while ANOTHER host thread waits for a point of it on a stream of its own -- the
// receivers' band pipeline (vdecompress_dxt_mi355x.c): copies from / to pageable memory block the thread that issues them
// (profiles/r06_copy_probe.txt), so an upload and a download are only ever in flight together when two threads issue them.
int ug_hip_event_create(ug_hip_event_t *event)
{
        if (!event) return UG_HIP_EINVAL;
        hipEvent_t e = nullptr;
        UG_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *event = (ug_hip_event_t) e;
        return UG_HIP_SUCCESS;
}

int ug_hip_event_destroy(ug_hip_event_t event)
{
        if (!event) return UG_HIP_SUCCESS;
        UG_HIP_TRY(hipEventDestroy((hipEvent_t) event));
        return UG_HIP_SUCCESS;
}

int ug_hip_event_record(ug_hip_event_t event, ug_hip_stream_t stream)
{
        if (!event) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipEventRecord((hipEvent_t) event, (hipStream_t) stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_stream_wait_event(ug_hip_stream_t stream, ug_hip_event_t event)
{
        if (!event) return UG_HIP_EINVAL;
        UG_HIP_TRY(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
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
} // extern "C"  (the helpers below are C++ with internal linkage: inside the block their names would be exported as plain C symbols)
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
// `to` waits for everything queued on `from` so far.  The event is the calling thread's own (two per device, one per direction of the
// hand-over, created on first use and destroyed with the thread): a wait takes the event's state at the time of the call, so the next record
// on the same event cannot disturb it, and a copy costs two records + two waits instead of two event creations and destructions on top.
struct ThreadEvents {
        hipEvent_t ev[kMaxDevices][2] = {};
        ~ThreadEvents()
        {
                for (auto &d : ev) {
                        for (hipEvent_t e : d) {
                                if (e) (void) hipEventDestroy(e);
                        }
                }
        }
};
thread_local ThreadEvents t_events;
hipError_t chain(int device, int which, hipStream_t from, hipStream_t to)
{
        hipEvent_t &ev = t_events.ev[device][which];
        if (ev == nullptr) {
                const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e != hipSuccess) {
                        ev = nullptr;
                        return e;
                }
        }
        hipError_t e = hipEventRecord(ev, from);
        if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
        return e;
}
} // namespace
extern "C" {

int ug_hip_upload_ordered_ex(int device, void *dst_dev, const void *src, size_t count, int kind, ug_hip_stream_t then_stream, int flags)
{
        if (flags & ~UG_HIP_COPY_NO_WAIT) {
                ug::set_last_error_msg("ug_hip_upload_ordered_ex: unknown flag");
                return UG_HIP_EINVAL;
        }
        if (!lanes_enabled()) return ug_hip_memcpy_async(dst_dev, src, count, kind, then_stream);
        Lanes l;
        UG_HIP_TRY(lanes_of(device, l));
        if (!(flags & UG_HIP_COPY_NO_WAIT)) {
                UG_HIP_TRY(chain(device, 0, (hipStream_t) then_stream, l.up)); // the destination may still be read by what the caller queued before (its previous frame)
        }
        UG_HIP_TRY(hipMemcpyAsync(dst_dev, src, count, kind_of(kind), l.up));
        UG_HIP_TRY(chain(device, 1, l.up, (hipStream_t) then_stream));
        return UG_HIP_SUCCESS;
}

int ug_hip_upload_ordered(int device, void *dst_dev, const void *src, size_t count, int kind, ug_hip_stream_t then_stream)
{
        return ug_hip_upload_ordered_ex(device, dst_dev, src, count, kind, then_stream, 0);
}

int ug_hip_download_ordered_ex(int device, void *dst_host, const void *src_dev, size_t count, ug_hip_stream_t after_stream, int flags)
{
        if (flags & ~UG_HIP_COPY_NO_JOIN) {
                ug::set_last_error_msg("ug_hip_download_ordered_ex: unknown flag");
                return UG_HIP_EINVAL;
        }
        if (!lanes_enabled()) return ug_hip_memcpy_async(dst_host, src_dev, count, UG_HIP_MEMCPY_DEVICE_TO_HOST, after_stream);
        Lanes l;
        UG_HIP_TRY(lanes_of(device, l));
        UG_HIP_TRY(chain(device, 0, (hipStream_t) after_stream, l.down));
        UG_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, count, hipMemcpyDeviceToHost, l.down));
        if (!(flags & UG_HIP_COPY_NO_JOIN)) {
                UG_HIP_TRY(chain(device, 1, l.down, (hipStream_t) after_stream)); // ug_hip_stream_sync(after_stream) now also waits for the download (and every earlier one of the lane)
        }
        return UG_HIP_SUCCESS;
}

int ug_hip_download_2d_ordered_ex(int device, void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width_bytes, size_t rows,
                                  ug_hip_stream_t after_stream, int flags)
{
        if (flags & ~UG_HIP_COPY_NO_JOIN) {
                ug::set_last_error_msg("ug_hip_download_2d_ordered_ex: unknown flag");
                return UG_HIP_EINVAL;
        }
        if (!copy_2d_ok(dst_host, dpitch, src_dev, spitch, width_bytes, rows, "ug_hip_download_2d_ordered_ex: bad geometry (0 < width_bytes <= both pitches, 0 < rows <= 65536)")) return UG_HIP_EINVAL;
        if (!lanes_enabled()) return ug_hip_memcpy_2d_async(dst_host, dpitch, src_dev, spitch, width_bytes, rows, UG_HIP_MEMCPY_DEVICE_TO_HOST, after_stream);
        Lanes l;
        UG_HIP_TRY(lanes_of(device, l));
        UG_HIP_TRY(chain(device, 0, (hipStream_t) after_stream, l.down));
        UG_HIP_TRY(hipMemcpy2DAsync(dst_host, dpitch, src_dev, spitch, width_bytes, rows, hipMemcpyDeviceToHost, l.down));
        if (!(flags & UG_HIP_COPY_NO_JOIN)) {
                UG_HIP_TRY(chain(device, 1, l.down, (hipStream_t) after_stream));
        }
        return UG_HIP_SUCCESS;
}

int ug_hip_download_ordered(int device, void *dst_host, const void *src_dev, size_t count, ug_hip_stream_t after_stream)
{
        return ug_hip_download_ordered_ex(device, dst_host, src_dev, count, after_stream, 0);
}

int ug_hip_linesize(ug_pixfmt_t fmt, int width)
{
        const int ls = ug::linesize(fmt, width); // 0: unknown format, or a width outside 1..65536 (never a wrapped int)
        if (ls <= 0) {
                ug::set_last_error_msg("ug_hip_linesize: unknown pixel format, or width outside 1..65536");
                return UG_HIP_EINVAL;
        }
        return ls;
}

// ---- NUMA placement of the threads that feed a GPU ("expect host-side limits before 8x scaling") ----
// A worker of the frame sharder copies frames into pinned memory, queues the transfers and waits for them; on a two-socket box with
// eight GPUs it should do that on the socket the GPU's PCIe root complex hangs off, and its pinned pool should be first touched there.
// The node comes from sysfs (numa_node of the PCI function), the CPUs from the node's cpulist; `sysfs_root` (NULL = "/sys") lets the
// CPU tests point both at a fake tree.
} // extern "C"
namespace {
constexpr int kMaxCpus = 4096;
const char *sysroot(const char *r) { return r && *r ? r : "/sys"; }
} // namespace
extern "C" {

int ug_hip_numa_node_of_pci(const char *bdf, const char *sysfs_root, int *node)
{
        if (!bdf || !node || strchr(bdf, '/') || strlen(bdf) > 32) {
                ug::set_last_error_msg("ug_hip_numa_node_of_pci: bad arguments");
                return UG_HIP_EINVAL;
        }
        *node = -1;
        char path[512], lower[40];
        size_t i = 0;
        for (; bdf[i]; i++) lower[i] = (char) (bdf[i] >= 'A' && bdf[i] <= 'F' ? bdf[i] - 'A' + 'a' : bdf[i]); // sysfs names are lower case
        lower[i] = 0;
        snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", sysroot(sysfs_root), lower);
        FILE *f = fopen(path, "r");
        if (!f) return UG_HIP_SUCCESS; // not said: -1
        int v = -1;
        if (fscanf(f, "%d", &v) == 1 && v >= 0) *node = v; // single-node boxes say -1
        fclose(f);
        return UG_HIP_SUCCESS;
}

int ug_hip_device_numa_node(int device, int *node)
{
        if (!node) return UG_HIP_EINVAL;
        char bdf[64] = "";
        UG_HIP_TRY(hipDeviceGetPCIBusId(bdf, (int) sizeof bdf, device));
        return ug_hip_numa_node_of_pci(bdf, nullptr, node);
}

int ug_hip_bind_thread_to_numa_node(int node, const char *sysfs_root, int *cpus_bound)
{
        if (cpus_bound) *cpus_bound = 0;
        if (node < 0) return UG_HIP_SUCCESS; // unknown node: the thread is left where it is
        char path[512], list[8192];
        snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", sysroot(sysfs_root), node);
        FILE *f = fopen(path, "r");
        if (!f) return UG_HIP_SUCCESS;
        const size_t n = fread(list, 1, sizeof list - 1, f);
        fclose(f);
        list[n] = 0;
        cpu_set_t *want = CPU_ALLOC(kMaxCpus), *have = CPU_ALLOC(kMaxCpus);
        const size_t sz = CPU_ALLOC_SIZE(kMaxCpus);
        if (!want || !have) {
                if (want) CPU_FREE(want);
                if (have) CPU_FREE(have);
                ug::set_last_error_msg("ug_hip_bind_thread_to_numa_node: out of memory");
                return UG_HIP_ERUNTIME;
        }
        CPU_ZERO_S(sz, want);
        for (const char *p = list; *p;) { // "0-31,64-95"
                char *e;
                const long a = strtol(p, &e, 10);
                if (e == p) break;
                long b = a;
                if (*e == '-') b = strtol(e + 1, &e, 10);
                for (long c = a; c <= b && c < kMaxCpus; c++) {
                        if (c >= 0) CPU_SET_S((size_t) c, sz, want);
                }
                p = *e == ',' ? e + 1 : e;
                if (*e != ',') break;
        }
        int rc = UG_HIP_SUCCESS, count = 0;
        if (sched_getaffinity(0, sz, have) == 0) { // 0 = the calling thread
                CPU_AND_S(sz, want, want, have); // never widen what the process was given (cpusets, taskset)
                count = CPU_COUNT_S(sz, want);
                if (count > 0 && !CPU_EQUAL_S(sz, want, have)) {
                        if (sched_setaffinity(0, sz, want) != 0) {
                                ug::set_last_error_msg("ug_hip_bind_thread_to_numa_node: sched_setaffinity failed");
                                rc = UG_HIP_ERUNTIME;
                                count = 0;
                        }
                } // (already exactly there: nothing to change; the count still says where the thread runs)
        }
        CPU_FREE(want);
        CPU_FREE(have);
        if (cpus_bound) *cpus_bound = count;
        return rc;
}

int ug_hip_bind_thread_to_device(int device, int *cpus_bound)
{
        if (cpus_bound) *cpus_bound = 0;
        int node = -1;
        const int rc = ug_hip_device_numa_node(device, &node);
        if (rc != UG_HIP_SUCCESS) return rc;
        return ug_hip_bind_thread_to_numa_node(node, nullptr, cpus_bound);
}

} // extern "C"
