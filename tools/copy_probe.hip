// copy_probe.hip -- what a receiver's download costs on this box (VERDICT r5 "What's weak" #3): device -> host copies of one decoded frame
// into PAGEABLE memory (what decompress() is handed), into pinned memory, and into pageable memory registered on first sight
// (hipHostRegister); contiguous, as one 2-D copy with a destination pitch, and as one copy per line; plus the upload of a compressed frame.
//   hipcc --offload-arch=gfx950 -O2 tools/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double time_ms(F f, int iters)
{
        f();
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < iters; i++) f();
        CK(hipDeviceSynchronize());
        return (now() - t0) * 1e3 / iters;
}

int main()
{
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct { const char *name; int w, h, bpp; } cases[] = { { "4K RGBA", 3840, 2160, 4 }, { "4K UYVY", 3840, 2160, 2 }, { "8K RGBA", 7680, 4320, 4 }, { "8K UYVY", 7680, 4320, 2 } };
        for (auto &c : cases) {
                const size_t ls = (size_t) c.w * c.bpp, pitch = ls + 64, n = ls * c.h, np = pitch * c.h;
                void *dev;
                CK(hipMalloc(&dev, np));
                CK(hipMemset(dev, 1, np));
                char *pageable = (char *) aligned_alloc(4096, np), *pinned, *registered = (char *) aligned_alloc(4096, np);
                memset(pageable, 0, np);
                memset(registered, 0, np);
                CK(hipHostMalloc((void **) &pinned, np, hipHostMallocDefault));
                const double t_reg0 = now();
                CK(hipHostRegister(registered, np, hipHostRegisterDefault));
                const double t_reg = (now() - t_reg0) * 1e3;
                CK(hipHostUnregister(registered)); // (measure it, then hand it on as plain pageable memory: the duplex probe below wants two pageable buffers)
                const int it = c.w > 4000 ? 10 : 30;
                auto gbs = [&](double ms, size_t bytes) { return bytes / ms * 1e-6; };
                printf("== %s: %zu MB per frame (hipHostRegister of the destination: %.2f ms, once)\n", c.name, n >> 20, t_reg);
                char *reg2 = (char *) aligned_alloc(4096, np);
                memset(reg2, 0, np);
                CK(hipHostRegister(reg2, np, hipHostRegisterDefault));
                struct { const char *n; char *p; } dsts[] = { { "pageable", pageable }, { "pinned", pinned }, { "registered", reg2 } };
                for (auto &d : dsts) {
                        const double a = time_ms([&] { CK(hipMemcpyAsync(d.p, dev, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }, it);
                        const double b = time_ms([&] { CK(hipMemcpy2DAsync(d.p, pitch, dev, ls, ls, c.h, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }, it);
                        const double e = time_ms([&] { CK(hipMemcpy2DAsync(d.p, pitch, dev, pitch, ls, c.h, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }, it);
                        const double l = time_ms([&] { for (int y = 0; y < c.h; y++) CK(hipMemcpyAsync(d.p + y * pitch, (char *) dev + y * ls, ls, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }, 3);
                        printf("  D2H -> %-10s contiguous %7.3f ms %6.1f GB/s | 2-D (dst pitch+64) %7.3f ms %6.1f GB/s | 2-D (both pitched) %7.3f ms | per line %8.3f ms %6.1f GB/s\n", d.n, a, gbs(a, n), b, gbs(b, n), e, l, gbs(l, n));
                }
                // the upload of what arrives: a DXT5 frame = w * h bytes, a JPEG about a tenth of that
                for (auto &d : dsts) {
                        const size_t m = (size_t) c.w * c.h;
                        const double a = time_ms([&] { CK(hipMemcpyAsync(dev, d.p, m, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }, it);
                        const double b = time_ms([&] { CK(hipMemcpyAsync(dev, d.p, m / 10, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }, it);
                        printf("  H2D <- %-10s %zu MB %7.3f ms %6.1f GB/s | %zu MB %7.3f ms %6.1f GB/s\n", d.n, m >> 20, a, gbs(a, m), (m / 10) >> 20, b, gbs(b, m / 10));
                }
                // full duplex from PAGEABLE memory: the copy calls block their host thread, so one thread alone can never have an upload and a
                // download in flight together; two threads (one per direction, a stream each) can -- is the link then used both ways at once?
                {
                        const size_t up = (size_t) c.w * c.h; // a DXT5 frame
                        void *dev_in;
                        CK(hipMalloc(&dev_in, up));
                        hipStream_t st2;
                        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
                        const double serial = time_ms([&] { CK(hipMemcpyAsync(dev_in, pageable, up, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
                                                            CK(hipMemcpyAsync(registered, dev, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }, it);
                        const double duplex = time_ms([&] {
                                std::thread t([&] { CK(hipMemcpyAsync(registered, dev, n, hipMemcpyDeviceToHost, st2)); CK(hipStreamSynchronize(st2)); });
                                CK(hipMemcpyAsync(dev_in, pageable, up, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
                                t.join(); }, it);
                        // the same in 4 bands each way (what a band pipeline would issue): thread A uploads band after band, thread B downloads band after band
                        const double duplex4 = time_ms([&] {
                                std::thread t([&] { for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(registered + k * (n / 4), (char *) dev + k * (n / 4), n / 4, hipMemcpyDeviceToHost, st2)); CK(hipStreamSynchronize(st2)); });
                                for (int k = 0; k < 4; k++) CK(hipMemcpyAsync((char *) dev_in + k * (up / 4), pageable + k * (up / 4), up / 4, hipMemcpyHostToDevice, st));
                                CK(hipStreamSynchronize(st));
                                t.join(); }, it);
                        {       // does the "async" copy call return before the copy is done?  (pageable: it does not)
                                CK(hipDeviceSynchronize());
                                const double c0 = now();
                                CK(hipMemcpyAsync(registered, dev, n, hipMemcpyDeviceToHost, st));
                                const double c1 = now();
                                CK(hipStreamSynchronize(st));
                                const double c2 = now();
                                CK(hipMemcpyAsync(pinned, dev, n, hipMemcpyDeviceToHost, st));
                                const double c3 = now();
                                CK(hipStreamSynchronize(st));
                                printf("  hipMemcpyAsync D2H returns after %.3f ms of %.3f (pageable) | %.3f ms of %.3f (pinned)\n", (c1 - c0) * 1e3, (c2 - c0) * 1e3, (c3 - c2) * 1e3, (now() - c2) * 1e3);
                        }
                        printf("  pageable both ways, %zu MB up + %zu MB down: one thread %7.3f ms | two threads %7.3f ms | two threads, 4 bands each %7.3f ms (the longer copy alone: see above)\n", up >> 20, n >> 20, serial, duplex, duplex4);
                        CK(hipStreamDestroy(st2));
                        CK(hipFree(dev_in));
                }
                CK(hipHostUnregister(reg2));
                free(reg2);
                CK(hipHostFree(pinned));
                free(pageable);
                free(registered);
                CK(hipFree(dev));
        }
        return 0;
}
