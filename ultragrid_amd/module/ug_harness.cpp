/**
 * @file ug_harness.cpp
 * Drop-in proof: links UltraGrid's OWN compress framework + plugin registry
 * (src/video_compress.cpp, src/lib_common.cpp and their support objects, compiled from
 * /root/reference by ultragrid_amd/module/Makefile -- recipe of SURVEY.md 8(b) [probe]) with our
 * module object, then drives the public API exactly as rxtx.cpp does:
 *
 *     compress_init(nullptr, "dxt:DXT5", &c); compress_frame(c, frame); compress_pop(c); compress_done(c);
 *
 * usage: ug_harness <cfg> <codec> <width> <height> <in.raw> <out.bin> [tiles] [dev|host] [frames]
 *        frames: number of frames in in.raw (default 1); all are pushed, the compressed frames are written in pop order
 *        [repeat]: push the frame set that many times (throughput measurement; only the last pass is kept)
 *        UG_HARNESS_PACE_US=<n>: sleep that long between pushes (a source at display rate: one frame in flight); the run then also prints
 *        "LATENCY frames=.. median_ms=.. p95_ms=.. min_ms=.." over compress_end - compress_start of every frame that came back
 *        dev: hand the frame over device-resident (tile data = device pointers, mem_location = CUDA_MEM)
 *        ug_harness list
 * The compressed tile(s) are written to <out.bin> (tile after tile); a test compares them with the
 * CPU oracle.  Exit code 0 = OK, 2 = module refused (no GPU / bad cfg), 3 = frame dropped.
 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "debug.h"
#include "host.h"
#include "lib_common.h"
#include "types.h"
#include "tv.h"
#include "video_codec.h"
#include "../../include/ug_mi355x.h"
#include "video_compress.h"
#include "video_frame.h"

int main(int argc, char **argv)
{
        if (argc == 2 && strcmp(argv[1], "list") == 0) {
                list_modules(LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION, true);
                return 0;
        }
        if (argc < 7) {
                fprintf(stderr, "usage: %s <cfg> <codec> <w> <h> <in.raw> <out.bin> [tiles]\n       %s list\n", argv[0], argv[0]);
                return 1;
        }
        const char *cfg = argv[1];
        const codec_t codec = get_codec_from_name(argv[2]);
        const unsigned w = atoi(argv[3]), h = atoi(argv[4]);
        const unsigned tiles = argc > 7 ? atoi(argv[7]) : 1;
        if (codec == VIDEO_CODEC_NONE) {
                fprintf(stderr, "unknown codec %s\n", argv[2]);
                return 1;
        }
        struct video_desc desc{};
        desc.width = w; desc.height = h; desc.color_spec = codec; desc.fps = 30; desc.interlacing = PROGRESSIVE;
        desc.tile_count = tiles;
        if (const char *il = getenv("UG_HARNESS_INTERLACING")) { // "merged": an INTERLACED_MERGED source (both fields in one frame, as capture cards deliver 1080i)
                if (strcmp(il, "merged") == 0) desc.interlacing = INTERLACED_MERGED;
        }
        const unsigned nframes = argc > 9 ? atoi(argv[9]) : 1;
        const unsigned repeat = argc > 10 ? atoi(argv[10]) : 1;
        const bool devmem = argc > 8 && strcmp(argv[8], "dev") == 0;
        FILE *in = fopen(argv[5], "rb");
        if (!in) { perror("in"); return 1; }
        std::vector<void *> dev_bufs, pinned_bufs;
        std::vector<std::shared_ptr<video_frame>> inputs;
        for (unsigned n = 0; n < nframes; n++) {
                struct video_frame *f = vf_alloc_desc_data(desc);
                for (unsigned t = 0; t < tiles; t++) {
                        if (fread(f->tiles[t].data, 1, f->tiles[t].data_len, in) != f->tiles[t].data_len) {
                                fprintf(stderr, "short read (%u bytes per tile expected)\n", f->tiles[t].data_len);
                                return 1;
                        }
                }
                if (devmem) { // device-resident video_frame (types.h:295-298): the module must not upload it again
                        struct video_frame *fd = vf_alloc_desc(desc);
                        for (unsigned t = 0; t < tiles; t++) {
                                void *p = nullptr;
                                if (ug_hip_malloc(&p, f->tiles[t].data_len + 64) != UG_HIP_SUCCESS ||
                                    ug_hip_memcpy(p, f->tiles[t].data, f->tiles[t].data_len, UG_HIP_MEMCPY_HOST_TO_DEVICE) != UG_HIP_SUCCESS) {
                                        fprintf(stderr, "device staging failed: %s\n", ug_hip_last_error_string());
                                        return 1;
                                }
                                dev_bufs.push_back(p);
                                fd->tiles[t].data = (char *) p;
                                fd->tiles[t].data_len = f->tiles[t].data_len;
                        }
                        fd->mem_location = CUDA_MEM;
                        vf_free(f); // the host copy is gone: only the device copy can be what gets encoded
                        f = fd;
                }
                if (!devmem && getenv("UG_HARNESS_PINNED")) { // the source's frames in pinned host memory (a capture module that allocates through the module's allocator hook)
                        struct video_frame *fp = vf_alloc_desc(desc);
                        for (unsigned t = 0; t < tiles; t++) {
                                void *p = nullptr;
                                if (ug_hip_malloc_host(&p, f->tiles[t].data_len + 64) != UG_HIP_SUCCESS) { fprintf(stderr, "pinned allocation failed\n"); return 1; }
                                memcpy(p, f->tiles[t].data, f->tiles[t].data_len);
                                pinned_bufs.push_back(p);
                                fp->tiles[t].data = (char *) p;
                                fp->tiles[t].data_len = f->tiles[t].data_len;
                        }
                        vf_free(f);
                        f = fp;
                }
                inputs.emplace_back(f, vf_free);
        }
        fclose(in);

        struct compress_state *c = nullptr;
        int rc = compress_init(nullptr, cfg, &c);
        if (rc != 0) {
                fprintf(stderr, "compress_init(\"%s\") rc=%d\n", cfg, rc);
                return 2;
        }
        // sender-thread stand-in (rxtx.cpp:260-288): pops until the poison pill arrives
        std::vector<std::shared_ptr<video_frame>> popped;
        std::vector<double> latency_ms;
        const unsigned pace_us = getenv("UG_HARNESS_PACE_US") ? (unsigned) atoi(getenv("UG_HARNESS_PACE_US")) : 0;
        std::thread sender([&] {
                while (std::shared_ptr<video_frame> f2 = compress_pop(c)) {
                        latency_ms.push_back((double) (f2->compress_end - f2->compress_start) / 1e6);
                        popped.push_back(f2);
                        if (repeat > 1 && popped.size() > nframes) { // like the real sender: frames go back to the module's pool
                                popped.erase(popped.begin());
                        }
                }
        });
        const time_ns_t t_start = get_time_in_ns();
        for (unsigned r = 0; r < repeat; r++) {
                for (auto &frame : inputs) {
                        compress_frame(c, frame);
                        if (pace_us) std::this_thread::sleep_for(std::chrono::microseconds(pace_us));
                }
        }
        inputs.clear();
        compress_frame(c, {}); // poison pill, as rxtx does on exit
        sender.join();
        const double wall_s = (double) (get_time_in_ns() - t_start) / 1e9;
        printf("THROUGHPUT frames=%u wall_s=%.4f fps=%.1f\n", nframes * repeat, wall_s, nframes * repeat / wall_s);
        if (pace_us && latency_ms.size() > 4) {
                std::vector<double> v(latency_ms.begin() + 2, latency_ms.end()); // (the first frames configure the state and page the buffers in)
                std::sort(v.begin(), v.end());
                printf("LATENCY frames=%zu median_ms=%.3f p95_ms=%.3f min_ms=%.3f\n", v.size(), v[v.size() / 2], v[v.size() * 95 / 100], v[0]);
        }
        if (popped.empty()) { // only the pill came back: the module dropped the frame (video_compress.cpp:394-398)
                fprintf(stderr, "frame dropped\n");
                compress_done(c);
                return 3;
        }
        FILE *o = fopen(argv[6], "wb");
        if (!o) { perror("out"); return 1; }
        for (auto &out : popped) {
                for (unsigned t = 0; t < out->tile_count; t++) {
                        fwrite(out->tiles[t].data, 1, out->tiles[t].data_len, o);
                }
        }
        fclose(o);
        {
                std::shared_ptr<video_frame> out = popped[0];
                printf("OK codec=%s interlacing=%s frames=%zu tiles=%u tile0=%ux%u len=%u compress_ms=%.3f seq=", get_codec_name(out->color_spec),
                       get_interlacing_suffix(out->interlacing), popped.size(),
                       out->tile_count, out->tiles[0].width, out->tiles[0].height, out->tiles[0].data_len,
                       (double) (out->compress_end - out->compress_start) / 1e6);
                for (auto &f2 : popped) printf("%u,", f2->seq);
                printf("\n");
        }
        popped.clear();
        compress_done(c);
        for (void *p : dev_bufs) ug_hip_free(p);
        for (void *p : pinned_bufs) ug_hip_free_host(p);
        return 0;
}
