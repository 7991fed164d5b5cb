/**
 * @file ug_dec_harness.c
 * Receiver-side drop-in proof: the reference's own src/video_decompress.c (module selection by priority through
 * lib_common.cpp) drives our decompress module:
 *     decompress_init_multi(DXT5, {}, RGBA, &s, 1); decompress_reconfigure(s, desc, 0, 8, 16, pitch, RGBA);
 *     decompress_frame(s, dst, src, len, 0, NULL, NULL); decompress_done(s);
 * usage: ug_dec_harness <DXT1|DXT1_YUV|DXT5|JPEG> <out codec> <w> <h> <in.bin> <out.raw> [pitch] [src_len]
 *        ug_dec_harness list
 *        ug_dec_harness devices        (prints the device list / state rotation / band count the modules derive from UG_PARAM; no GPU needed)
 * UG_DEC_REPEAT=<n>: the frame is decompressed n more times and the rate printed (THROUGHPUT ...), host frame in, host frame out.
 * UG_PARAM=<k>=<v>[,<k>=<v>...]: what `uv --param ...` would hold (host.cpp:1090-1121): get_commandline_param() answers from it, e.g.
 *                   UG_PARAM=mi355x-device=0:0,mi355x-bands=8.  A module that hands frames out late (jpeg_to_dxt_mi355x with several devices) is fed the
 *                   same frame until every frame that went in has come out; each one must equal the first.
 * UG_DEC_COPY_TWIN=1: after the THROUGHPUT loop, the same number of rounds of the frame's two copies alone (compressed bytes up, decoded picture down with
 *                   its pitch, pageable host memory, one stream, one after the other): "COPYONLY ..." -- what the link allows a synchronous decompress().
 * UG_DEC_TILES=<n>: the receiver's tile fan-out (rtp/video_decoders.cpp:590-612,676-690: one decompress state per tile, decompress_frame of all tiles
 *                   at the same time on worker threads): n states from ONE decompress_init_multi(..., n), n threads, UG_DEC_TILE_ROUNDS (default 20) frames each
 *                   into buffers of their own; every output must equal the single-state result ("TILES n=.. rounds=.. OK").
 */
#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lib_common.h"
#include "types.h"
#include "video_codec.h"
#include "video_decompress.h"

#include "mi355x_receiver.h"

/* what host.cpp would provide (the reference's tools/ug_stub.c answers NULL to every key; this one answers from UG_PARAM) */
static char *uv_argv_store[] = { "ug_dec_harness", NULL };
char **uv_argv = uv_argv_store;
void register_param(const char *param, const char *doc) { (void) param, (void) doc; }
bool tok_in_argv(char **argv, const char *tok) { (void) argv, (void) tok; return false; }
const char *get_commandline_param(const char *key)
{
        static char vals[8][128];
        static int slot;
        const char *p = getenv("UG_PARAM");
        const size_t kl = strlen(key);
        while (p != NULL && *p != '\0') {
                const char *end = strchr(p, ',');
                const size_t len = end ? (size_t) (end - p) : strlen(p);
                if (len >= kl && strncmp(p, key, kl) == 0 && (len == kl || p[kl] == '=')) {
                        char *v = vals[slot++ % 8];
                        snprintf(v, sizeof vals[0], "%.*s", len > kl ? (int) (len - kl - 1) : 0, p + kl + (len > kl ? 1 : 0));
                        return v;
                }
                p = end ? end + 1 : NULL;
        }
        return NULL;
}

struct tile_job {
        struct state_decompress *s;
        unsigned char *dst;
        const unsigned char *src, *want;
        unsigned src_len;
        size_t out_bytes;
        int rounds, bad;
};
static void *tile_worker(void *arg)
{
        struct tile_job *j = arg;
        for (int r = 0; r < j->rounds; r++) {
                memset(j->dst, 0, j->out_bytes);
                if (decompress_frame(j->s, j->dst, (unsigned char *) j->src, j->src_len, r, NULL, NULL) != DECODER_GOT_FRAME || memcmp(j->dst, j->want, j->out_bytes) != 0) j->bad++;
        }
        return NULL;
}

int main(int argc, char **argv)
{
        if (argc == 2 && strcmp(argv[1], "list") == 0) {
                list_modules(LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION, true);
                return 0;
        }
        if (argc == 2 && strcmp(argv[1], "devices") == 0) { // the device list the decompress modules would use (UG_PARAM / -D), and the states' turn; no GPU involved
                int devs[MI355X_MAX_DEVICES];
                bool bad = false;
                const int n = mi355x_receiver_devices(devs, MI355X_MAX_DEVICES, &bad);
                printf("DEVICES%s n=%d:", bad ? " BAD" : "", n);
                for (int i = 0; i < n; i++) printf(" %d", devs[i]);
                unsigned counter = 0;
                printf(" | states:");
                for (int i = 0; i < 5 && !bad; i++) printf(" %d", mi355x_next_state_device(&counter, "[harness] "));
                printf(" | bands=%d\n", mi355x_receiver_bands(MI355X_AUTO_BANDS)); // (0 = chosen by the module from the frame size)
                return bad ? 4 : 0;
        }
        if (argc < 7) {
                fprintf(stderr, "usage: %s <DXT1|DXT1_YUV|DXT5> <out codec> <w> <h> <in.bin> <out.raw> [pitch] [src_len: hand over only that many bytes (a short frame)]\n", argv[0]);
                return 1;
        }
        const codec_t in = get_codec_from_name(argv[1]), out = get_codec_from_name(argv[2]);
        const unsigned w = atoi(argv[3]), h = atoi(argv[4]);
        const int linesize = vc_get_linesize(w, out);
        const int pitch = argc > 7 ? atoi(argv[7]) : linesize;
        struct video_desc desc = { .width = w, .height = h, .color_spec = in, .fps = 30, .interlacing = PROGRESSIVE, .tile_count = 1 };
        size_t in_len = (size_t) ((w + 3) / 4 * 4) * ((h + 3) / 4 * 4) / (in == DXT1 || in == DXT1_YUV ? 2 : 1); // dxt_get_size (dxt_util.h:59-67)
        FILE *f = fopen(argv[5], "rb");
        if (!f) { fprintf(stderr, "cannot read input\n"); return 1; }
        if (in == JPEG) { // a compressed frame is as long as it is
                fseek(f, 0, SEEK_END);
                in_len = (size_t) ftell(f);
                fseek(f, 0, SEEK_SET);
        }
        const bool out_dxt = out == DXT1 || out == DXT5;
        const size_t out_bytes = out == I420 ? (size_t) w * h + 2 * (size_t) ((w + 1) / 2) * ((h + 1) / 2)
                                 : (out_dxt ? (size_t) ((w + 3) / 4 * 4) * ((h + 3) / 4 * 4) / (out == DXT1 ? 2 : 1) : (size_t) pitch * h);
        unsigned char *src = malloc(in_len), *dst = calloc(out_bytes + 64, 1);
        if (fread(src, 1, in_len, f) != in_len) { fprintf(stderr, "cannot read input\n"); return 1; }
        fclose(f);

        struct state_decompress *s = NULL;
        struct pixfmt_desc internal = { 0 };
        if (in == JPEG) { // what the receiver does first (video_decompress.c / rtp/video_decoders.cpp): a probe decoder tells the stream's internal format
                struct state_decompress *probe = NULL;
                if (decompress_init_multi(in, internal, VIDEO_CODEC_NONE, &probe, 1) && decompress_reconfigure(probe, desc, 0, 8, 16, 0, VIDEO_CODEC_NONE)) {
                        const decompress_status ps = decompress_frame(probe, NULL, src, (unsigned) in_len, 0, NULL, &internal);
                        printf("PROBE status=%d depth=%d subsampling=%d rgb=%d\n", (int) ps, internal.depth, internal.subsampling, (int) internal.rgb);
                        decompress_done(probe);
                }
        }
        if (!decompress_init_multi(in, internal, out, &s, 1)) {
                fprintf(stderr, "no decompressor for %s -> %s\n", argv[1], argv[2]);
                return 2;
        }
        if (!decompress_reconfigure(s, desc, 0, 8, 16, pitch, out)) {
                fprintf(stderr, "reconfigure failed\n");
                return 2;
        }
        const unsigned src_len = argc > 8 ? (unsigned) atoi(argv[8]) : (unsigned) in_len; // < in_len: a frame that lost its tail on the way
        decompress_status st = decompress_frame(s, dst, src, src_len, 0, NULL, NULL);
        int delay = 0; // frames a module keeps on their way (jpeg_to_dxt_mi355x: one per extra device)
        while (st == DECODER_NO_FRAME && delay < 64 && getenv("UG_PARAM")) {
                delay++;
                st = decompress_frame(s, dst, src, src_len, delay, NULL, NULL);
        }
        if (st != DECODER_GOT_FRAME) {
                fprintf(stderr, "decompress_frame status %d\n", (int) st);
                return 3;
        }
        if (delay) printf("DELAY frames=%d\n", delay);
        f = fopen(argv[6], "wb");
        fwrite(dst, 1, out_bytes, f);
        fclose(f);
        printf("OK %s -> %s %ux%u pitch=%d\n", argv[1], argv[2], w, h, pitch);
        const int tiles = getenv("UG_DEC_TILES") ? atoi(getenv("UG_DEC_TILES")) : 0;
        if (tiles > 1 && tiles <= 16) {
                struct state_decompress *ts[16] = { 0 };
                struct tile_job jobs[16];
                pthread_t th[16];
                if (!decompress_init_multi(in, internal, out, ts, tiles)) { fprintf(stderr, "tiles: no decompressor\n"); return 2; }
                int bad = 0;
                for (int t = 0; t < tiles; t++) {
                        if (!decompress_reconfigure(ts[t], desc, 0, 8, 16, pitch, out)) { fprintf(stderr, "tiles: reconfigure failed\n"); return 2; }
                        jobs[t] = (struct tile_job){ ts[t], calloc(out_bytes + 64, 1), src, dst, src_len, out_bytes, getenv("UG_DEC_TILE_ROUNDS") ? atoi(getenv("UG_DEC_TILE_ROUNDS")) : 20, 0 };
                }
                for (int t = 0; t < tiles; t++) pthread_create(&th[t], NULL, tile_worker, &jobs[t]);
                for (int t = 0; t < tiles; t++) { pthread_join(th[t], NULL); bad += jobs[t].bad; }
                for (int t = 0; t < tiles; t++) { decompress_done(ts[t]); free(jobs[t].dst); }
                printf("TILES n=%d rounds=%d %s bad=%d\n", tiles, jobs[0].rounds, bad ? "MISMATCH" : "OK", bad);
                if (bad) return 5;
        }
        const int repeat = getenv("UG_DEC_REPEAT") ? atoi(getenv("UG_DEC_REPEAT")) : 0;
        if (repeat > 0) {
                // UG_DEC_ROUNDS (default 1) rounds of `repeat` frames; with UG_DEC_COPY_TWIN each round is followed by the same number of rounds of the frame's
                // two copies alone (compressed bytes up, decoded picture down with its pitch; the same pageable buffers, one stream, one after the other): the
                // twin is measured BESIDE what it bounds, and the best round of each is what is compared
                const int rounds = getenv("UG_DEC_ROUNDS") ? atoi(getenv("UG_DEC_ROUNDS")) : 1;
                const bool twin = getenv("UG_DEC_COPY_TWIN") != NULL;
                struct timespec t0, t1;
                unsigned char *first = delay ? malloc(out_bytes) : NULL; // a delaying module: every frame that comes out later is the same picture
                if (first) memcpy(first, dst, out_bytes);
                void *dev_in = NULL, *dev_out = NULL;
                ug_hip_stream_t st2 = NULL;
                const size_t line = out_dxt || out == I420 ? out_bytes : (size_t) linesize, rows = out_dxt || out == I420 ? 1 : h;
                const size_t dp = out_dxt || out == I420 ? out_bytes : (size_t) pitch;
                unsigned char *twin_dst = twin ? calloc(out_bytes + 64, 1) : NULL;
                bool twin_ok = twin && ug_hip_set_device(0) == 0 && ug_hip_stream_create(&st2) == 0 && ug_hip_malloc(&dev_in, src_len + 64) == 0 && ug_hip_malloc(&dev_out, line * rows + 64) == 0;
                double best = 0, best_twin = 0;
                int seq = delay;
                for (int r = 0; r < (rounds < 1 ? 1 : rounds); r++) {
                        double sec = 0; // the decompress_frame calls alone: the comparison of a delaying module's frames (33 MB of memcmp at 8K) is the harness's, not the module's
                        for (int i = 0; i < repeat; i++) {
                                clock_gettime(CLOCK_MONOTONIC, &t0);
                                const decompress_status fs = decompress_frame(s, dst, src, src_len, ++seq, NULL, NULL);
                                clock_gettime(CLOCK_MONOTONIC, &t1);
                                sec += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
                                if (fs != DECODER_GOT_FRAME) return 3;
                                if (first && memcmp(first, dst, out_bytes) != 0) { fprintf(stderr, "frame %d differs from the first one\n", i); return 5; }
                        }
                        printf("THROUGHPUT frames=%d wall_s=%.4f fps=%.1f\n", repeat, sec, repeat / sec);
                        if (repeat / sec > best) best = repeat / sec;
                        if (twin_ok) {
                                clock_gettime(CLOCK_MONOTONIC, &t0);
                                for (int i = 0; i < repeat && twin_ok; i++) {
                                        twin_ok = ug_hip_memcpy_async(dev_in, src, src_len, UG_HIP_MEMCPY_HOST_TO_DEVICE, st2) == 0 &&
                                                  ug_hip_memcpy_2d_async(twin_dst, dp, dev_out, line, line, rows, UG_HIP_MEMCPY_DEVICE_TO_HOST, st2) == 0 && ug_hip_stream_sync(st2) == 0;
                                }
                                clock_gettime(CLOCK_MONOTONIC, &t1);
                                const double s2 = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
                                if (twin_ok && repeat / s2 > best_twin) best_twin = repeat / s2;
                        }
                }
                free(first);
                if (twin && twin_ok) {
                        printf("COPYONLY fps=%.1f up_bytes=%u down_bytes=%zu (best of %d rounds beside the decoder's) BEST fps=%.1f frac_of_copy_only=%.3f\n", best_twin, src_len, line * rows,
                               rounds < 1 ? 1 : rounds, best, best / best_twin);
                } else if (twin) {
                        printf("COPYONLY failed: %s\n", ug_hip_last_error_string());
                }
                free(twin_dst);
                if (dev_in) ug_hip_free(dev_in);
                if (dev_out) ug_hip_free(dev_out);
                if (st2) ug_hip_stream_destroy(st2);
        }
        decompress_done(s);
        return 0;
}
