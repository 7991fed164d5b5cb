/**
 * @file ug_dec_harness.c
 * Receiver-side drop-in proof: the reference's own src/video_decompress.c (module selection by priority through
 * lib_common.cpp) drives our decompress module:
 *     decompress_init_multi(DXT5, {}, RGBA, &s, 1); decompress_reconfigure(s, desc, 0, 8, 16, pitch, RGBA);
 *     decompress_frame(s, dst, src, len, 0, NULL, NULL); decompress_done(s);
 * usage: ug_dec_harness <DXT1|DXT1_YUV|DXT5|JPEG> <out codec> <w> <h> <in.bin> <out.raw> [pitch] [src_len]
 *        ug_dec_harness list
 * UG_DEC_REPEAT=<n>: the frame is decompressed n more times and the rate printed (THROUGHPUT ...), host frame in, host frame out.
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
        if (argc < 7) {
                fprintf(stderr, "usage: %s <DXT1|DXT1_YUV|DXT5> <out codec> <w> <h> <in.bin> <out.raw> [pitch] [src_len: hand over only that many bytes (a short frame)]\n", argv[0]);
                return 1;
        }
        const codec_t in = get_codec_from_name(argv[1]), out = get_codec_from_name(argv[2]);
        const unsigned w = atoi(argv[3]), h = atoi(argv[4]);
        const int linesize = vc_get_linesize(w, out);
        const int pitch = argc > 7 ? atoi(argv[7]) : linesize;
        struct video_desc desc = { .width = w, .height = h, .color_spec = in, .fps = 30, .interlacing = PROGRESSIVE, .tile_count = 1 };
        size_t in_len = (size_t) w * h / (in == DXT1 || in == DXT1_YUV ? 2 : 1);
        FILE *f = fopen(argv[5], "rb");
        if (!f) { fprintf(stderr, "cannot read input\n"); return 1; }
        if (in == JPEG) { // a compressed frame is as long as it is
                fseek(f, 0, SEEK_END);
                in_len = (size_t) ftell(f);
                fseek(f, 0, SEEK_SET);
        }
        const size_t out_bytes = out == I420 ? (size_t) w * h + 2 * (size_t) ((w + 1) / 2) * ((h + 1) / 2) : (size_t) pitch * h;
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
        const decompress_status st = decompress_frame(s, dst, src, src_len, 0, NULL, NULL);
        if (st != DECODER_GOT_FRAME) {
                fprintf(stderr, "decompress_frame status %d\n", (int) st);
                return 3;
        }
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
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                for (int i = 0; i < repeat; i++) {
                        if (decompress_frame(s, dst, src, src_len, i + 1, NULL, NULL) != DECODER_GOT_FRAME) return 3;
                }
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
                printf("THROUGHPUT frames=%d wall_s=%.4f fps=%.1f\n", repeat, sec, repeat / sec);
        }
        decompress_done(s);
        return 0;
}
