/**
 * @file ug_plugin_harness.cpp
 * The PLUGIN route of the drop-in boundary: what `uv` does in a modular build (--enable-plugins, BUILD_LIBRARIES).
 *
 * This binary links UltraGrid's compress + decompress frameworks and the registry (src/video_compress.cpp, src/video_decompress.c,
 * src/lib_common.cpp compiled with -DBUILD_LIBRARIES, and their support objects) -- and NO module object.  It is linked -rdynamic, as `uv`
 * is, so that a module's undefined symbols (register_library, vc_get_linesize, log_msg, ...) bind to the executable.  The modules arrive the
 * way host.cpp:613 brings them in:
 *
 *     open_all("ultragrid_*.so", libs)        lib_common.cpp:186-223: glob <dirname(argv[0])>/../lib/ultragrid/ultragrid_*.so,
 *                                             dlopen(RTLD_NOW | RTLD_GLOBAL) each; the static constructor REGISTER_MODULE leaves in the
 *                                             .so (lib_common.h:124-131) calls register_library() during that dlopen
 *
 * and are then found by name: compress_init("dxt:DXT5") / decompress_init_multi(DXT5 -> RGBA).  Unresolved symbols, a constructor that
 * runs before the registry exists, or a file name the loader's pattern (ultragrid_<class prefix>_<name>.so, :296-304) does not match would
 * show here and nowhere else.  The test lays the files out as an installation does: <tmp>/bin/ug_plugin_harness, <tmp>/lib/ultragrid/*.so.
 *
 * usage: ug_plugin_harness list
 *        ug_plugin_harness compress <cfg> <codec> <w> <h> <in.raw> <out.bin>
 *        ug_plugin_harness decompress <in codec> <out codec> <w> <h> <in.bin> <out.raw>
 * Exit code 0 = OK, 2 = no module took the job, 3 = frame dropped, 4 = no plugin was opened.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <vector>

#include "debug.h"
#include "host.h"
#include "lib_common.h"
#include "types.h"
#include "video_codec.h"
#include "video_compress.h"
#include "video_decompress.h"
#include "video_frame.h"

// what src/host.cpp supplies in `uv` and this harness does not link
char **uv_argv;
extern "C" {
const char *get_commandline_param(const char *key) { (void) key; return nullptr; }
void register_param(const char *param, const char *doc) { (void) param, (void) doc; }
bool tok_in_argv(char **argv, const char *tok) { (void) argv, (void) tok; return false; }
}

static std::vector<unsigned char> read_file(const char *path)
{
        std::vector<unsigned char> v;
        FILE *f = fopen(path, "rb");
        if (!f) { perror(path); exit(1); }
        fseek(f, 0, SEEK_END);
        v.resize((size_t) ftell(f));
        fseek(f, 0, SEEK_SET);
        if (fread(v.data(), 1, v.size(), f) != v.size()) { perror(path); exit(1); }
        fclose(f);
        return v;
}

static void write_file(const char *path, const void *p, size_t n)
{
        FILE *f = fopen(path, "wb");
        if (!f || fwrite(p, 1, n, f) != n) { perror(path); exit(1); }
        fclose(f);
}

int main(int argc, char **argv)
{
        uv_argv = argv; // host.cpp:521
        std::list<void *> libs;
        open_all("ultragrid_*.so", libs); // host.cpp:613
        printf("PLUGINS opened=%zu\n", libs.size());
        if (libs.empty()) {
                fprintf(stderr, "no plugin under <dir of %s>/../lib/ultragrid\n", argv[0]);
                return 4;
        }
        if (argc == 2 && strcmp(argv[1], "list") == 0) {
                list_modules(LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION, true);
                list_modules(LIBRARY_CLASS_VIDEO_DECOMPRESS, VIDEO_DECOMPRESS_ABI_VERSION, true);
                return 0;
        }
        if (argc < 8) {
                fprintf(stderr, "usage: %s list | compress <cfg> <codec> <w> <h> <in> <out> | decompress <in codec> <out codec> <w> <h> <in> <out>\n", argv[0]);
                return 1;
        }
        const unsigned w = atoi(argv[4]), h = atoi(argv[5]);
        if (strcmp(argv[1], "compress") == 0) {
                struct video_desc desc{};
                desc.width = w; desc.height = h; desc.color_spec = get_codec_from_name(argv[3]); desc.fps = 30; desc.interlacing = PROGRESSIVE;
                desc.tile_count = 1;
                const std::vector<unsigned char> in = read_file(argv[6]);
                std::shared_ptr<video_frame> frame(vf_alloc_desc_data(desc), vf_free);
                if (in.size() < frame->tiles[0].data_len) { fprintf(stderr, "short input\n"); return 1; }
                memcpy(frame->tiles[0].data, in.data(), frame->tiles[0].data_len);
                struct compress_state *c = nullptr;
                if (compress_init(nullptr, argv[2], &c) != 0) {
                        fprintf(stderr, "compress_init(\"%s\") failed\n", argv[2]);
                        return 2;
                }
                compress_frame(c, frame);
                compress_frame(c, {}); // poison pill, as rxtx does on exit
                std::shared_ptr<video_frame> out = compress_pop(c);
                if (!out) { fprintf(stderr, "frame dropped\n"); compress_done(c); return 3; }
                write_file(argv[7], out->tiles[0].data, out->tiles[0].data_len);
                printf("OK compress codec=%s tile0=%ux%u len=%u\n", get_codec_name(out->color_spec), out->tiles[0].width, out->tiles[0].height, out->tiles[0].data_len);
                out.reset();
                while (compress_pop(c)) {}
                compress_done(c);
                return 0;
        }
        if (strcmp(argv[1], "decompress") == 0) {
                const codec_t in_c = get_codec_from_name(argv[2]), out_c = get_codec_from_name(argv[3]);
                struct video_desc desc{};
                desc.width = w; desc.height = h; desc.color_spec = in_c; desc.fps = 30; desc.interlacing = PROGRESSIVE; desc.tile_count = 1;
                const std::vector<unsigned char> in = read_file(argv[6]);
                const int pitch = vc_get_linesize(w, out_c);
                std::vector<unsigned char> dst((size_t) pitch * h + 64);
                struct pixfmt_desc internal{};
                if (in_c == JPEG) { // the receiver's probe decoder first (video_decompress.c; rtp/video_decoders.cpp)
                        struct state_decompress *probe = nullptr;
                        if (decompress_init_multi(in_c, internal, VIDEO_CODEC_NONE, &probe, 1) && decompress_reconfigure(probe, desc, 0, 8, 16, 0, VIDEO_CODEC_NONE)) {
                                decompress_frame(probe, nullptr, const_cast<unsigned char *>(in.data()), (unsigned) in.size(), 0, nullptr, &internal);
                                decompress_done(probe);
                        }
                }
                struct state_decompress *s = nullptr;
                if (!decompress_init_multi(in_c, internal, out_c, &s, 1) || !decompress_reconfigure(s, desc, 0, 8, 16, pitch, out_c)) {
                        fprintf(stderr, "no decompressor for %s -> %s\n", argv[2], argv[3]);
                        return 2;
                }
                if (decompress_frame(s, dst.data(), const_cast<unsigned char *>(in.data()), (unsigned) in.size(), 0, nullptr, nullptr) != DECODER_GOT_FRAME) {
                        fprintf(stderr, "no frame\n");
                        return 3;
                }
                write_file(argv[7], dst.data(), (size_t) pitch * h);
                printf("OK decompress %s -> %s %ux%u\n", argv[2], argv[3], w, h);
                decompress_done(s);
                return 0;
        }
        return 1;
}
