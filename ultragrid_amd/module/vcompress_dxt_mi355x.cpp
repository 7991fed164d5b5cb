/**
 * @file vcompress_dxt_mi355x.cpp
 * UltraGrid video_compress module "dxt" (-c dxt[:DXT1|:DXT1_YUV|:DXT5][:dev=<n>[,<n>...]]) backed by the MI355X
 * kernel library libug_mi355x.so (include/ug_mi355x.h).
 *
 * This is the host-side half of the drop-in boundary.  It is compiled against UltraGrid's own headers and
 * registers through lib_common.cpp exactly like the reference's GPU modules do (src/video_compress/cuda_dxt.cpp:278-290,
 * src/video_compress/gpujpeg.cpp:761-771).  API shape: the asynchronous frame API (push / pop), as gpujpeg.cpp uses it, on
 * top of mi355x_frame_sharder.h; the per-tile encoder below (init / compress_tile / done: lazy reconfigure on format change,
 * empty shared_ptr on error) is what every worker of the sharder runs.
 *
 * What is different from cuda_dxt.cpp by design (MI355X-first):
 *  - no CPU decoder_t pre-pass (cuda_dxt.cpp:206-220) and no 4:2:2->4:4:4 intermediate (cuda_dxt.cpp:223-232): the frame is
 *    uploaded in its wire format.  The module takes EVERY codec cuda_dxt.cpp takes -- whatever get_best_decoder_from(codec,
 *    {RGB, UYVY}) can reach (cuda_dxt.cpp:152-158, pixfmt_conv.c:3126-3172): RGB, RGBA, BGR, UYVY, YUYV, v210, R10k, R12L, RG48,
 *    Y216, Y416, VUYA, DVS10 -- and produces the bytes the reference's "line decoder, then encoder" sequence would: RGB, RGBA,
 *    UYVY and v210 (any width % 4 == 0) are unpacked + colour-converted + encoded by ONE fused kernel, the others go through the
 *    same decoders[] arithmetic on the device (ug_hip_pixfmt_convert to the codec ug_hip_pixfmt_best picks, == the reference's
 *    choice) and then the fused kernel; device-resident frames are encoded in place;
 *  - one HIP stream per encoder state, asynchronous H2D -> kernel -> D2H, a single stream synchronisation per tile
 *    (cuda_dxt.cu:759 synchronises after every launch);
 *  - devices come from the module option dev=<n>[,<n>...] (default 0); UltraGrid's -D list is capped at MAX_CUDA_DEVICES = 4
 *    (host.h:97), too small for an 8-GPU MI355X node.  Frames are dealt to one worker thread per listed device and delivered
 *    in order -- frames are independent, there is no inter-GPU traffic.
 *
 * There is deliberately no CPU fallback: a codec the reference refuses too (no decoder to RGB or UYVY: compressed and planar
 * codecs) is refused with the reference's message and the frame is dropped (video_compress.cpp:394-398 semantics).
 */
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "debug.h"
#include "host.h"
#include "lib_common.h"
#include "types.h"
#include "utils/video_frame_pool.h"
#include "video_codec.h"
#include "video_compress.h"
#include "video_frame.h"

#include "../../include/ug_mi355x.h"
#include "ug_codec_map.h"
#include "mi355x_frame_sharder.h"

#define MOD_NAME "[DXT MI355X] "

#define CHECK_HIP(cmd, msg, action) \
        if ((cmd) != UG_HIP_SUCCESS) { \
                MSG(ERROR, "%s: %s\n", msg, ug_hip_last_error_string()); \
                action; \
        }

namespace {

/// pinned host memory for the compressed output frames (cuda_dxt.cpp:68-83 does the same with CUDA)
struct hip_pinned_allocator : public video_frame_pool_allocator {
        void *allocate(size_t size) override {
                void *ptr = nullptr;
                if (ug_hip_malloc_host(&ptr, size) != UG_HIP_SUCCESS) {
                        return nullptr;
                }
                return ptr;
        }
        void deallocate(void *ptr) override { ug_hip_free_host(ptr); }
        video_frame_pool_allocator *clone() const override { return new hip_pinned_allocator(*this); }
};

struct state_video_compress_dxt_mi355x {
        struct video_desc saved_desc{};
        int               device = 0;
        codec_t           out_codec = DXT1;
        ug_dxt_t          out_fmt = UG_DXT1;
        int               ties = UG_DXT_TIES_DEFAULT; ///< ties=even|away (include/ug_mi355x.h UG_DXT_TIES_*)
        int               bands = 1;                ///< bands=<k>: a host frame is uploaded, encoded and downloaded in k row bands that overlap each other (latency of ONE frame)
        bool              deinterlace = true;       ///< deinterlace=no switches off RTDXT's automatic de-interlacing of INTERLACED_MERGED input
        bool              interlaced_input = false; ///< the current geometry is de-interlaced before encoding (dxt_glsl.cpp:195-201)
        ug_pixfmt_t       target = UG_PF_NONE;      ///< the 8-bit line format the reference would encode from (RGB or UYVY)
        ug_pixfmt_t       in_fmt = UG_PF_NONE;      ///< format the encoder kernel reads
        ug_pixfmt_t       pre_in = UG_PF_NONE;      ///< != NONE: device-side conversion first (YUYV->UYVY, BGR->RGB, DXT1_YUV: anything->UYVY)
        ug_pixfmt_t       pre_out = UG_PF_NONE;     ///< its target format
        ug_hip_stream_t   stream = nullptr;
        void             *dev_in = nullptr;         ///< uploaded frame, wire format
        void             *dev_pre = nullptr;        ///< swizzle result (only if pre_in != NONE)
        void             *dev_out = nullptr;        ///< DXT blocks
        size_t            in_len = 0, out_len = 0;
        // the same buffers once per frame of a batch: allocated on first use, batch_slices slices (= the module's batch=<n>, handed down by
        // the sharder as batch_slices=<n>; 16 if a caller hands batches to a state directly)
        int               batch_slices = 16;
        void             *b_in = nullptr, *b_pre = nullptr, *b_out = nullptr;
        size_t            b_in_stride = 0, b_pre_stride = 0, b_out_stride = 0;
        std::shared_ptr<video_frame_pool> pool = std::make_shared<video_frame_pool>(0, hip_pinned_allocator()); ///< shared with the frames it gives out (mi355x::get_frame_keeping_pool)
};

void cleanup(state_video_compress_dxt_mi355x *s)
{
        for (void **p : { &s->dev_in, &s->dev_pre, &s->dev_out, &s->b_in, &s->b_pre, &s->b_out }) {
                if (*p) {
                        ug_hip_free(*p);
                        *p = nullptr;
                }
        }
}

void usage()
{
        printf("MI355X DXT compression usage:\n"
               "\t-c dxt[:DXT1|:DXT1_YUV|:DXT5][:dev=<index>[,<index>...]][:workers=<per device>][:batch=<frames>][:bands=<k>][:numa=<0|1>][:ties=even|away][:deinterlace=no]\n"
               "\t\tdeinterlace=no - do not blend the lines of interlaced (merged) input before encoding (default: blended and sent as progressive, as RTDXT does)\n"
               "\t\tnuma  - 1 (default): every worker thread runs on the CPUs of its GPU's NUMA node, so that its pinned frame pool is local to the GPU; 0: left to the scheduler\n"
               "\t\tbands - cut every frame into <k> row bands (1-16, default 1): upload of band i+1, kernels of band i and download of band i-1 run at the same time -- the\n"
               "\t\t        latency of ONE frame drops towards its longer copy (8K v210: 2.30 -> 1.95 ms at k = 4, 8K UYVY: 1.90 -> 1.53; profiles/r05_row_bands.txt); same bytes; no effect on interlaced (de-interlaced) input\n"
               "\t\tbatch - frames a busy worker may queue and encode in one launch (1-16, default 1); only matters for sources faster than the encoder\n"
               "\t\tDXT1 - 4 bpp S3TC (default), DXT5 - 8 bpp DXT5-YCoCg, DXT1_YUV - DXT1 blocks holding Y,Cb,Cr\n"
               "\t\tdev  - HIP device index or list (default 0); the tiles of a frame are dealt out over the list\n"
               "\t\tties - what GLSL leaves to the implementation in the reference's encoder shaders: even (default) = round() ties to\n"
               "\t\t       even, as the shaders compute on Mesa; away = roundf(), as the reference's CUDA port is written\n");
}

void *dxt_mi355x_compress_init(struct module *parent, const char *fmt)
{
        (void) parent;
        auto *s = new state_video_compress_dxt_mi355x();
        std::string cfg = fmt ? fmt : "";
        size_t pos = 0;
        while (pos <= cfg.size() && !cfg.empty()) {
                size_t end = cfg.find(':', pos);
                std::string tok = cfg.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                if (strcasecmp(tok.c_str(), "DXT5") == 0) {
                        s->out_codec = DXT5;
                        s->out_fmt = UG_DXT5_YCOCG;
                } else if (strcasecmp(tok.c_str(), "DXT1") == 0) {
                        s->out_codec = DXT1;
                        s->out_fmt = UG_DXT1;
                } else if (strcasecmp(tok.c_str(), "DXT1_YUV") == 0) { // dxt_glsl.cpp:233-234
                        s->out_codec = DXT1_YUV;
                        s->out_fmt = UG_DXT1;
                } else if (strncasecmp(tok.c_str(), "dev=", 4) == 0) {
                        std::vector<int> devs;
                        for (const char *p = tok.c_str() + 4; *p;) {
                                devs.push_back(atoi(p));
                                p = strchr(p, ',');
                                if (!p) break;
                                p++;
                        }
                        if (devs.empty()) devs.push_back(0);
                        static std::atomic<unsigned> instance_counter{0}; // one init per tile: deal the instances out
                        s->device = devs[instance_counter++ % devs.size()];
                } else if (strncasecmp(tok.c_str(), "batch_slices=", 13) == 0) { // internal: from mi355x::sharded_init
                        s->batch_slices = atoi(tok.c_str() + 13);
                        if (s->batch_slices < 1 || s->batch_slices > 16) s->batch_slices = 16;
                } else if (strncasecmp(tok.c_str(), "bands=", 6) == 0) {
                        s->bands = atoi(tok.c_str() + 6);
                        if (s->bands < 1 || s->bands > 16) {
                                MSG(ERROR, "bands=<k> must be 1..16\n");
                                delete s;
                                return nullptr;
                        }
                } else if (strncasecmp(tok.c_str(), "deinterlace=", 12) == 0) {
                        const char *v = tok.c_str() + 12;
                        s->deinterlace = !(strcasecmp(v, "no") == 0 || strcasecmp(v, "off") == 0 || strcmp(v, "0") == 0);
                } else if (strcasecmp(tok.c_str(), "ties=even") == 0) {
                        s->ties = UG_DXT_TIES_EVEN;
                } else if (strcasecmp(tok.c_str(), "ties=away") == 0) {
                        s->ties = UG_DXT_TIES_AWAY;
                } else if (tok == "help") {
                        usage();
                        delete s;
                        return INIT_NOERR;
                } else if (!tok.empty()) {
                        MSG(ERROR, "unknown option: %s\n", tok.c_str());
                        usage();
                        delete s;
                        return nullptr;
                }
                if (end == std::string::npos) {
                        break;
                }
                pos = end + 1;
        }
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot use HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                delete s;
                return nullptr;
        }
        return s;
}

bool configure_with(state_video_compress_dxt_mi355x *s, struct video_desc desc)
{
        cleanup(s);
        s->pre_in = s->pre_out = UG_PF_NONE;
        const ug_pixfmt_t wire = ug_pixfmt_from_codec(desc.color_spec);
        // The reference converts the wire format on the CPU to what get_best_decoder_from(codec, {RGB, UYVY}) ranks first
        // (cuda_dxt.cpp:152-158; DXT1_YUV: UYVY only, dxt_glsl.cpp:104-110) and encodes that.  ug_hip_pixfmt_best is the same
        // ranking over the same decoders[] table (equal to the compiled reference on random candidate sets, tests/), so `target`
        // is the codec the reference would encode from.
        const ug_pixfmt_t cand_dxt[] = { UG_PF_RGB, UG_PF_UYVY, UG_PF_NONE }, cand_yuv[] = { UG_PF_UYVY, UG_PF_NONE };
        ug_pixfmt_t target = UG_PF_NONE;
        if (wire == UG_PF_NONE || wire == UG_PF_I420 ||
            ug_hip_pixfmt_best(wire, s->out_codec == DXT1_YUV ? cand_yuv : cand_dxt, &target) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Unsupported codec: %s\n", get_codec_name(desc.color_spec)); // cuda_dxt.cpp:155-157
                return false;
        }
        // What the fused kernel reads natively gives the same bytes as "decode to target, then encode": RGBA -> RGB only drops
        // alpha (vc_copylineRGBAtoRGB), v210 -> UYVY is the >> 2 the v210 loader applies (vc_copylinev210, incl. its partial-group
        // tail: any width % 4 == 0).  Everything else is converted to `target` on the device first, with the decoders[] arithmetic.
        // INTERLACED_MERGED input: RTDXT -- the module this one stands in for (BASELINE configs[1]) -- blends the lines of the DECODED 8-bit frame
        // before it encodes and announces the result as progressive (dxt_glsl.cpp:195-201,291-293 -> vc_deinterlace); cuda_dxt.cpp has no such
        // step.  Done here on the device, on the frame in the `target` format: the shortcuts that skip that intermediate are not taken then.
        s->interlaced_input = s->deinterlace && desc.interlacing == INTERLACED_MERGED;
        s->target = target;
        if (s->interlaced_input && vc_get_linesize(desc.width, ug_codec_from_pixfmt(target)) < 16) {
                // (below one 16-byte column vc_deinterlace's vectors overlap themselves: ug_hip_deinterlace_blend refuses such lines)
                MSG(WARNING, "Pictures %u pixels wide are not de-interlaced.\n", desc.width);
                s->interlaced_input = false;
        }
        if (s->interlaced_input) {
                MSG(NOTICE, "Enabling automatic deinterlacing.\n");
        }
        const bool fused = wire == target || (!s->interlaced_input && ((wire == UG_PF_RGBA && target == UG_PF_RGB) || (wire == UG_PF_V210 && target == UG_PF_UYVY)));
        if (fused) {
                s->in_fmt = wire;
        } else {
                s->pre_in = wire;
                s->pre_out = s->in_fmt = target;
        }
        if (s->out_codec == DXT1_YUV && s->in_fmt == UG_PF_UYVY) {
                // DXT1 over the 4:2:2 samples themselves (chroma replicated, no colour conversion: dxt_encoder.c:318-323)
                s->in_fmt = UG_PF_UYVY_RAW;
        } else if (s->out_codec == DXT1_YUV) { // v210: the encoder's v210 loader colour-converts, so go through UYVY here
                s->pre_in = wire;
                s->pre_out = UG_PF_UYVY;
                s->in_fmt = UG_PF_UYVY_RAW;
        }
        // Any frame size, as RTDXT takes it (dxt_glsl.cpp:150-160 -> dxt_encoder_create; the stream holds whole blocks, dxt_util.h:59-67) --
        // not cuda_dxt.cu:745's multiples of 4.  What the encoder does past the picture's edge: include/ug_mi355x.h, ug_hip_dxt_encode.
        if (desc.width % 2 != 0 && (s->in_fmt == UG_PF_UYVY || s->in_fmt == UG_PF_UYVY_RAW || s->in_fmt == UG_PF_V210)) {
                MSG(ERROR, "A 4:2:2 frame %u pixels wide is not made of pixel pairs\n", desc.width);
                return false;
        }
        if (get_bits_per_component(desc.color_spec) > 8) {
                MSG(NOTICE, "Converting from %d to 8 bits on the GPU.\n", get_bits_per_component(desc.color_spec));
        }
        s->in_len = (size_t) vc_get_linesize(desc.width, desc.color_spec) * desc.height;
        s->out_len = ug_hip_dxt_size(s->out_fmt, (int) desc.width, (int) desc.height);
        CHECK_HIP(ug_hip_malloc(&s->dev_in, s->in_len + MAX_PADDING), "Could not allocate device input buffer", return false);
        if (s->pre_in != UG_PF_NONE) {
                const size_t pre_len = (size_t) vc_get_linesize(desc.width, ug_codec_from_pixfmt(s->pre_out)) * desc.height;
                CHECK_HIP(ug_hip_malloc(&s->dev_pre, pre_len + MAX_PADDING), "Could not allocate device conversion buffer", return false);
        }
        CHECK_HIP(ug_hip_malloc(&s->dev_out, s->out_len), "Could not allocate device output buffer", return false);

        struct video_desc compressed_desc = desc;
        compressed_desc.color_spec = s->out_codec;
        compressed_desc.tile_count = 1;
        if (s->interlaced_input) {
                compressed_desc.interlacing = PROGRESSIVE; // dxt_glsl.cpp:196-198
        }
        s->pool->reconfigure(compressed_desc, s->out_len);
        return true;
}

/// bands=<k>: ONE host frame as k row bands -- the upload of band i + 1 (upload lane), the kernels of band i (the state's stream) and the download of band
/// i - 1 (download lane) run at the same time, so the frame is out after about its longer copy instead of upload + kernels + download in a row
/// (SURVEY.md 8(e) "tile-level split of a single 8K frame" / N4; the reference's only intra-frame parallelism is the tile fan-out of
/// video_compress.cpp:441-490, which needs a tiled source).  DXT blocks do not see their neighbours, line converters do not see other lines: the bytes
/// are those of the whole-frame path.  Band edges lie on multiples of 16 lines, which keeps every band's source and block addresses 16-byte aligned.
std::shared_ptr<video_frame> compress_tile_in_bands(state_video_compress_dxt_mi355x *s, const std::shared_ptr<video_frame> &tx, int w, int h)
{
        std::shared_ptr<video_frame> out = mi355x::get_frame_keeping_pool(s->pool);
        const size_t wire_ls = (size_t) vc_get_linesize(w, tx->color_spec);
        const bool pre = s->pre_in != UG_PF_NONE;
        const size_t pre_ls = pre ? (size_t) vc_get_linesize(w, ug_codec_from_pixfmt(s->pre_out)) : 0;
        const size_t out_row = s->out_len / (size_t) ((h + 3) / 4); // bytes of one row of blocks
        // a failure half way: downloads of earlier bands may still be writing into `out` (they were not joined) -- wait for the lane before the frame goes
        // back to its pool (a joined 16-byte download behind them, then the stream)
        auto fail = [&]() -> std::shared_ptr<video_frame> {
                (void) ug_hip_download_ordered_ex(s->device, out->tiles[0].data, s->dev_out, 16, s->stream, 0);
                (void) ug_hip_stream_sync(s->stream);
                return {};
        };
        int r0 = 0;
        for (int k = 0; k < s->bands && r0 < h; k++) {
                const int r1 = k == s->bands - 1 ? h : std::min(h, (int) ((long) h * (k + 1) / s->bands + 15) / 16 * 16);
                if (r1 <= r0) continue;
                const int rows = r1 - r0;
                const bool last = r1 == h;
                char *const src_band = (char *) s->dev_in + (size_t) r0 * wire_ls;
                // the first band's upload waits for whatever this state's stream still holds; the others follow it on the upload lane
                CHECK_HIP(ug_hip_upload_ordered_ex(s->device, src_band, tx->tiles[0].data + (size_t) r0 * wire_ls, (size_t) rows * wire_ls, UG_HIP_MEMCPY_HOST_TO_DEVICE,
                                                   s->stream, k == 0 ? 0 : UG_HIP_COPY_NO_WAIT),
                          "upload failed", return fail());
                const void *enc_src = src_band;
                if (pre) {
                        char *const pre_band = (char *) s->dev_pre + (size_t) r0 * pre_ls;
                        CHECK_HIP(ug_hip_pixfmt_convert(s->pre_in, s->pre_out, src_band, pre_band, w, rows, 0, 0, 0, 8, 16, s->stream), "device swizzle failed", return fail());
                        enc_src = pre_band;
                }
                char *const blocks = (char *) s->dev_out + (size_t) (r0 / 4) * out_row;
                CHECK_HIP(ug_hip_dxt_encode_batch_ex(s->in_fmt, s->out_fmt, enc_src, blocks, w, rows, 0, 1, 0, 0, s->ties, s->stream), "Encoding failed", return fail());
                // every download but the last leaves the stream free to go on with the next band; the last one joins, and the lane is in order
                CHECK_HIP(ug_hip_download_ordered_ex(s->device, out->tiles[0].data + (size_t) (r0 / 4) * out_row, blocks, (size_t) ((rows + 3) / 4) * out_row, s->stream,
                                                     last ? 0 : UG_HIP_COPY_NO_JOIN),
                          "D2H copy failed", return fail());
                r0 = r1;
        }
        CHECK_HIP(ug_hip_stream_sync(s->stream), "stream sync failed", return {});
        out->tiles[0].data_len = (unsigned int) s->out_len;
        return out;
}

std::shared_ptr<video_frame> dxt_mi355x_compress_tile(void *state, std::shared_ptr<video_frame> tx)
{
        if (!tx) {
                return {}; // poison pill (video_compress.cpp:345-347)
        }
        auto *s = static_cast<state_video_compress_dxt_mi355x *>(state);
        CHECK_HIP(ug_hip_set_device(s->device), "set device", return {}); // tile callbacks run on pool threads

        if (!video_desc_eq_excl_param(video_desc_from_frame(tx.get()), s->saved_desc, PARAM_TILE_COUNT)) {
                if (configure_with(s, video_desc_from_frame(tx.get()))) {
                        s->saved_desc = video_desc_from_frame(tx.get());
                } else {
                        MSG(ERROR, "Reconfiguration failed!\n");
                        s->saved_desc = {};
                        return {};
                }
        }
        const int w = (int) tx->tiles[0].width, h = (int) tx->tiles[0].height;

        // Device-resident frame (mem_location == CUDA_MEM, types.h:295-298; the tile fan-out of video_compress.cpp drops that flag,
        // so the pointer itself is asked as well): no upload -- the kernels read it in place (gpujpeg.cpp:617-622 does the same).
        const void *enc_src = s->dev_in;
        // In place only when the frame lives on THIS state's GPU: with dev=<list> / several workers a frame can be handed to a
        // worker of another device, whose kernels must not dereference foreign memory -- that one is copied over (peer copy).
        const bool blend_in_place = s->interlaced_input && s->pre_in == UG_PF_NONE; // the blend works in place: never on the caller's frame
        if (!blend_in_place && ug_hip_pointer_device(tx->tiles[0].data) == s->device && ((uintptr_t) tx->tiles[0].data & 15) == 0) {
                enc_src = tx->tiles[0].data;
        } else {
                const bool dev = tx->mem_location == CUDA_MEM || ug_hip_pointer_is_device(tx->tiles[0].data);
                if (s->bands > 1 && !dev && !s->interlaced_input && h >= 32) { // (the blend runs down the whole frame: no bands there)
                        return compress_tile_in_bands(s, tx, w, h);
                }
                CHECK_HIP(ug_hip_upload_ordered(s->device, s->dev_in, tx->tiles[0].data, s->in_len, dev ? UG_HIP_MEMCPY_DEVICE_TO_DEVICE : UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream),
                          "upload failed", return {});
        }
        const void *const wire_src = enc_src;
        if (s->pre_in != UG_PF_NONE) {
                CHECK_HIP(ug_hip_pixfmt_convert(s->pre_in, s->pre_out, wire_src, s->dev_pre, w, h, 0, 0, 0, 8, 16, s->stream),
                          "device swizzle failed", return {});
                enc_src = s->dev_pre;
        }
        if (s->interlaced_input) { // vc_deinterlace on the decoded frame (dxt_glsl.cpp:291-293): dev_pre, or the uploaded copy in dev_in
                CHECK_HIP(ug_hip_deinterlace_blend(const_cast<void *>(enc_src), (size_t) vc_get_linesize(w, ug_codec_from_pixfmt(s->target)), h, s->stream),
                          "de-interlacing failed", return {});
        }
        CHECK_HIP(ug_hip_dxt_encode_batch_ex(s->in_fmt, s->out_fmt, enc_src, s->dev_out, w, h, 0, 1, 0, 0, s->ties, s->stream),
                  "Encoding failed", return {});

        std::shared_ptr<video_frame> out = mi355x::get_frame_keeping_pool(s->pool);
        CHECK_HIP(ug_hip_download_ordered(s->device, out->tiles[0].data, s->dev_out, s->out_len, s->stream), "D2H copy failed", return {});
        CHECK_HIP(ug_hip_stream_sync(s->stream), "stream sync failed", return {});
        out->tiles[0].data_len = (unsigned int) s->out_len;
        return out;
}

/// `frames` queued frames of one geometry: uploads (and device-side conversions) into the slices of the batch buffers, ONE encoder
/// launch over all of them (ug_hip_dxt_encode_batch_ex, grid.z = frame), the downloads, one synchronisation.  Same bytes as one by one.
std::vector<std::shared_ptr<video_frame>> dxt_mi355x_compress_batch(void *state, std::vector<std::shared_ptr<video_frame>> in)
{
        auto *s = static_cast<state_video_compress_dxt_mi355x *>(state);
        const int n = (int) in.size();
        std::vector<std::shared_ptr<video_frame>> out(in.size());
        auto one_by_one = [&] {
                for (size_t i = 0; i < in.size(); i++) out[i] = dxt_mi355x_compress_tile(state, std::move(in[i]));
                return out;
        };
        if (n < 2 || n > s->batch_slices || ug_hip_set_device(s->device) != UG_HIP_SUCCESS ||
            !video_desc_eq_excl_param(video_desc_from_frame(in[0].get()), s->saved_desc, PARAM_TILE_COUNT)) {
                return one_by_one(); // (the first frame of a new geometry configures the state on the way)
        }
        const int w = (int) in[0]->tiles[0].width, h = (int) in[0]->tiles[0].height;
        if (s->b_in == nullptr) {
                auto round16 = [](size_t v) { return (v + 15) / 16 * 16; };
                s->b_in_stride = round16(s->in_len + MAX_PADDING);
                s->b_out_stride = round16(s->out_len);
                bool ok = ug_hip_malloc(&s->b_in, s->b_in_stride * s->batch_slices) == UG_HIP_SUCCESS &&
                          ug_hip_malloc(&s->b_out, s->b_out_stride * s->batch_slices) == UG_HIP_SUCCESS;
                if (ok && s->pre_in != UG_PF_NONE) {
                        s->b_pre_stride = round16((size_t) vc_get_linesize(w, ug_codec_from_pixfmt(s->pre_out)) * h + MAX_PADDING);
                        ok = ug_hip_malloc(&s->b_pre, s->b_pre_stride * s->batch_slices) == UG_HIP_SUCCESS;
                }
                if (!ok) {
                        MSG(WARNING, "no device memory for the batch buffers (%s): frames are encoded one by one\n", ug_hip_last_error_string());
                        for (void **p : { &s->b_in, &s->b_pre, &s->b_out }) {
                                if (*p) { ug_hip_free(*p); *p = nullptr; }
                        }
                        return one_by_one();
                }
        }
        // all the uploads of the batch first, then the conversions: the upload lane of the device is shared by every worker, and an upload queued
        // behind this worker's conversion kernels would keep the other workers' uploads waiting with it
        for (int f = 0; f < n; f++) {
                const bool dev = in[f]->mem_location == CUDA_MEM || ug_hip_pointer_is_device(in[f]->tiles[0].data);
                char *slice = (char *) s->b_in + f * s->b_in_stride;
                CHECK_HIP(ug_hip_upload_ordered(s->device, slice, in[f]->tiles[0].data, s->in_len, dev ? UG_HIP_MEMCPY_DEVICE_TO_DEVICE : UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream),
                          "upload failed", return out);
        }
        for (int f = 0; f < n && s->pre_in != UG_PF_NONE; f++) {
                CHECK_HIP(ug_hip_pixfmt_convert(s->pre_in, s->pre_out, (char *) s->b_in + f * s->b_in_stride, (char *) s->b_pre + f * s->b_pre_stride, w, h, 0, 0, 0, 8, 16, s->stream),
                          "device swizzle failed", return out);
        }
        const bool pre = s->pre_in != UG_PF_NONE;
        if (s->interlaced_input) {
                CHECK_HIP(ug_hip_deinterlace_blend_batch(pre ? s->b_pre : s->b_in, (size_t) vc_get_linesize(w, ug_codec_from_pixfmt(s->target)), h, n,
                                                         pre ? s->b_pre_stride : s->b_in_stride, s->stream),
                          "de-interlacing failed", return out);
        }
        CHECK_HIP(ug_hip_dxt_encode_batch_ex(s->in_fmt, s->out_fmt, pre ? s->b_pre : s->b_in, s->b_out, w, h, 0, n, pre ? s->b_pre_stride : s->b_in_stride, s->b_out_stride,
                                             s->ties, s->stream),
                  "Encoding failed", return out);
        for (int f = 0; f < n; f++) {
                out[f] = mi355x::get_frame_keeping_pool(s->pool);
                CHECK_HIP(ug_hip_download_ordered(s->device, out[f]->tiles[0].data, (char *) s->b_out + f * s->b_out_stride, s->out_len, s->stream), "D2H copy failed",
                          { for (auto &o : out) o.reset(); return out; });
                out[f]->tiles[0].data_len = (unsigned int) s->out_len;
        }
        CHECK_HIP(ug_hip_stream_sync(s->stream), "stream sync failed", { for (auto &o : out) o.reset(); return out; });
        return out;
}

void dxt_mi355x_compress_done(void *state)
{
        auto *s = static_cast<state_video_compress_dxt_mi355x *>(state);
        ug_hip_set_device(s->device);
        cleanup(s);
        if (s->stream) {
                ug_hip_stream_destroy(s->stream);
        }
        delete s;
}

compress_module_info get_dxt_mi355x_module_info()
{
        compress_module_info module_info;
        module_info.name = "dxt";
        for (const char *c : { "DXT1", "DXT1_YUV", "DXT5" }) {
                codec codec_info;
                codec_info.name = c;
                codec_info.priority = 400;
                codec_info.encoders.emplace_back(encoder{ "default", std::string(":") + c });
                module_info.codecs.emplace_back(std::move(codec_info));
        }
        return module_info;
}

/// module-level init: consumes dev=<list>, creates one worker (thread + per-tile encoder states) per listed device
void *dxt_mi355x_module_init(struct module *parent, const char *cfg)
{
        return mi355x::sharded_init(parent, cfg, dxt_mi355x_compress_init, dxt_mi355x_compress_tile, dxt_mi355x_compress_done, ug_hip_set_device,
                                    dxt_mi355x_compress_batch, ug_hip_bind_thread_to_device, ug_hip_device_numa_node);
}

const struct video_compress_info dxt_mi355x_info = {
        dxt_mi355x_module_init,
        mi355x::sharded_done,
        NULL,
        NULL,
        mi355x::sharded_push, // asynchronous frame API: frames are dealt to one worker per listed GPU and popped in order
        mi355x::sharded_pop,
        NULL,
        NULL,
        get_dxt_mi355x_module_info,
};

// "dxt" is a free name in the reference (it registers rtdxt, cuda_dxt, gpujpeg/jpeg)
REGISTER_MODULE(dxt, &dxt_mi355x_info, LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION);

} // end of anonymous namespace
