/**
 * @file vcompress_jpeg_mi355x.cpp
 * UltraGrid video_compress module "jpeg" (-c jpeg[:q=<1-100>][:restart=<MCUs>][:subsampling=<444|422|420>][:dev=<n>]) backed by the MI355X kernel
 * library (include/ug_mi355x.h: ug_hip_jpeg_encoder_*).  It occupies the name the reference registers as a hidden alias
 * of its GPUJPEG module (src/video_compress/gpujpeg.cpp:791-792) and follows that module's conventions:
 * quality / restart-interval options (gpujpeg.cpp:279-285,345-352,479-485), UYVY handed to the encoder as 4:2:x YCbCr in
 * BT.709 limited range without conversion (gpujpeg.cpp:303-305,329-339), output codec JPEG with restart intervals.
 *
 * Differences by design: 4:2:0 baseline stream from a fused UYVY -> 4:2:0 -> FDCT -> quantise kernel; every input
 * conversion (v210 / YUYV / RGB / RGBA / BGR -> UYVY: the pixfmt_conv.c arithmetic) runs on the device instead of the CPU
 * line loop of gpujpeg.cpp:592-608; tile API with one stream per module instance; no CPU fallback.
 */
#include <algorithm>
#include <cctype>
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

#define MOD_NAME "[JPEG MI355X] "

namespace {

struct hip_pinned_allocator : public video_frame_pool_allocator {
        void *allocate(size_t size) override {
                void *ptr = nullptr;
                return ug_hip_malloc_host(&ptr, size) == UG_HIP_SUCCESS ? ptr : nullptr;
        }
        void deallocate(void *ptr) override { ug_hip_free_host(ptr); }
        video_frame_pool_allocator *clone() const override { return new hip_pinned_allocator(*this); }
};

struct state_video_compress_jpeg_mi355x {
        struct video_desc    saved_desc{};
        int                  device = 0, quality = 75, restart = 2;
        int                  subsampling = 0;  ///< 0 = autoselect: that of the input codec (gpujpeg.cpp:168,295-302)
        int                  internal_cs = 0;  ///< color_space_internal asked for (gpujpeg.cpp:398-405): 0 = as the input dictates, else UG_JPEG_CS_RGB / _YCBCR_BT601 / _YCBCR_BT601_256LVLS / _YCBCR_BT709
        bool                 force_interleaved = false; ///< `:interleaved` (gpujpeg.cpp:396-397): RGB input as ONE scan instead of one scan per component
        bool                 alpha = false;    ///< `:alpha` (gpujpeg.cpp:409-414)
        ug_pixfmt_t          wire = UG_PF_NONE;     ///< format of the uploaded frame
        ug_pixfmt_t          target = UG_PF_NONE;   ///< what the reference's CPU line decoder would convert it to (UYVY, RGB or RGBA)
        ug_pixfmt_t          enc_in = UG_PF_NONE;   ///< what the encoder is fed: UYVY, RGB or I420
        ug_hip_stream_t      stream = nullptr;
        ug_hip_jpeg_encoder *enc = nullptr;
        void                *dev_in = nullptr, *dev_target = nullptr, *dev_uyvy = nullptr, *dev_out = nullptr;
        size_t               in_len = 0, max_out = 0;
        // the same buffers once per frame of a batch (batch=<n>): grown on first use
        int                  batch_cap = 0;
        int                  batch_slices = 16; // slices to allocate: the module's batch=<n>, handed down by the sharder as batch_slices=<n>
        size_t               b_in_stride = 0, b_target_stride = 0, b_enc_stride = 0, b_out_stride = 0;
        void                *b_in = nullptr, *b_target = nullptr, *b_enc = nullptr, *b_out = nullptr;
        std::shared_ptr<video_frame_pool> pool = std::make_shared<video_frame_pool>(0, hip_pinned_allocator()); ///< shared with the frames it gives out (mi355x::get_frame_keeping_pool)
};

void cleanup(state_video_compress_jpeg_mi355x *s)
{
        if (s->enc) { ug_hip_jpeg_encoder_destroy(s->enc); s->enc = nullptr; }
        for (void **p : { &s->dev_in, &s->dev_target, &s->dev_uyvy, &s->dev_out, &s->b_in, &s->b_target, &s->b_enc, &s->b_out }) {
                if (*p) { ug_hip_free(*p); *p = nullptr; }
        }
        s->batch_cap = 0;
}

void usage()
{
        printf("MI355X JPEG compression usage:\n"
               "\t-c jpeg[:<quality>[:<restart>]][:q=<quality 1-100>][:restart=<MCUs per restart interval; 0 = none>][:subsampling=<444|422|420>][:interleaved][:RGB|:Y601|:Y601full|:Y709][:alpha][:dev=<index>[,<index>...]][:workers=<per device>][:batch=<frames>][:numa=<0|1>]\n"
               "\t\tnuma        - 1 (default): every worker thread runs on the CPUs of its GPU's NUMA node (pinned frame pool local to the GPU); 0: left to the scheduler\n"
               "\t\tbatch       - frames a busy worker may queue and encode together (1-16, default 1); only matters for sources faster than the encoder\n"
               "\t\tinterleaved - RGB input as one interleaved scan; default (as the reference's): one scan per component -- three coder launches, a little slower\n"
               "\t\tRGB | Y601 | Y601full | Y709 - colour space the samples are coded in (default: R,G,B for RGB input, BT.709 limited range for the rest)\n"
               "\t\tsubsampling - JPEG subsampling; default = that of the codec the input is decoded to (get_best_decoder_from over\n"
               "\t\t              UYVY, RGB, RGBA): 422 for UYVY/YUYV/v210/Y216/DVS10, 444 (R,G,B components) for\n"
               "\t\t              RGB/RGBA/BGR/R10k/R12L/RG48/Y416/VUYA, 420 for I420; 420 from 4:2:2 input averages line pairs,\n"
               "\t\t              444 from 4:2:2 input gives every pixel its pair's chroma (Y'CbCr, or R,G,B with :RGB)\n");
}

/// the device-side decoder_t pass wire -> target; planar I420 (which has no decoder_t) with the reference's i420_8_to_uyvy shuffle
int wire_to_target(state_video_compress_jpeg_mi355x *s, const void *src, void *dst, int w, int h)
{
        if (s->wire == UG_PF_I420) {
                const int cw = (w + 1) / 2, ch = (h + 1) / 2;
                const char *y = (const char *) src, *u = y + (size_t) w * h, *v = u + (size_t) cw * ch;
                return ug_hip_yuv420p_to_uyvy(y, w, u, cw, v, cw, dst, 2 * w, w, h, s->stream);
        }
        return ug_hip_pixfmt_convert(s->wire, s->target, src, dst, w, h, 0, 0, 0, 8, 16, s->stream);
}

/// IS_KEY_PREFIX (utils/macros.h:162-164): tok is <k>=<v> and <k> is a (non-empty) prefix of key
bool key_prefix(const std::string &tok, const char *key)
{
        const size_t eq = tok.find('=');
        return eq != std::string::npos && eq > 0 && eq <= strlen(key) && strncmp(key, tok.c_str(), eq) == 0;
}

void *jpeg_mi355x_compress_init(struct module *parent, const char *fmt)
{
        (void) parent;
        auto *s = new state_video_compress_jpeg_mi355x();
        std::string cfg = fmt ? fmt : "";
        size_t pos = 0;
        int numeric = 0;
        while (!cfg.empty() && pos <= cfg.size()) {
                size_t end = cfg.find(':', pos);
                std::string tok = cfg.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                if (!tok.empty() && isdigit((unsigned char) tok[0])) { // gpujpeg.cpp:379-391: "-c gpujpeg:<quality>[:<restart interval>]"
                        (numeric++ == 0 ? s->quality : s->restart) = atoi(tok.c_str());
                } else if (strncasecmp(tok.c_str(), "dev=", 4) == 0) {
                        s->device = atoi(tok.c_str() + 4);
                } else if (strncasecmp(tok.c_str(), "batch_slices=", 13) == 0) { // internal: from mi355x::sharded_init
                        s->batch_slices = atoi(tok.c_str() + 13);
                        if (s->batch_slices < 1 || s->batch_slices > 16) s->batch_slices = 16;
                } else if (key_prefix(tok, "quality")) { // <k>=<v> with <k> any prefix of the key, as IS_KEY_PREFIX reads it (utils/macros.h:162-164; gpujpeg.cpp:392-395,406): q=, qual=, r=, sub= ...
                        s->quality = atoi(strchr(tok.c_str(), '=') + 1);
                } else if (key_prefix(tok, "restart")) {
                        s->restart = atoi(strchr(tok.c_str(), '=') + 1);
                } else if (key_prefix(tok, "subsampling")) {
                        s->subsampling = atoi(strchr(tok.c_str(), '=') + 1); // gpujpeg.cpp:406-408
                } else if (key_prefix(tok, "interleaved") || (tok.find('=') == std::string::npos && strncmp("interleaved", tok.c_str(), tok.size()) == 0)) { // IS_PREFIX: "i", "inter", "interleaved[=…]"
                        s->force_interleaved = true; // gpujpeg.cpp:396-397: one interleaved scan for RGB input too (the default there: one scan per component, :303)
                } else if (strcasecmp(tok.c_str(), "RGB") == 0) { // gpujpeg.cpp:398-405: color_space_internal; what it means for the input at hand: configure_with
                        s->internal_cs = UG_JPEG_CS_RGB;
                } else if (strcasecmp(tok.c_str(), "Y709") == 0) {
                        s->internal_cs = UG_JPEG_CS_YCBCR_BT709;
                } else if (strcasecmp(tok.c_str(), "Y601") == 0) {
                        s->internal_cs = UG_JPEG_CS_YCBCR_BT601;
                } else if (strcasecmp(tok.c_str(), "Y601full") == 0) {
                        s->internal_cs = UG_JPEG_CS_YCBCR_BT601_256LVLS;
                } else if (tok == "alpha") {
                        s->alpha = true; // gpujpeg.cpp:409-414; decided against the input at configure (:318-330)
                } else if (tok == "help") {
                        usage();
                        delete s;
                        return INIT_NOERR;
                } else if (tok == "check" || tok == "list_devices") { // gpujpeg.cpp:530-542: is there a device to run on / which ones
                        int count = 0;
                        const bool ok = ug_hip_device_count(&count) == UG_HIP_SUCCESS && count > 0;
                        if (tok == "list_devices") printf("HIP devices: %d\n", ok ? count : 0);
                        delete s;
                        return ok ? INIT_NOERR : nullptr;
                } else if (!tok.empty()) {
                        MSG(ERROR, "unknown option: %s\n", tok.c_str());
                        usage();
                        delete s;
                        return nullptr;
                }
                if (end == std::string::npos) break;
                pos = end + 1;
        }
        if (s->subsampling != 0 && s->subsampling != 420 && s->subsampling != 422 && s->subsampling != 444) {
                MSG(ERROR, "subsampling must be 444, 422 or 420\n");
                delete s;
                return nullptr;
        }
        if (s->quality < 1 || s->quality > 100 || s->restart < 0 || s->restart > 65535) {
                MSG(ERROR, "quality must be 1-100 and restart 0-65535\n");
                delete s;
                return nullptr;
        }
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS || ug_hip_stream_create(&s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "cannot use HIP device %d: %s\n", s->device, ug_hip_last_error_string());
                delete s;
                return nullptr;
        }
        return s;
}

bool configure_with(state_video_compress_jpeg_mi355x *s, struct video_desc desc)
{
        cleanup(s);
        s->wire = ug_pixfmt_from_codec(desc.color_spec);
        s->target = s->wire;
        // What the encoder is fed and which sampling it codes, as the reference decides it (gpujpeg.cpp:227-236,262-272,295-305,
        // 333-344): I420 passes through; everything else is converted to what get_best_decoder_from(codec, {UYVY, RGB, RGBA}) ranks
        // first (ug_hip_pixfmt_best: same ranking, same decoders[] table) -- here on the device, with the same arithmetic -- and
        // autoselect codes that codec's own subsampling: R,G,B 4:4:4 for an RGB-family target (R10k, R12L, RG48, but also Y416 and
        // VUYA: the ranking puts subsampling before colour space), 4:2:2 for UYVY (v210, Y216, YUYV, DVS10).  An RGBA target is fed as
        // RGB (the pad byte is ignored by GPUJPEG's 444_U8_P012Z too).  RGB-family targets with subsampling=422/420 go through
        // the pixfmt_conv.c RGB->UYVY arithmetic.
        bool rgb_family = false;
        // Planar I420 goes to the encoder as it is (GPUJPEG_420_U8_P0P1P2, :335) -- unless an option asks for something a 4:2:0 planar picture is not
        // (another subsampling, BT.601, R,G,B): GPUJPEG resamples / converts such input in its preprocessor; here the picture is first brought to UYVY
        // with the reference's own i420_8_to_uyvy shuffle (video_codec.c:1073-1094: both lines of a pair take the chroma line) and then treated as UYVY input.
        const bool planar_as_it_is = s->wire == UG_PF_I420 && (s->subsampling == 0 || s->subsampling == 420) &&
                                     (s->internal_cs == UG_JPEG_CS_ASIS || s->internal_cs == UG_JPEG_CS_YCBCR_BT709);
        if (planar_as_it_is) {
                s->enc_in = UG_PF_I420;
        } else if (s->wire == UG_PF_I420) {
                if (desc.width % 2) {
                        MSG(ERROR, "I420 %u pixels wide with subsampling= / a colour space option: the conversion goes through UYVY, which needs pixel pairs\n", desc.width);
                        return false;
                }
                s->target = UG_PF_UYVY;
        } else {
                const ug_pixfmt_t cand[] = { UG_PF_UYVY, UG_PF_RGB, UG_PF_RGBA, UG_PF_NONE };
                if (s->wire == UG_PF_NONE || ug_hip_pixfmt_best(s->wire, cand, &s->target) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "Unsupported codec: %s\n", get_codec_name(desc.color_spec)); // gpujpeg.cpp:267-271
                        return false;
                }
                rgb_family = s->target != UG_PF_UYVY;
        }
        const int sub = s->subsampling ? s->subsampling : (rgb_family ? 444 : (s->wire == UG_PF_I420 ? 420 : 422));     // (get_subsampling(m_enc_input_codec), gpujpeg.cpp:297-302)
        // color_space_internal (gpujpeg.cpp:303-305: the option, else RGB for RGB input and BT.709 for the rest).  4:4:4 from an RGB-family input: R, G, B
        // as they are, or converted to the Y'CbCr space asked for; 4:2:x (UYVY, or RGB-family input brought to UYVY with pixfmt_conv.c's BT.709
        // arithmetic): BT.709 limited range as the samples are, or converted to BT.601.  subsampling=444 on a 4:2:2 source: every pixel with its pair's
        // chroma, coded as BT.709 / BT.601 Y'CbCr or as R, G, B.  Not done: R, G, B components subsampled (4:2:x + RGB); planar input converted.
        int enc_cs = UG_JPEG_CS_ASIS;
        const bool uyvy_as_444 = sub == 444 && !rgb_family;
        if (uyvy_as_444) {
                enc_cs = s->internal_cs;
        } else if (sub == 444) {
                enc_cs = s->internal_cs == UG_JPEG_CS_RGB ? UG_JPEG_CS_ASIS : s->internal_cs;
        } else if (s->internal_cs == UG_JPEG_CS_RGB) {
                MSG(ERROR, "internal colour space RGB: R, G, B components are coded 4:4:4 only (add subsampling=444)\n");
                return false;
        } else if (s->internal_cs == UG_JPEG_CS_YCBCR_BT601 || s->internal_cs == UG_JPEG_CS_YCBCR_BT601_256LVLS) {
                enc_cs = s->internal_cs;
        }
        // one scan per component for RGB input unless `:interleaved` (gpujpeg.cpp:303) -- where the components are not subsampled (the reference
        // writes subsampled RGB-input streams that way too; here those are 4:2:x Y'CbCr streams of one scan)
        const int enc_flags = (rgb_family && sub == 444 && !s->force_interleaved ? UG_JPEG_NONINTERLEAVED : 0) | (uyvy_as_444 ? UG_JPEG_INPUT_UYVY : 0);
        if (s->alpha) { // gpujpeg.cpp:318-330
                if (desc.color_spec == RGBA) {
                        MSG(ERROR, "alpha: a fourth component is not coded by this encoder (the reference needs GPUJPEG >= 0.20.2 for it, gpujpeg.cpp:410-413); "
                                   "without the option the colour planes are coded and the pad byte dropped\n");
                        return false;
                }
                MSG(WARNING, "Requested alpha encode but input codec is unsupported pixel format: %s\n", get_codec_name(desc.color_spec)); // :327-328, and on it goes
        }
        if (planar_as_it_is) {
                // (s->enc_in = I420, above)
        } else if (sub == 444 && rgb_family) {
                s->enc_in = UG_PF_RGB;
        } else {
                s->enc_in = UG_PF_UYVY;
        }
        s->in_len = s->wire == UG_PF_I420 ? (size_t) desc.width * desc.height + 2 * (size_t) ((desc.width + 1) / 2) * ((desc.height + 1) / 2)
                                          : (size_t) vc_get_linesize(desc.width, desc.color_spec) * desc.height;
        if (ug_hip_jpeg_encoder_create_ex((int) desc.width, (int) desc.height, s->quality, s->restart, sub, enc_cs, enc_flags, &s->enc) != UG_HIP_SUCCESS) {
                MSG(ERROR, "encoder creation failed: %s\n", ug_hip_last_error_string());
                return false;
        }
        // Output capacity: the never-overflowing bound of the kernel library is ~10 B per pixel; a JPEG bigger than the raw RGB
        // frame only comes from noise at q=100, so the pooled (pinned) frames are sized for that and an overflow drops the frame.
        s->max_out = std::min(ug_hip_jpeg_encoder_max_size(s->enc), (size_t) desc.width * desc.height * 3 + 4096);
        bool ok = ug_hip_malloc(&s->dev_in, s->in_len + MAX_PADDING) == UG_HIP_SUCCESS &&
                  ug_hip_malloc(&s->dev_out, s->max_out) == UG_HIP_SUCCESS;
        if (ok && s->wire != s->target) { // staging buffer for the device-side decoder_t pass (wire -> target)
                ok = ug_hip_malloc(&s->dev_target, (size_t) vc_get_linesize(desc.width, ug_codec_from_pixfmt(s->target)) * (desc.height + 1) + MAX_PADDING) == UG_HIP_SUCCESS;
        }
        if (ok && s->target != s->enc_in) { // and for target -> encoder input (RGBA -> RGB; RGB-family -> UYVY on subsampling=422/420)
                ok = ug_hip_malloc(&s->dev_uyvy, (size_t) vc_get_linesize(desc.width, s->enc_in == UG_PF_RGB ? RGB : UYVY) * desc.height + MAX_PADDING) == UG_HIP_SUCCESS;
        }
        if (!ok) {
                MSG(ERROR, "Could not allocate device buffers: %s\n", ug_hip_last_error_string());
                return false;
        }
        struct video_desc compressed_desc = desc;
        compressed_desc.color_spec = JPEG;
        compressed_desc.tile_count = 1;
        s->pool->reconfigure(compressed_desc, s->max_out);
        return true;
}

std::shared_ptr<video_frame> jpeg_mi355x_compress_tile(void *state, std::shared_ptr<video_frame> tx)
{
        if (!tx) {
                return {};
        }
        auto *s = static_cast<state_video_compress_jpeg_mi355x *>(state);
        if (ug_hip_set_device(s->device) != UG_HIP_SUCCESS) {
                return {};
        }
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
        // device-resident frame (types.h:295-298; gpujpeg.cpp:617-622): used in place, no upload -- when it lives on this state's GPU
        const bool on_dev = tx->mem_location == CUDA_MEM || ug_hip_pointer_is_device(tx->tiles[0].data);
        const void *enc_src = s->dev_in;
        if (ug_hip_pointer_device(tx->tiles[0].data) == s->device && ((uintptr_t) tx->tiles[0].data & 15) == 0) {
                enc_src = tx->tiles[0].data;
        } else if (ug_hip_upload_ordered(s->device, s->dev_in, tx->tiles[0].data, s->in_len, on_dev ? UG_HIP_MEMCPY_DEVICE_TO_DEVICE : UG_HIP_MEMCPY_HOST_TO_DEVICE,
                                         s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "upload failed: %s\n", ug_hip_last_error_string());
                return {};
        }
        if (s->wire != s->target) {
                if (wire_to_target(s, enc_src, s->dev_target, w, h) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "device conversion %s -> %s failed: %s\n", get_codec_name(tx->color_spec), get_codec_name(ug_codec_from_pixfmt(s->target)), ug_hip_last_error_string());
                        return {};
                }
                enc_src = s->dev_target;
        }
        if (s->target != s->enc_in) {
                if (ug_hip_pixfmt_convert(s->target, s->enc_in, enc_src, s->dev_uyvy, w, h, 0, 0, 0, 8, 16, s->stream) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "device conversion to the encoder input format failed: %s\n", ug_hip_last_error_string());
                        return {};
                }
                enc_src = s->dev_uyvy;
        }
        size_t len = 0;
        if (ug_hip_jpeg_encoder_encode(s->enc, s->enc_in, enc_src, 0, s->dev_out, s->max_out, &len, s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Encoding failed: %s\n", ug_hip_last_error_string());
                return {};
        }
        std::shared_ptr<video_frame> out = mi355x::get_frame_keeping_pool(s->pool);
        if (ug_hip_download_ordered(s->device, out->tiles[0].data, s->dev_out, len, s->stream) != UG_HIP_SUCCESS ||
            ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "D2H copy failed: %s\n", ug_hip_last_error_string());
                return {};
        }
        out->tiles[0].data_len = (unsigned int) len;
        return out;
}

/// `frames` queued frames of one geometry in one go: per frame the upload (+ the device-side decoder_t passes) into its slice of the batch
/// buffers, then ONE ug_hip_jpeg_encoder_encode_batch (fused front end grid.z = frame, entropy coder + compaction grid.y = frame, one
/// synchronisation) and the downloads.  Streams are byte-identical to the one-frame path's.  Anything unusual -> the one-frame path.
std::vector<std::shared_ptr<video_frame>> jpeg_mi355x_compress_batch(void *state, std::vector<std::shared_ptr<video_frame>> in)
{
        auto *s = static_cast<state_video_compress_jpeg_mi355x *>(state);
        const int n = (int) in.size();
        std::vector<std::shared_ptr<video_frame>> out(in.size());
        auto one_by_one = [&] {
                for (size_t i = 0; i < in.size(); i++) out[i] = jpeg_mi355x_compress_tile(state, std::move(in[i]));
                return out;
        };
        if (n < 2 || n > s->batch_slices || ug_hip_set_device(s->device) != UG_HIP_SUCCESS) return one_by_one();
        if (!video_desc_eq_excl_param(video_desc_from_frame(in[0].get()), s->saved_desc, PARAM_TILE_COUNT)) {
                out[0] = jpeg_mi355x_compress_tile(state, in[0]); // (re)configures; the rest of this round follows one by one
                for (size_t i = 1; i < in.size(); i++) out[i] = jpeg_mi355x_compress_tile(state, std::move(in[i]));
                return out;
        }
        const int w = (int) in[0]->tiles[0].width, h = (int) in[0]->tiles[0].height;
        if (n > s->batch_cap) {
                for (void **p : { &s->b_in, &s->b_target, &s->b_enc, &s->b_out }) {
                        if (*p) { ug_hip_free(*p); *p = nullptr; }
                }
                auto round16 = [](size_t v) { return (v + 15) / 16 * 16; };
                s->b_in_stride = round16(s->in_len + MAX_PADDING);
                s->b_target_stride = round16((size_t) vc_get_linesize(w, ug_codec_from_pixfmt(s->target == UG_PF_I420 ? UG_PF_UYVY : s->target)) * h + MAX_PADDING);
                s->b_enc_stride = round16((size_t) vc_get_linesize(w, s->enc_in == UG_PF_RGB ? RGB : UYVY) * h + MAX_PADDING);
                s->b_out_stride = round16(s->max_out);
                const size_t slices = (size_t) s->batch_slices;
                bool ok = ug_hip_malloc(&s->b_in, s->b_in_stride * slices) == UG_HIP_SUCCESS && ug_hip_malloc(&s->b_out, s->b_out_stride * slices) == UG_HIP_SUCCESS;
                if (ok && s->wire != s->target) ok = ug_hip_malloc(&s->b_target, s->b_target_stride * slices) == UG_HIP_SUCCESS;
                if (ok && s->target != s->enc_in) ok = ug_hip_malloc(&s->b_enc, s->b_enc_stride * slices) == UG_HIP_SUCCESS;
                if (!ok) {
                        MSG(WARNING, "no device memory for the batch buffers (%s): frames are encoded one by one\n", ug_hip_last_error_string());
                        return one_by_one();
                }
                s->batch_cap = s->batch_slices;
        }
        // every frame into its slice: upload (or device-to-device), wire -> target -> encoder input with the pixfmt_conv.c arithmetic
        const char *enc_base = (const char *) s->b_in;
        size_t enc_stride = s->b_in_stride;
        // (all the uploads of the batch first, then the conversions: the device's upload lane is shared by every worker -- ADVICE r3)
        for (int f = 0; f < n; f++) {
                const bool on_dev = in[f]->mem_location == CUDA_MEM || ug_hip_pointer_is_device(in[f]->tiles[0].data);
                char *slice = (char *) s->b_in + f * s->b_in_stride;
                if (ug_hip_upload_ordered(s->device, slice, in[f]->tiles[0].data, s->in_len, on_dev ? UG_HIP_MEMCPY_DEVICE_TO_DEVICE : UG_HIP_MEMCPY_HOST_TO_DEVICE, s->stream) != UG_HIP_SUCCESS) {
                        MSG(ERROR, "upload failed: %s\n", ug_hip_last_error_string());
                        return out;
                }
        }
        for (int f = 0; f < n; f++) {
                const void *cur = (char *) s->b_in + f * s->b_in_stride;
                if (s->wire != s->target) {
                        void *t = (char *) s->b_target + f * s->b_target_stride;
                        if (wire_to_target(s, cur, t, w, h) != UG_HIP_SUCCESS) return out;
                        cur = t;
                }
                if (s->target != s->enc_in) {
                        void *t = (char *) s->b_enc + f * s->b_enc_stride;
                        if (ug_hip_pixfmt_convert(s->target, s->enc_in, cur, t, w, h, 0, 0, 0, 8, 16, s->stream) != UG_HIP_SUCCESS) return out;
                }
        }
        if (s->target != s->enc_in) { enc_base = (const char *) s->b_enc; enc_stride = s->b_enc_stride; }
        else if (s->wire != s->target) { enc_base = (const char *) s->b_target; enc_stride = s->b_target_stride; }
        size_t lens[16] = {};
        if (ug_hip_jpeg_encoder_encode_batch(s->enc, s->enc_in, n, enc_base, 0, enc_stride, s->b_out, s->b_out_stride, s->max_out, lens, s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "Encoding failed: %s\n", ug_hip_last_error_string());
                return out;
        }
        for (int f = 0; f < n; f++) {
                if (lens[f] > s->max_out) { // only this frame is lost, as on the one-frame path
                        MSG(ERROR, "Encoding failed: stream of %zu bytes does not fit the output buffer\n", lens[f]);
                        continue;
                }
                out[f] = mi355x::get_frame_keeping_pool(s->pool);
                if (ug_hip_download_ordered(s->device, out[f]->tiles[0].data, (char *) s->b_out + f * s->b_out_stride, lens[f], s->stream) != UG_HIP_SUCCESS) {
                        out[f].reset();
                        continue;
                }
                out[f]->tiles[0].data_len = (unsigned int) lens[f];
        }
        if (ug_hip_stream_sync(s->stream) != UG_HIP_SUCCESS) {
                MSG(ERROR, "D2H copy failed: %s\n", ug_hip_last_error_string());
                for (auto &o : out) o.reset();
        }
        return out;
}

void jpeg_mi355x_compress_done(void *state)
{
        auto *s = static_cast<state_video_compress_jpeg_mi355x *>(state);
        ug_hip_set_device(s->device);
        cleanup(s);
        if (s->stream) ug_hip_stream_destroy(s->stream);
        delete s;
}

compress_module_info get_jpeg_mi355x_module_info()
{
        compress_module_info module_info;
        module_info.name = "jpeg";
        module_info.opts.emplace_back(module_option{ "Quality", "Quality 1-100", "75", "quality", ":q=", false });
        module_info.opts.emplace_back(module_option{ "Restart interval", "MCUs per restart interval", "2", "restart_interval", ":restart=", false });
        module_info.opts.emplace_back(module_option{ "Subsampling", "JPEG subsampling (444, 422 or 420; default: that of the input)", "", "subsampling", ":subsampling=", false });
        codec codec_info;
        codec_info.name = "JPEG";
        codec_info.priority = 300;
        codec_info.encoders.emplace_back(encoder{ "default", "" });
        module_info.codecs.emplace_back(std::move(codec_info));
        return module_info;
}

/// module-level init: consumes dev=<list>, creates one worker (thread + per-tile encoder states) per listed device
void *jpeg_mi355x_module_init(struct module *parent, const char *cfg)
{
        return mi355x::sharded_init(parent, cfg, jpeg_mi355x_compress_init, jpeg_mi355x_compress_tile, jpeg_mi355x_compress_done, ug_hip_set_device,
                                    jpeg_mi355x_compress_batch, ug_hip_bind_thread_to_device, ug_hip_device_numa_node);
}

const struct video_compress_info jpeg_mi355x_info = {
        jpeg_mi355x_module_init,
        mi355x::sharded_done,
        NULL,
        NULL,
        mi355x::sharded_push, // asynchronous frame API: frames are dealt to one worker per listed GPU and popped in order
        mi355x::sharded_pop,
        NULL,
        NULL,
        get_jpeg_mi355x_module_info,
};

// Always reachable as "jpeg_mi355x".  The short name "jpeg" is the hidden alias the reference gives its GPUJPEG module
// (gpujpeg.cpp:791-792, REGISTER_HIDDEN_MODULE(jpeg, ...)): in a build that contains that module (config.h: HAVE_GPUJPEG,
// configure.ac:2673) the two registrations would collide in lib_common's registry and "-c jpeg" would resolve to whichever
// constructor ran first, so the alias is taken only when GPUJPEG is absent -- then "-c jpeg" is this module, as a drop-in.
REGISTER_MODULE(jpeg_mi355x, &jpeg_mi355x_info, LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION);
// The same goes for the module's own name: without GPUJPEG in the build, "-c GPUJPEG[:<options>]" -- what users and the reference's own
// unit test (test/gpujpeg_test.cpp:57-62) ask for -- is this module.
#ifndef HAVE_GPUJPEG
REGISTER_HIDDEN_MODULE(jpeg, &jpeg_mi355x_info, LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION);
REGISTER_HIDDEN_MODULE(gpujpeg, &jpeg_mi355x_info, LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION);
#endif

} // end of anonymous namespace
