/**
 * @file ug_fake_compress.cpp
 * TEST-ONLY video_compress module "fake": the structure of vcompress_dxt_mi355x.cpp / vcompress_jpeg_mi355x.cpp -- per-tile encoder state with a
 * saved video_desc that reconfigures lazily on a format change (cuda_dxt.cpp:196-204), a batch entry, an output video_frame_pool, all of it behind
 * mi355x::sharded_init (workers=, batch=, dev=) and the asynchronous frame API -- with the GPU replaced by a hash, so that the run-time conventions
 * of the boundary (format change in flight, CHANGE_COMPRESS, compress_done with frames queued) can be driven through the reference's framework on a
 * CPU box, under ThreadSanitizer and AddressSanitizer (tests/test_runtime_conventions.py).  Never part of the product: it is linked into
 * oracle/_ref/ug_runtime_harness_fake* only.
 *
 * -c fake[:tag=<n>][:delay_us=<max>][:fail_every=<k>] + the sharder's options.  A "compressed" frame is 80 bytes:
 *   u32 magic 'FAKE', tag, cfg_w, cfg_h, cfg_codec, cfg_interlacing   <- what the state was CONFIGURED for when it encoded the frame
 *   u32 w, h, codec, interlacing                                      <- the frame's own desc
 *   u64 fnv1a(tile bytes), u32 batch_n, device, in_len, state_serial, pad[2]
 * so a frame of one format encoded under the configuration of another is visible in the output (cfg_* != own desc).
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "debug.h"
#include "host.h"
#include "lib_common.h"
#include "types.h"
#include "utils/video_frame_pool.h"
#include "video_codec.h"
#include "video_compress.h"
#include "video_frame.h"

#include "mi355x_frame_sharder.h"

#define MOD_NAME "[fake] "

namespace {

constexpr size_t OUT_LEN = 80;
std::atomic<uint32_t> g_state_serial{0};
std::atomic<int> g_live_states{0}; // printed at exit: every init must have met its done

struct state_fake {
        struct video_desc saved_desc{};
        struct video_desc cfg_desc{};
        uint32_t tag = 0;
        unsigned delay_us = 0, fail_every = 0;
        int device = 0;
        int batch_slices = 16;
        uint32_t serial = 0;
        uint64_t encoded = 0;
        std::mt19937 rng{1};
        std::vector<char> dev_in; // stands for the device buffers: sized by configure_with(), so a stale size is an out-of-bounds write ASan sees
        std::shared_ptr<video_frame_pool> pool = std::make_shared<video_frame_pool>(0, default_data_allocator()); ///< shared with the frames it gives out (mi355x::get_frame_keeping_pool)
};

void *fake_init(struct module *, const char *fmt)
{
        auto *s = new state_fake();
        std::string cfg = fmt ? fmt : "";
        size_t pos = 0;
        while (pos <= cfg.size() && !cfg.empty()) {
                const size_t end = cfg.find(':', pos);
                const std::string tok = cfg.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                if (tok.rfind("tag=", 0) == 0) s->tag = (uint32_t) atoi(tok.c_str() + 4);
                else if (tok.rfind("delay_us=", 0) == 0) s->delay_us = (unsigned) atoi(tok.c_str() + 9);
                else if (tok.rfind("fail_every=", 0) == 0) s->fail_every = (unsigned) atoi(tok.c_str() + 11);
                else if (tok.rfind("dev=", 0) == 0) s->device = atoi(tok.c_str() + 4);
                else if (tok.rfind("batch_slices=", 0) == 0) s->batch_slices = atoi(tok.c_str() + 13);
                else if (tok == "help") { printf("fake compress: test only\n"); delete s; return INIT_NOERR; }
                else if (!tok.empty()) { MSG(ERROR, "unknown option: %s\n", tok.c_str()); delete s; return nullptr; }
                if (end == std::string::npos) break;
                pos = end + 1;
        }
        s->serial = g_state_serial++;
        s->rng.seed(s->serial * 7919u + 13u);
        g_live_states++;
        return s;
}

bool configure_with(state_fake *s, struct video_desc desc)
{
        if (desc.width % 4 != 0 || desc.height % 4 != 0) {
                MSG(ERROR, "Frame size %ux%u is not a multiple of the 4x4 block\n", desc.width, desc.height);
                return false;
        }
        s->cfg_desc = desc;
        s->dev_in.assign((size_t) vc_get_linesize(desc.width, desc.color_spec) * desc.height, 0);
        struct video_desc out = desc;
        out.color_spec = DXT1;
        out.tile_count = 1;
        s->pool->reconfigure(out, OUT_LEN);
        return true;
}

uint64_t fnv1a(const char *p, size_t n)
{
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) { h ^= (unsigned char) p[i]; h *= 1099511628211ull; }
        return h;
}

std::shared_ptr<video_frame> encode_configured(state_fake *s, const std::shared_ptr<video_frame> &tx, uint32_t batch_n)
{
        if (s->delay_us) std::this_thread::sleep_for(std::chrono::microseconds(s->rng() % (s->delay_us + 1)));
        s->encoded++;
        if (s->fail_every && s->encoded % s->fail_every == 0) return {};
        memcpy(s->dev_in.data(), tx->tiles[0].data, tx->tiles[0].data_len); // the "upload": into the buffer the CONFIGURED geometry sized
        std::shared_ptr<video_frame> out = mi355x::get_frame_keeping_pool(s->pool);
        const struct video_desc d = video_desc_from_frame(tx.get());
        uint32_t rec[20] = { 0x454b4146u, s->tag, s->cfg_desc.width, s->cfg_desc.height, (uint32_t) s->cfg_desc.color_spec, (uint32_t) s->cfg_desc.interlacing,
                             d.width, d.height, (uint32_t) d.color_spec, (uint32_t) d.interlacing };
        const uint64_t h = fnv1a(s->dev_in.data(), tx->tiles[0].data_len);
        memcpy(&rec[10], &h, 8);
        rec[12] = batch_n; rec[13] = (uint32_t) s->device; rec[14] = tx->tiles[0].data_len; rec[15] = s->serial;
        memcpy(out->tiles[0].data, rec, OUT_LEN);
        out->tiles[0].data_len = OUT_LEN;
        return out;
}

std::shared_ptr<video_frame> fake_compress_tile(void *state, std::shared_ptr<video_frame> tx)
{
        if (!tx) return {};
        auto *s = static_cast<state_fake *>(state);
        if (!video_desc_eq_excl_param(video_desc_from_frame(tx.get()), s->saved_desc, PARAM_TILE_COUNT)) {
                if (configure_with(s, video_desc_from_frame(tx.get()))) {
                        s->saved_desc = video_desc_from_frame(tx.get());
                } else {
                        s->saved_desc = {};
                        return {};
                }
        }
        return encode_configured(s, tx, 1);
}

std::vector<std::shared_ptr<video_frame>> fake_compress_batch(void *state, std::vector<std::shared_ptr<video_frame>> in)
{
        auto *s = static_cast<state_fake *>(state);
        std::vector<std::shared_ptr<video_frame>> out(in.size());
        if (in.size() < 2 || (int) in.size() > s->batch_slices || !video_desc_eq_excl_param(video_desc_from_frame(in[0].get()), s->saved_desc, PARAM_TILE_COUNT)) {
                for (size_t i = 0; i < in.size(); i++) out[i] = fake_compress_tile(state, std::move(in[i]));
                return out;
        }
        for (size_t i = 0; i < in.size(); i++) out[i] = encode_configured(s, in[i], (uint32_t) in.size());
        return out;
}

void fake_done(void *state)
{
        g_live_states--;
        delete static_cast<state_fake *>(state);
}

int fake_set_device(int d) { return d >= 0 && d < 64 ? 0 : -1; }

void *fake_module_init(struct module *parent, const char *cfg)
{
        return mi355x::sharded_init(parent, cfg, fake_init, fake_compress_tile, fake_done, fake_set_device, fake_compress_batch);
}

compress_module_info get_fake_module_info()
{
        compress_module_info mi;
        mi.name = "fake";
        return mi;
}

const struct video_compress_info fake_info = {
        fake_module_init, mi355x::sharded_done, NULL, NULL, mi355x::sharded_push, mi355x::sharded_pop, NULL, NULL, get_fake_module_info,
};
REGISTER_MODULE(fake, &fake_info, LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION);

struct report_at_exit {
        ~report_at_exit() { printf("FAKE live_states=%d created=%u\n", g_live_states.load(), g_state_serial.load()); }
} g_report;

} // namespace
