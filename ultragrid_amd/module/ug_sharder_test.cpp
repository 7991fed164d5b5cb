/**
 * @file ug_sharder_test.cpp
 * CPU-only test of mi355x::frame_sharder: a fake tile encoder with random delays and injected failures
 * stands in for the GPU.  Checks: in-order delivery with the sequence numbers of the pushed frames, failed frames skipped,
 * tiled frames split / merged, every worker used, the poison pill ends pop() after all frames came out, metadata kept.
 * Links the reference's own video_frame / vf_split objects (no HIP).  usage: ug_sharder_test [workers] [frames] [batch]
 * batch > 1: workers queue up to that many frames and a fake batch encoder takes what has queued up (single-tile frames of one
 * geometry); every third single-tile frame OBJECT is pushed twice in a row (two sequence numbers, one object), as a file source would.
 * A round that holds a tiled frame is encoded one at a time, so in batch mode tiled frames are rare (every 40th, not every 5th) and every
 * worker's first call takes 30 ms: the producer fills the queues meanwhile and the first round a worker drains after it is a pure one --
 * that a batch forms does not depend on scheduling luck.
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <thread>

#include "mi355x_frame_sharder.h"
#include "video_codec.h"

int main(int argc, char **argv)
{
        const int workers = argc > 1 ? atoi(argv[1]) : 4;
        const unsigned frames = argc > 2 ? atoi(argv[2]) : 200;
        const unsigned batch = argc > 3 ? atoi(argv[3]) : 1;
        const unsigned tiled_every = batch > 1 ? 40 : 5;
        std::atomic<unsigned> batch_calls{0}, batched_frames{0};
        std::vector<int> devices;
        for (int i = 0; i < workers; i++) devices.push_back(i);
        std::atomic<unsigned> calls{0};
        std::mutex used_lock;
        std::set<int> used;
        // worker start hook (the product binds the thread to the GPU's NUMA node there): must have run on the SAME thread, before its first encode
        std::mutex started_lock;
        std::map<int, std::thread::id> started;
        std::atomic<unsigned> hook_errors{0};
        mi355x::frame_sharder sh(devices, [&](int device) -> mi355x::tile_encoder_t {
                auto rng = std::make_shared<std::mt19937>(1234 + device);
                auto first = std::make_shared<std::atomic<bool>>(true);
                return [&, rng, first, device](int dev, unsigned tile, std::shared_ptr<video_frame> in) -> std::shared_ptr<video_frame> {
                        if (dev != device) abort();
                        {
                                std::lock_guard<std::mutex> lk(started_lock);
                                auto it = started.find(dev);
                                if (it == started.end() || it->second != std::this_thread::get_id()) hook_errors++;
                        }
                        calls++;
                        { std::lock_guard<std::mutex> lk(used_lock); used.insert(dev); }
                        if (batch > 1 && first->exchange(false)) std::this_thread::sleep_for(std::chrono::milliseconds(30)); // lets the queues fill
                        std::this_thread::sleep_for(std::chrono::microseconds((*rng)() % 3000));
                        uint32_t tag;
                        memcpy(&tag, in->tiles[0].data, 4);
                        if (tag % 17 == 5) return {}; // injected encoder failure: this frame must be skipped
                        struct video_desc d = video_desc_from_frame(in.get());
                        d.color_spec = DXT5;
                        std::shared_ptr<video_frame> out(vf_alloc_desc_data(d), vf_free);
                        memcpy(out->tiles[0].data, &tag, 4);
                        out->tiles[0].data[4] = (char) tile;
                        out->tiles[0].data[5] = (char) dev;
                        return out;
                };
        }, batch, [&](int, mi355x::tile_encoder_t one) -> mi355x::batch_encoder_t {
                return [&, one](int dev, std::vector<std::shared_ptr<video_frame>> in) {
                        batch_calls++;
                        batched_frames += (unsigned) in.size();
                        if (in.size() < 2 || in.size() > batch) abort();
                        std::vector<std::shared_ptr<video_frame>> out;
                        for (auto &f : in) {
                                if (f->tile_count != 1) abort();
                                out.push_back(one(dev, 0, std::move(f)));
                        }
                        return out;
                };
        }, [&](int device) {
                std::lock_guard<std::mutex> lk(started_lock);
                if (!started.emplace(device, std::this_thread::get_id()).second) hook_errors++; // once per worker
        });
        std::vector<std::shared_ptr<video_frame>> got;
        std::thread consumer([&] {
                while (auto f = sh.pop()) got.push_back(f);
        });
        unsigned expected = 0;
        for (unsigned i = 0; i < frames; i++) {
                struct video_desc d{};
                d.width = 64; d.height = 16; d.color_spec = UYVY; d.fps = 25; d.interlacing = PROGRESSIVE;
                d.tile_count = i % tiled_every == 0 ? 4 : 1;
                std::shared_ptr<video_frame> f(vf_alloc_desc_data(d), vf_free);
                for (unsigned t = 0; t < d.tile_count; t++) memcpy(f->tiles[t].data, &i, 4);
                f->compress_start = 1000 + i;
                if (i % 17 != 5) expected++;
                sh.push(f);
                if (batch > 1 && d.tile_count == 1 && i % 3 == 1 && i + 1 < frames && (i + 1) % tiled_every != 0) { // the same object again, as frame i + 1
                        if (i % 17 != 5) expected++; // same payload (tag i): succeeds or fails like the first push
                        i++;
                        sh.push(f);
                }
        }
        sh.push({});
        consumer.join();
        int rc = 0;
        if (got.size() != expected) { fprintf(stderr, "got %zu frames, expected %u\n", got.size(), expected); rc = 1; }
        uint32_t last = 0;
        bool first = true;
        for (auto &f : got) {
                uint32_t tag;
                memcpy(&tag, f->tiles[0].data, 4);
                if (tag != f->seq && !(batch > 1 && tag + 1 == f->seq)) { fprintf(stderr, "payload %u under seq %u\n", tag, f->seq); rc = 1; }
                if (!first && f->seq <= last) { fprintf(stderr, "out of order: %u after %u\n", f->seq, last); rc = 1; }
                if (tag % 17 == 5) { fprintf(stderr, "failed frame %u was delivered\n", f->seq); rc = 1; }
                if (f->tile_count != (f->seq % tiled_every == 0 ? 4u : 1u)) { fprintf(stderr, "tile count of %u\n", f->seq); rc = 1; }
                for (unsigned t = 0; t < f->tile_count; t++) {
                        if ((unsigned char) f->tiles[t].data[4] != t) { fprintf(stderr, "tile order in %u\n", f->seq); rc = 1; }
                }
                if (f->compress_start != (time_ns_t) (1000 + tag)) { fprintf(stderr, "metadata of %u lost\n", f->seq); rc = 1; }
                if (f->compress_end <= f->compress_start) { fprintf(stderr, "compress_end of %u not set\n", f->seq); rc = 1; }
                last = f->seq;
                first = false;
        }
        if ((int) used.size() != workers && frames >= 50) { fprintf(stderr, "only %zu of %d workers used\n", used.size(), workers); rc = 1; }
        if (hook_errors != 0 || (int) started.size() != workers) { fprintf(stderr, "worker start hook: %u errors, ran on %zu of %d workers\n", hook_errors.load(), started.size(), workers); rc = 1; }
        if (batch > 1 && batch_calls == 0) { fprintf(stderr, "no batch ever formed\n"); rc = 1; }
        printf("%s frames=%zu tile_encodes=%u workers_used=%zu batch_calls=%u batched_frames=%u\n", rc ? "FAIL" : "OK", got.size(), calls.load(), used.size(),
               batch_calls.load(), batched_frames.load());
        return rc;
}
