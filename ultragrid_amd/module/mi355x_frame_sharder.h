/**
 * @file mi355x_frame_sharder.h
 * Frame-level multi-GPU dispatch for the MI355X compress modules.
 *
 * Frames are independent, so N GPUs are used by giving each incoming frame to the first idle worker (one worker thread per
 * listed device), numbering the frames, and delivering the results in order -- the scheme of the reference's GPUJPEG module
 * (src/video_compress/gpujpeg.cpp:446-466 workers per device, :643-676 dispatch, :688-722 reorder), re-implemented here on
 * std::thread / std::mutex.  There is no inter-GPU traffic and no collective.  The module exposes it through the asynchronous
 * frame API (compress_frame_async_push_func / _pop_func, video_compress.h:146-177): push() runs on the capture thread,
 * pop() on the framework's consumer thread (video_compress.cpp:285-292,583-599).
 *
 * A worker encodes whole frames: the tiles of a tiled frame are encoded one after another on the worker's GPU with per-tile
 * encoder states that are created on first use.
 */
#ifndef MI355X_FRAME_SHARDER_H
#define MI355X_FRAME_SHARDER_H

#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <string>
#include <strings.h>

#include "debug.h"
#include "host.h"
#include "tv.h"
#include "types.h"
#include "utils/vf_split.h"
#include "video_frame.h"

namespace mi355x {

/// encodes one single-tile frame on `device` with the (lazily created) encoder state number `tile`; returns {} on error
using tile_encoder_t = std::function<std::shared_ptr<video_frame>(int device, unsigned tile, std::shared_ptr<video_frame> in)>;

class frame_sharder {
public:
        /// one worker per entry of `devices` (a device listed twice gets two workers, i.e. two frames in flight on it)
        frame_sharder(const std::vector<int> &devices, std::function<tile_encoder_t(int device)> make_encoder)
        {
                for (int d : devices) {
                        auto *w = new worker();
                        w->device = d;
                        w->encode_tile = make_encoder(d);
                        m_workers.emplace_back(w);
                }
                for (auto &w : m_workers) {
                        w->th = std::thread(&frame_sharder::run, this, w.get());
                }
        }

        ~frame_sharder()
        {
                for (auto &w : m_workers) {
                        {
                                std::lock_guard<std::mutex> lk(w->m);
                                w->quit = true;
                        }
                        w->cv.notify_all();
                }
                for (auto &w : m_workers) {
                        if (w->th.joinable()) w->th.join();
                }
        }

        size_t worker_count() const { return m_workers.size(); }

        /// capture thread.  Empty frame = poison pill, forwarded to every worker (gpujpeg.cpp:653-658).
        void push(std::shared_ptr<video_frame> in)
        {
                if (!in) {
                        for (auto &w : m_workers) give(w.get(), {}, true);
                        return;
                }
                in->seq = m_in_seq++;
                size_t index = 0;
                {       // wait for / select a worker that is not occupied (gpujpeg.cpp:660-673)
                        std::unique_lock<std::mutex> lk(m_occupancy_lock);
                        m_worker_finished.wait(lk, [&] {
                                for (index = 0; index < m_workers.size(); index++) {
                                        if (!m_workers[index]->occupied) return true;
                                }
                                return false;
                        });
                        m_workers[index]->occupied = true;
                }
                give(m_workers[index].get(), std::move(in), false);
        }

        /// consumer thread: frames in the order they were pushed; frames that failed to encode are skipped; {} after the pill
        /// has passed through every worker (gpujpeg.cpp:688-722)
        std::shared_ptr<video_frame> pop()
        {
                for (;;) {
                        auto it = m_out_frames.find(m_out_seq);
                        if (it != m_out_frames.end()) {
                                result r = std::move(it->second);
                                m_out_frames.erase(it);
                                m_out_seq++;
                                if (r.frame) return r.frame;
                                continue; // encoding error: skip this sequence number
                        }
                        result r;
                        {
                                std::unique_lock<std::mutex> lk(m_out_lock);
                                m_out_cv.wait(lk, [&] { return !m_out_queue.empty(); });
                                r = std::move(m_out_queue.front());
                                m_out_queue.pop_front();
                        }
                        if (r.pill) {
                                if (++m_ended_count == m_workers.size()) {
                                        m_ended_count = 0; // the module may be fed again after a pill (not done by UltraGrid, cheap to allow)
                                        return {};
                                }
                                continue;
                        }
                        m_out_frames.emplace(r.seq, std::move(r));
                }
        }

private:
        struct result {
                uint32_t seq = 0;
                bool pill = false;
                std::shared_ptr<video_frame> frame;
        };
        struct worker {
                int device = 0;
                tile_encoder_t encode_tile;
                std::thread th;
                std::mutex m;
                std::condition_variable cv;
                std::deque<std::pair<std::shared_ptr<video_frame>, bool>> q; // (frame, is_pill)
                bool occupied = false, quit = false;
        };

        void give(worker *w, std::shared_ptr<video_frame> f, bool pill)
        {
                {
                        std::lock_guard<std::mutex> lk(w->m);
                        w->q.emplace_back(std::move(f), pill);
                }
                w->cv.notify_one();
        }

        void deliver(result r)
        {
                {
                        std::lock_guard<std::mutex> lk(m_out_lock);
                        m_out_queue.emplace_back(std::move(r));
                }
                m_out_cv.notify_one();
        }

        std::shared_ptr<video_frame> encode_frame(worker *w, std::shared_ptr<video_frame> in)
        {
                if (in->tile_count == 1) {
                        return w->encode_tile(w->device, 0, in);
                }
                std::vector<std::shared_ptr<video_frame>> tiles = vf_separate_tiles(in);
                in.reset();
                std::vector<std::shared_ptr<video_frame>> out(tiles.size());
                for (unsigned t = 0; t < tiles.size(); t++) {
                        out[t] = w->encode_tile(w->device, t, std::move(tiles[t]));
                        if (!out[t]) return {};
                }
                return vf_merge_tiles(out);
        }

        void run(worker *w)
        {
                for (;;) {
                        std::pair<std::shared_ptr<video_frame>, bool> item;
                        {
                                std::unique_lock<std::mutex> lk(w->m);
                                w->cv.wait(lk, [&] { return w->quit || !w->q.empty(); });
                                if (w->q.empty()) return; // quit
                                item = std::move(w->q.front());
                                w->q.pop_front();
                        }
                        if (item.second) {
                                result r;
                                r.pill = true;
                                deliver(std::move(r));
                                continue;
                        }
                        result r;
                        r.seq = item.first->seq;
                        char metadata[VF_METADATA_SIZE];
                        vf_store_metadata(item.first.get(), metadata);
                        r.frame = encode_frame(w, std::move(item.first));
                        if (r.frame) {
                                vf_restore_metadata(r.frame.get(), metadata); // gpujpeg.cpp:188-193
                                r.frame->seq = r.seq;
                                r.frame->compress_end = get_time_in_ns(); // the async frame API leaves this to the module (gpujpeg.cpp:188-195)
                        }
                        deliver(std::move(r));
                        {
                                std::lock_guard<std::mutex> lk(m_occupancy_lock);
                                w->occupied = false;
                        }
                        m_worker_finished.notify_one();
                }
        }

        std::vector<std::unique_ptr<worker>> m_workers;
        std::mutex m_occupancy_lock;
        std::condition_variable m_worker_finished;
        uint32_t m_in_seq = 0;
        // consumer side (pop() is only ever called from one thread)
        std::mutex m_out_lock;
        std::condition_variable m_out_cv;
        std::deque<result> m_out_queue;
        std::map<uint32_t, result> m_out_frames;
        uint32_t m_out_seq = 0;
        size_t m_ended_count = 0;
};

/// "dev=<n>[,<n>...]" -> device list (default {0})
inline std::vector<int> parse_device_list(const char *s)
{
        std::vector<int> devs;
        for (const char *p = s; p && *p;) {
                devs.push_back(atoi(p));
                p = strchr(p, ',');
                if (!p) break;
                p++;
        }
        if (devs.empty()) devs.push_back(0);
        return devs;
}

// ------------------------------------------------------------------------------------------------------------------------
// Glue that turns a tile encoder (init / compress_tile / done, one state per tile and device, as the tile API of
// video_compress.h:115-145 shapes it) into a module with the asynchronous frame API on top of frame_sharder.
// ------------------------------------------------------------------------------------------------------------------------
using tile_init_t = void *(*)(struct module *parent, const char *cfg);
using tile_compress_t = std::shared_ptr<video_frame> (*)(void *state, std::shared_ptr<video_frame> in);
using tile_done_t = void (*)(void *state);

struct sharded_module {
        std::unique_ptr<frame_sharder> sharder;
};

/// owns the per-tile encoder states of one worker
struct tile_state_set {
        tile_done_t done;
        std::vector<void *> states;
        explicit tile_state_set(tile_done_t d) : done(d) {}
        ~tile_state_set() { for (void *s : states) done(s); }
};

/// cfg = the module's option string; "dev=<n>[,<n>...]" and "workers=<per device>" are consumed here, everything else goes to tile_init unchanged
/// (with ":dev=<n>" of the worker appended).  Returns what tile_init returns for a bad / help configuration.
inline void *sharded_init(struct module *parent, const char *cfg, tile_init_t tile_init, tile_compress_t tile_compress, tile_done_t tile_done,
                          int (*set_device)(int))
{
        std::string rest, all = cfg ? cfg : "";
        std::vector<int> devices{ 0 };
        // Two workers per device by default: upload, kernels and download of consecutive frames overlap on one GPU (measured through
        // the reference framework over 4 000 4K frames: DXT5 1 979 -> 3 079 fps, JPEG 2 126 -> 2 393 fps, 8K v210 346 -> 445 fps); workers=1 gives the reference's one-per-device.
        int workers_per_device = 2;
        size_t pos = 0;
        while (pos <= all.size() && !all.empty()) {
                const size_t end = all.find(':', pos);
                const std::string tok = all.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                if (tok == "help") { // usage text, INIT_NOERR; must work without a GPU
                        return tile_init(parent, "help");
                }
                if (strncasecmp(tok.c_str(), "dev=", 4) == 0) {
                        devices = parse_device_list(tok.c_str() + 4);
                } else if (strncasecmp(tok.c_str(), "workers=", 8) == 0) {
                        workers_per_device = atoi(tok.c_str() + 8);
                } else if (!tok.empty()) {
                        rest += (rest.empty() ? "" : ":") + tok;
                }
                if (end == std::string::npos) break;
                pos = end + 1;
        }
        if (workers_per_device < 1 || workers_per_device > 8) {
                log_msg(LOG_LEVEL_ERROR, "[MI355X] workers=<n> must be 1..8\n");
                return nullptr;
        }
        {
                std::vector<int> expanded;
                for (int k = 0; k < workers_per_device; k++) expanded.insert(expanded.end(), devices.begin(), devices.end());
                devices.swap(expanded); // d0 d1 .. d0 d1 ..: consecutive frames go to different GPUs first
        }
        auto cfg_for = [rest](int dev) { return rest + (rest.empty() ? "" : ":") + "dev=" + std::to_string(dev); };
        void *probe = tile_init(parent, cfg_for(devices[0]).c_str()); // validates the options first, then the first device
        if (probe == nullptr || probe == INIT_NOERR) {
                return probe;
        }
        for (int d : devices) { // refuse unusable devices at init time, not on the first frame
                if (d < 0 || set_device(d) != 0) {
                        log_msg(LOG_LEVEL_ERROR, "[MI355X] cannot use HIP device %d\n", d);
                        tile_done(probe);
                        return nullptr;
                }
        }
        tile_done(probe);
        auto *m = new sharded_module();
        m->sharder.reset(new frame_sharder(devices, [=](int) -> tile_encoder_t {
                auto set = std::make_shared<tile_state_set>(tile_done);
                return [=](int dev, unsigned tile, std::shared_ptr<video_frame> in) -> std::shared_ptr<video_frame> {
                        while (set->states.size() <= tile) {
                                void *st = tile_init(parent, cfg_for(dev).c_str());
                                if (st == nullptr || st == INIT_NOERR) return {};
                                set->states.push_back(st);
                        }
                        return tile_compress(set->states[tile], std::move(in));
                };
        }));
        return m;
}

inline void sharded_done(void *state) { delete static_cast<sharded_module *>(state); }
inline void sharded_push(void *state, std::shared_ptr<video_frame> in) { static_cast<sharded_module *>(state)->sharder->push(std::move(in)); }
inline std::shared_ptr<video_frame> sharded_pop(void *state) { return static_cast<sharded_module *>(state)->sharder->pop(); }

} // namespace mi355x
#endif
