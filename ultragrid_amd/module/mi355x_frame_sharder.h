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

#include <sched.h>
#include <string>
#include <strings.h>

#include "debug.h"
#include "host.h"
#include "tv.h"
#include "types.h"
#include "utils/vf_split.h"
#include "utils/video_frame_pool.h"
#include "video_frame.h"

namespace mi355x {

/// encodes one single-tile frame on `device` with the (lazily created) encoder state number `tile`; returns {} on error
using tile_encoder_t = std::function<std::shared_ptr<video_frame>(int device, unsigned tile, std::shared_ptr<video_frame> in)>;
/// encodes several single-tile frames of one geometry in one go (one launch sequence, one synchronisation); result i belongs to
/// frame i, {} = that frame failed.  Optional: a worker without one encodes its queued frames one after another.
using batch_encoder_t = std::function<std::vector<std::shared_ptr<video_frame>>(int device, std::vector<std::shared_ptr<video_frame>> in)>;

class frame_sharder {
public:
        /// one worker per entry of `devices` (a device listed twice gets two workers, i.e. two frames in flight on it).
        /// max_pending > 1 ("batch=<n>"): a worker that is busy takes up to that many frames into its queue instead of making push()
        /// wait, and encodes whatever has queued up as one batch -- throughput for sources that deliver faster than one frame per encode
        /// (files, transcoding); a source at display rate never queues and sees the one-frame path, as with max_pending = 1.
        /// on_worker_start(device) runs first thing on every worker thread -- before the worker's encoder state, and with it the pinned
        /// frame pool, exists (states are created on the worker's first frame): the place to put the thread on the GPU's NUMA node.
        frame_sharder(const std::vector<int> &devices, std::function<tile_encoder_t(int device)> make_encoder, unsigned max_pending = 1,
                      std::function<batch_encoder_t(int device, tile_encoder_t)> make_batch_encoder = nullptr,
                      std::function<void(int device)> on_worker_start = nullptr)
            : m_max_pending(max_pending < 1 ? 1 : max_pending), m_on_worker_start(std::move(on_worker_start))
        {
                for (int d : devices) {
                        auto *w = new worker();
                        w->device = d;
                        w->encode_tile = make_encoder(d);
                        if (make_batch_encoder) w->encode_batch = make_batch_encoder(d, w->encode_tile);
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
                        for (auto &w : m_workers) give(w.get(), {}, true, 0);
                        return;
                }
                // The sequence number travels with the queue entry, not only in the frame: a caller may hand the same frame object in
                // again while an earlier push of it is still queued (file sources, the test harness), and would overwrite it there.
                const uint32_t seq = m_in_seq++;
                in->seq = seq;
                size_t index = 0;
                {       // wait for / select a worker that is not occupied (gpujpeg.cpp:660-673); with batching, else the least loaded one with room
                        std::unique_lock<std::mutex> lk(m_occupancy_lock);
                        m_worker_finished.wait(lk, [&] {
                                size_t best = m_workers.size();
                                for (size_t i = 0; i < m_workers.size(); i++) {
                                        if (m_workers[i]->pending < m_max_pending && (best == m_workers.size() || m_workers[i]->pending < m_workers[best]->pending)) best = i;
                                }
                                index = best;
                                return best != m_workers.size();
                        });
                        m_workers[index]->pending++;
                }
                give(m_workers[index].get(), std::move(in), false, seq);
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
                batch_encoder_t encode_batch;
                unsigned pending = 0; // frames given to the worker and not yet delivered (guarded by m_occupancy_lock)
                std::thread th;
                std::mutex m;
                std::condition_variable cv;
                struct entry {
                        std::shared_ptr<video_frame> frame;
                        bool pill;
                        uint32_t seq;
                };
                std::deque<entry> q;
                bool quit = false;
        };

        void give(worker *w, std::shared_ptr<video_frame> f, bool pill, uint32_t seq)
        {
                {
                        std::lock_guard<std::mutex> lk(w->m);
                        w->q.push_back({ std::move(f), pill, seq });
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

        void finish(worker *w, std::shared_ptr<video_frame> out, uint32_t seq, const char *metadata)
        {
                result r;
                r.seq = seq;
                r.frame = std::move(out);
                if (r.frame) {
                        vf_restore_metadata(r.frame.get(), const_cast<char *>(metadata)); // gpujpeg.cpp:188-193
                        r.frame->seq = r.seq;
                        r.frame->compress_end = get_time_in_ns(); // the async frame API leaves this to the module (gpujpeg.cpp:188-195)
                }
                deliver(std::move(r));
                {
                        std::lock_guard<std::mutex> lk(m_occupancy_lock);
                        w->pending--;
                }
                m_worker_finished.notify_one();
        }

        void run(worker *w)
        {
                if (m_on_worker_start) m_on_worker_start(w->device);
                for (;;) {
                        std::vector<std::shared_ptr<video_frame>> frames; // what has queued up, up to the pill
                        std::vector<uint32_t> seqs;
                        bool pill = false;
                        {
                                std::unique_lock<std::mutex> lk(w->m);
                                w->cv.wait(lk, [&] { return w->quit || !w->q.empty(); });
                                if (w->q.empty()) return; // quit
                                while (!w->q.empty() && !pill && frames.size() < m_max_pending) {
                                        if (w->q.front().pill) {
                                                pill = frames.empty(); // a pill behind frames waits for the next round: order is kept
                                                if (!pill) break;
                                        } else {
                                                frames.push_back(std::move(w->q.front().frame));
                                                seqs.push_back(w->q.front().seq);
                                        }
                                        w->q.pop_front();
                                }
                        }
                        if (pill) {
                                result r;
                                r.pill = true;
                                deliver(std::move(r));
                                continue;
                        }
                        std::vector<std::vector<char>> meta(frames.size(), std::vector<char>(VF_METADATA_SIZE));
                        bool batchable = frames.size() > 1 && (bool) w->encode_batch;
                        for (size_t i = 0; i < frames.size(); i++) {
                                vf_store_metadata(frames[i].get(), meta[i].data());
                                batchable = batchable && frames[i]->tile_count == 1 &&
                                            video_desc_eq(video_desc_from_frame(frames[i].get()), video_desc_from_frame(frames[0].get()));
                        }
                        if (batchable) {
                                std::vector<std::shared_ptr<video_frame>> outs = w->encode_batch(w->device, std::move(frames));
                                for (size_t i = 0; i < seqs.size(); i++) finish(w, i < outs.size() ? std::move(outs[i]) : nullptr, seqs[i], meta[i].data());
                        } else {
                                for (size_t i = 0; i < seqs.size(); i++) finish(w, encode_frame(w, std::move(frames[i])), seqs[i], meta[i].data());
                        }
                }
        }

        std::vector<std::unique_ptr<worker>> m_workers;
        const unsigned m_max_pending;
        const std::function<void(int device)> m_on_worker_start;
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

/// A frame of `pool` that keeps the pool alive.  The reference's video_frame_pool waits in its destructor for every frame it gave out
/// (video_frame_pool.cpp:150-155), so a module that owns its pool by value blocks in done() -- on the capture thread, inside CHANGE_COMPRESS
/// (video_compress.cpp:191-195) -- until the sender has let go of the last compressed frame, and deadlocks with a sender that keeps a frame
/// while it waits for the next one.  Here the encoder state holds the pool through a shared_ptr and every frame holds it too: done() never
/// waits, and the pool (with its pinned buffers) goes when the last frame does.  The frame returns to the pool BEFORE the reference to the pool
/// is dropped -- the other order would be the pool's destructor waiting for the very frame whose deleter runs it.
inline std::shared_ptr<video_frame> get_frame_keeping_pool(const std::shared_ptr<video_frame_pool> &pool)
{
        std::shared_ptr<video_frame> pooled = pool->get_frame();
        video_frame *raw = pooled.get();
        return std::shared_ptr<video_frame>(raw, [pooled = std::move(pooled), pool = std::shared_ptr<video_frame_pool>(pool)](video_frame *) mutable {
                pooled.reset();
                pool.reset();
        });
}

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
using tile_compress_batch_t = std::vector<std::shared_ptr<video_frame>> (*)(void *state, std::vector<std::shared_ptr<video_frame>> in);

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

/// cfg = the module's option string; "dev=<n>[,<n>...]", "workers=<per device>", "batch=<frames>" and "numa=<0|1>" are consumed here, everything
/// else goes to tile_init unchanged (with ":dev=<n>" of the worker -- and ":batch_slices=<frames>" when batching -- appended).  Returns what
/// tile_init returns for a bad / help configuration.  bind_thread (ug_hip_bind_thread_to_device): puts the calling thread on the CPUs of the
/// device's NUMA node; every worker calls it before its first frame unless numa=0.
inline void *sharded_init(struct module *parent, const char *cfg, tile_init_t tile_init, tile_compress_t tile_compress, tile_done_t tile_done,
                          int (*set_device)(int), tile_compress_batch_t tile_compress_batch = nullptr, int (*bind_thread)(int device, int *cpus_bound) = nullptr,
                          int (*numa_node)(int device, int *node) = nullptr)
{
        std::string rest, all = cfg ? cfg : "";
        std::vector<int> devices{ 0 };
        // Two workers per device by default: upload, kernels and download of consecutive frames overlap on one GPU (measured through
        // the reference framework over 4 000 4K frames: DXT5 1 979 -> 3 079 fps, JPEG 2 126 -> 2 393 fps, 8K v210 346 -> 445 fps); workers=1 gives the reference's one-per-device.
        int workers_per_device = 2;
        int batch = 1; // frames a busy worker may queue and then encode together ("batch=<n>"); 1 = the reference's one frame per worker
        bool numa = true; // workers run on the CPUs of their GPU's NUMA node ("numa=0": left to the scheduler, as the reference's workers are)
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
                } else if (strncasecmp(tok.c_str(), "batch=", 6) == 0) {
                        batch = atoi(tok.c_str() + 6);
                } else if (strncasecmp(tok.c_str(), "numa=", 5) == 0) {
                        const char *v = tok.c_str() + 5;
                        numa = !(strcmp(v, "0") == 0 || strcasecmp(v, "no") == 0 || strcasecmp(v, "off") == 0);
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
        if (batch < 1 || batch > 16) {
                log_msg(LOG_LEVEL_ERROR, "[MI355X] batch=<n> must be 1..16\n");
                return nullptr;
        }
        {
                std::vector<int> expanded;
                for (int k = 0; k < workers_per_device; k++) expanded.insert(expanded.end(), devices.begin(), devices.end());
                devices.swap(expanded); // d0 d1 .. d0 d1 ..: consecutive frames go to different GPUs first
        }
        auto cfg_for = [rest, batch](int dev) {
                return rest + (rest.empty() ? "" : ":") + "dev=" + std::to_string(dev) + (batch > 1 ? ":batch_slices=" + std::to_string(batch) : "");
        };
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
        // the per-tile encoder states of a worker are shared by its tile encoder and its batch encoder (which uses state 0)
        auto ensure = [=](const std::shared_ptr<tile_state_set> &set, int dev, unsigned tile) -> void * {
                while (set->states.size() <= tile) {
                        void *st = tile_init(parent, cfg_for(dev).c_str());
                        if (st == nullptr || st == INIT_NOERR) return nullptr;
                        set->states.push_back(st);
                }
                return set->states[tile];
        };
        std::function<batch_encoder_t(int, tile_encoder_t)> make_batch;
        auto current = std::make_shared<std::shared_ptr<tile_state_set>>(); // handed from make_encoder to make_batch of the same worker
        if (tile_compress_batch != nullptr && batch > 1) {
                make_batch = [=](int, tile_encoder_t) -> batch_encoder_t {
                        std::shared_ptr<tile_state_set> set = *current;
                        return [=](int dev, std::vector<std::shared_ptr<video_frame>> in) -> std::vector<std::shared_ptr<video_frame>> {
                                void *st = ensure(set, dev, 0);
                                if (st == nullptr) return {};
                                return tile_compress_batch(st, std::move(in));
                        };
                };
        }
        m->sharder.reset(new frame_sharder(devices, [=](int) -> tile_encoder_t {
                auto set = std::make_shared<tile_state_set>(tile_done);
                *current = set;
                return [=](int dev, unsigned tile, std::shared_ptr<video_frame> in) -> std::shared_ptr<video_frame> {
                        void *st = ensure(set, dev, tile);
                        if (st == nullptr) return {};
                        return tile_compress(st, std::move(in));
                };
        }, (unsigned) batch, make_batch, numa && bind_thread != nullptr ? std::function<void(int)>([bind_thread, numa_node](int dev) {
                int cpus = 0, node = -1;
                const int rc = bind_thread(dev, &cpus);
                const int nrc = numa_node ? numa_node(dev, &node) : 0;
                if (rc == 0 && cpus > 0) {
                        log_msg(LOG_LEVEL_VERBOSE, "[MI355X] worker of device %d runs on the %d CPUs of NUMA node %d (the GPU's)\n", dev, cpus, node);
                } else if (rc != 0 || nrc != 0) { // the worker runs where the scheduler puts it: correct, possibly slower copies -- say so
                        log_msg(LOG_LEVEL_WARNING, "[MI355X] worker of device %d could not be bound to the GPU's NUMA node (bind rc=%d, node query rc=%d); numa=0 silences this\n",
                                dev, rc, nrc);
                }
                if (getenv("UG_MI355X_NUMA_REPORT")) { // test hook: where this worker thread may run now; on stderr -- a host application's stdout may be a pipe it owns
                        std::string cpus_now;
                        cpu_set_t set;
                        if (sched_getaffinity(0, sizeof set, &set) == 0) {
                                for (int c = 0; c < CPU_SETSIZE; c++) {
                                        if (CPU_ISSET(c, &set)) cpus_now += (cpus_now.empty() ? "" : ",") + std::to_string(c);
                                }
                        }
                        fprintf(stderr, "NUMA worker dev=%d node=%d bound=%d rc=%d affinity=%s\n", dev, node, cpus, rc, cpus_now.c_str());
                }
        }) : nullptr));
        return m;
}

inline void sharded_done(void *state) { delete static_cast<sharded_module *>(state); }
inline void sharded_push(void *state, std::shared_ptr<video_frame> in) { static_cast<sharded_module *>(state)->sharder->push(std::move(in)); }
inline std::shared_ptr<video_frame> sharded_pop(void *state) { return static_cast<sharded_module *>(state)->sharder->pop(); }

} // namespace mi355x
#endif
