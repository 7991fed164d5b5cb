/**
 * @file ug_runtime_harness.cpp
 * The run-time conventions of the drop-in boundary, driven through UltraGrid's OWN compress framework
 * (src/video_compress.cpp + lib_common.cpp + messaging.cpp + module.c compiled from /root/reference by this directory's Makefile)
 * with the threads the reference has: a capture thread calling compress_frame() (rxtx.cpp:182-194), a sender thread popping
 * until the poison pill (rxtx.cpp:260-288), and a control thread sending msg_change_compress_data to "sender.compress" (the path
 * control_socket.cpp uses; module tree root -> sender -> compress as rxtx.cpp:120-131,363 builds it).
 *
 * What it exercises, in any order a script asks for:
 *   - a stream whose video_desc changes while frames are in flight (cuda_dxt.cpp:196-204: compare with the saved desc, reconfigure lazily),
 *   - CHANGE_COMPRESS at run time (video_compress.cpp:154-200: new state first, discard_frames, async_poison(old), delete old),
 *   - compress_done() with frames still queued in the workers and results not yet popped (video_compress.cpp:508-525).
 *
 * usage: ug_runtime_harness <script> <out.rec>
 * script lines (# comments):
 *   init <cfg>                                       compress_init(&sender_mod, cfg, &c); starts the sender thread
 *   init_nosender <cfg>                              the same without a sender thread: the script pops itself (`pop <n>`), as rxtx.cpp:136-141 does when
 *                                                    the sender was never created
 *   pop <n>                                          compress_pop() n frames on the capture thread (no-sender mode only)
 *   frames <set> <codec> <w> <h> <interlacing> <file> <n>   a set of n distinct frames read from file (interlacing: p | i = INTERLACED_MERGED)
 *   push <set> <count>                               compress_frame() count frames, cycling through the set; timestamp = running push index
 *   pace_us <n>                                      sleep between pushes (default 0)
 *   pop_delay_us <n>                                 the sender sleeps that long per popped frame (a slow network)
 *   sender_holds <0|1>                               1: the sender keeps the frame it popped until it has popped the next one (a sender that pipelines its
 *                                                    transmission).  The reference's senders release the frame BEFORE the next compress_pop() (rxtx.cpp:266-284,
 *                                                    hd-rum-recompress.cpp:188); its own video_frame_pool then lets a module's done() wait for that frame
 *                                                    (video_frame_pool.cpp:150-155).  The product's modules must not deadlock either way.
 *   msg <cfg>                                        send_compess_change(): CHANGE_COMPRESS, queued now, handled by the next compress_frame() (video_compress.cpp:338-341)
 *   msg_ctl <delay_ms> <cfg>                         the same from a control thread, <delay_ms> from now: lands wherever the capture thread happens to be
 *   sleep_ms <n>
 *   mark                                             prints "MARK pushed=<n>" (the push index the next frame will get)
 *   pill                                             compress_frame(c, {}) and wait for the sender thread to see it
 *   done                                             the reference's teardown order (rxtx.cpp:133-146): the pill if none was sent yet -- whatever is still queued in
 *                                                    the workers or not yet popped at that moment --, join the sender (it drains everything up to the pill), then
 *                                                    compress_done().  Without a sender thread: compress_done() as is -- it sends the pill itself
 *                                                    (video_compress.cpp:516-518) and deletes the state with the pill never popped.
 *                                                    (compress_done() concurrent with a sender that still pops is NOT a convention of the reference: `delete proxy`
 *                                                    frees the queue the sender's last compress_pop() sleeps on -- ASan shows it in the reference's own queue.)
 * Output records (little endian), in pop order:
 *   "UGRF" u32 index(timestamp) u32 seq u32 width u32 height u32 interlacing u32 tile_count u32 data_len char codec[16] data[data_len]
 * Exit: 0 OK, 2 init refused, 4 watchdog (no progress for UG_RT_WATCHDOG_S seconds, default 60: a hang is a failure, not a timeout of the test).
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "debug.h"
#include "host.h"
#include "lib_common.h"
#include "messaging.h"
#include "module.h"
#include "tv.h"
#include "types.h"
#include "video_codec.h"
#include "video_compress.h"
#include "video_frame.h"

namespace {

struct frame_set {
        struct video_desc desc{};
        std::vector<std::vector<char>> data; // one buffer per frame, all tiles back to back
};

std::atomic<uint64_t> g_progress{0}; // bumped by every push / pop / script step: the watchdog watches it
std::atomic<bool> g_finished{false};

void watchdog()
{
        const char *e = getenv("UG_RT_WATCHDOG_S");
        const int limit = e ? atoi(e) : 60;
        uint64_t last = g_progress;
        int idle = 0;
        while (!g_finished) {
                std::this_thread::sleep_for(std::chrono::milliseconds(250));
                const uint64_t now = g_progress;
                idle = now == last ? idle + 1 : 0;
                last = now;
                if (idle >= limit * 4) {
                        fprintf(stderr, "WATCHDOG: no progress for %d s (progress counter %llu) -- hang\n", limit, (unsigned long long) now);
                        fflush(nullptr);
                        _exit(4);
                }
        }
}

void write_u32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }

} // namespace

int main(int argc, char **argv)
{
        if (argc == 2 && strcmp(argv[1], "list") == 0) {
                list_modules(LIBRARY_CLASS_VIDEO_COMPRESS, VIDEO_COMPRESS_ABI_VERSION, true);
                return 0;
        }
        if (argc < 3) {
                fprintf(stderr, "usage: %s <script> <out.rec>\n", argv[0]);
                return 1;
        }
        std::ifstream script(argv[1]);
        if (!script) { perror(argv[1]); return 1; }
        FILE *rec = fopen(argv[2], "wb");
        if (!rec) { perror(argv[2]); return 1; }

        // root -> sender -> compress, as main.cpp / rxtx.cpp build it
        struct module root{}, sender_mod{};
        module_init_default(&root);
        root.cls = MODULE_CLASS_ROOT;
        module_register(&root, nullptr);
        module_init_default(&sender_mod);
        sender_mod.cls = MODULE_CLASS_SENDER;
        module_register(&sender_mod, &root);

        std::thread dog(watchdog);
        std::map<std::string, frame_set> sets;
        struct compress_state *c = nullptr;
        std::thread sender;
        std::atomic<unsigned> pop_delay_us{0};
        std::atomic<bool> sender_holds{false};
        std::atomic<unsigned> popped_count{0};
        std::vector<std::thread> control_threads;
        std::mutex print_lock;
        unsigned pace_us = 0;
        int64_t pushed = 0;
        bool pill_sent = false;
        int rc = 0;

        auto write_record = [&](const std::shared_ptr<video_frame> &f) {
                uint32_t len = 0;
                for (unsigned t = 0; t < f->tile_count; t++) len += f->tiles[t].data_len;
                fwrite("UGRF", 1, 4, rec);
                write_u32(rec, (uint32_t) f->timestamp);
                write_u32(rec, f->seq);
                write_u32(rec, f->tiles[0].width);
                write_u32(rec, f->tiles[0].height);
                write_u32(rec, (uint32_t) f->interlacing);
                write_u32(rec, f->tile_count);
                write_u32(rec, len);
                char name[16] = {};
                strncpy(name, get_codec_name(f->color_spec), sizeof name - 1);
                fwrite(name, 1, sizeof name, rec);
                for (unsigned t = 0; t < f->tile_count; t++) fwrite(f->tiles[t].data, 1, f->tiles[t].data_len, rec);
                if (f->compress_end < f->compress_start || f->compress_start == 0) {
                        std::lock_guard<std::mutex> lk(print_lock);
                        printf("BAD_TIMES index=%lld\n", (long long) f->timestamp);
                }
                popped_count++;
        };
        auto sender_loop = [&](struct compress_state *c) { // (its own copy of the handle: the capture thread clears the script's after compress_done)
                std::shared_ptr<video_frame> in_transmission; // sender_holds: the previous frame, kept while the next one is popped
                while (std::shared_ptr<video_frame> f = compress_pop(c)) {
                        g_progress++;
                        write_record(f);
                        if (pop_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(pop_delay_us.load()));
                        if (sender_holds) in_transmission = std::move(f);
                        else in_transmission.reset();
                }
                g_progress++;
        };

        // the reference's own call for this (messaging.cpp:465-480: CHANGE_COMPRESS to "sender.compress", asynchronously, as control_socket.c:619-636 does)
        auto send_change = [&](const std::string &cfg, const char *who) {
                send_compess_change(&sender_mod, cfg.c_str());
                std::lock_guard<std::mutex> lk(print_lock);
                printf("MSG %s cfg=%s\n", who, cfg.c_str());
                fflush(stdout);
        };

        std::string line;
        while (std::getline(script, line)) {
                g_progress++;
                std::istringstream is(line);
                std::string cmd;
                if (!(is >> cmd) || cmd[0] == '#') continue;
                if (cmd == "init" || cmd == "init_nosender") {
                        std::string cfg;
                        is >> cfg;
                        const int irc = compress_init(&sender_mod, cfg.c_str(), &c);
                        if (irc != 0) {
                                fprintf(stderr, "compress_init(\"%s\") rc=%d\n", cfg.c_str(), irc);
                                rc = 2;
                                break;
                        }
                        if (cmd == "init") sender = std::thread(sender_loop, c);
                } else if (cmd == "pop") {
                        unsigned n = 0;
                        is >> n;
                        if (sender.joinable() || c == nullptr) { fprintf(stderr, "pop: only without a sender thread\n"); rc = 1; break; }
                        for (unsigned i = 0; i < n; i++) {
                                std::shared_ptr<video_frame> f = compress_pop(c);
                                if (!f) { fprintf(stderr, "pop: pill\n"); rc = 1; break; }
                                write_record(f);
                                g_progress++;
                        }
                } else if (cmd == "frames") {
                        std::string name, codec, il, file;
                        unsigned w = 0, h = 0, n = 0;
                        is >> name >> codec >> w >> h >> il >> file >> n;
                        frame_set fs;
                        fs.desc.width = w; fs.desc.height = h; fs.desc.fps = 30; fs.desc.tile_count = 1;
                        fs.desc.color_spec = get_codec_from_name(codec.c_str());
                        fs.desc.interlacing = il == "i" ? INTERLACED_MERGED : PROGRESSIVE;
                        if (fs.desc.color_spec == VIDEO_CODEC_NONE) { fprintf(stderr, "unknown codec %s\n", codec.c_str()); rc = 1; break; }
                        const size_t len = (size_t) vc_get_linesize(w, fs.desc.color_spec) * h;
                        FILE *in = fopen(file.c_str(), "rb");
                        if (!in) { perror(file.c_str()); rc = 1; break; }
                        for (unsigned i = 0; i < n; i++) {
                                std::vector<char> buf(len + MAX_PADDING);
                                if (fread(buf.data(), 1, len, in) != len) { fprintf(stderr, "short read in %s\n", file.c_str()); rc = 1; break; }
                                fs.data.push_back(std::move(buf));
                        }
                        fclose(in);
                        if (rc) break;
                        sets[name] = std::move(fs);
                } else if (cmd == "push") {
                        std::string name;
                        unsigned count = 0;
                        is >> name >> count;
                        auto it = sets.find(name);
                        if (it == sets.end() || c == nullptr) { fprintf(stderr, "push: no set %s / no state\n", name.c_str()); rc = 1; break; }
                        frame_set &fs = it->second;
                        for (unsigned i = 0; i < count; i++) {
                                // a fresh video_frame per push over the set's buffer (a capture card's ring): its metadata is this push's own
                                struct video_frame *f = vf_alloc_desc(fs.desc);
                                f->tiles[0].data = fs.data[pushed % fs.data.size()].data();
                                f->tiles[0].data_len = (unsigned) (fs.data[0].size() - MAX_PADDING);
                                f->timestamp = pushed++;
                                compress_frame(c, std::shared_ptr<video_frame>(f, vf_free));
                                g_progress++;
                                if (pace_us) std::this_thread::sleep_for(std::chrono::microseconds(pace_us));
                        }
                } else if (cmd == "pace_us") {
                        is >> pace_us;
                } else if (cmd == "pop_delay_us") {
                        unsigned v = 0;
                        is >> v;
                        pop_delay_us = v;
                } else if (cmd == "sender_holds") {
                        unsigned v = 0;
                        is >> v;
                        sender_holds = v != 0;
                } else if (cmd == "msg") {
                        std::string cfg;
                        is >> cfg;
                        send_change(cfg, "capture");
                } else if (cmd == "msg_ctl") {
                        unsigned delay_ms = 0;
                        std::string cfg;
                        is >> delay_ms >> cfg;
                        control_threads.emplace_back([&, delay_ms, cfg] {
                                std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
                                send_change(cfg, "control");
                        });
                } else if (cmd == "sleep_ms") {
                        unsigned v = 0;
                        is >> v;
                        std::this_thread::sleep_for(std::chrono::milliseconds(v));
                } else if (cmd == "mark") {
                        std::lock_guard<std::mutex> lk(print_lock);
                        printf("MARK pushed=%lld popped=%u\n", (long long) pushed, popped_count.load());
                } else if (cmd == "pill") {
                        compress_frame(c, {});
                        pill_sent = true;
                        sender.join();
                } else if (cmd == "done") {
                        for (auto &t : control_threads) t.join(); // (their messages are queued or answered by now; a message to a dead module is the caller's bug)
                        control_threads.clear();
                        {
                                std::lock_guard<std::mutex> lk(print_lock);
                                printf("DONE_CALLED pushed=%lld popped=%u pill=%d\n", (long long) pushed, popped_count.load(), (int) pill_sent);
                                fflush(stdout);
                        }
                        if (sender.joinable()) {
                                if (!pill_sent) compress_frame(c, {});
                                pill_sent = true;
                                sender.join();
                        }
                        compress_done(c); // (no sender and no pill: sends the pill itself, video_compress.cpp:516-518); joins the consumer, deletes the module
                        c = nullptr;
                } else {
                        fprintf(stderr, "unknown script command: %s\n", cmd.c_str());
                        rc = 1;
                        break;
                }
        }
        for (auto &t : control_threads) t.join();
        if (c != nullptr) { // script ended (or failed) without `done`
                if (!pill_sent) compress_frame(c, {});
                if (sender.joinable()) sender.join();
                compress_done(c);
        }
        fclose(rec);
        printf("END pushed=%lld popped=%u rc=%d\n", (long long) pushed, popped_count.load(), rc);
        g_finished = true;
        dog.join();
        module_done(&sender_mod);
        module_done(&root);
        return rc;
}
