/**
 * @file ug_jpeg_parser_fuzz.cpp
 * Sanitizer target ("builder should add ASan/UBSan/TSan to its own harness"): the host half of the JPEG decoder --
 * ug_hip_jpeg_read_info(), the code that meets bytes from the network first (the reference's counterpart is src/utils/jpeg_reader.c
 * behind rtp/rtpdec_jpeg.c) -- under damage.  Built by `make sanitize` from jpeg_decode.hip with -fsanitize=address,undefined on the
 * host side; every mutated header is handed over in a heap block of exactly its length, so that an over-read of one byte is a report.
 * usage: ug_jpeg_parser_fuzz <iterations> <seed.jpg> [<seed.jpg> ...]      (no GPU needed: the parse makes no device call)
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ug_mi355x.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) // xorshift64*: [0, n)
{
        g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
        return (uint32_t) (((g_state * 0x2545F4914F6CDD1Dull) >> 33) % n);
}

int main(int argc, char **argv)
{
        if (argc < 3) {
                fprintf(stderr, "usage: %s <iterations> <seed.jpg> [...]\n", argv[0]);
                return 2;
        }
        const long iters = atol(argv[1]);
        std::vector<std::vector<uint8_t>> seeds;
        for (int i = 2; i < argc; i++) {
                FILE *f = fopen(argv[i], "rb");
                if (!f) { perror(argv[i]); return 2; }
                std::vector<uint8_t> d;
                uint8_t buf[4096];
                size_t n;
                while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
                fclose(f);
                if (d.size() < 64) { fprintf(stderr, "%s: too short\n", argv[i]); return 2; }
                seeds.push_back(d);
        }
        long accepted = 0, refused = 0;
        for (long it = 0; it < iters; it++) {
                std::vector<uint8_t> d = seeds[it % seeds.size()];
                size_t hdr_end = d.size();
                for (size_t i = 0; i + 1 < d.size(); i++) {
                        if (d[i] == 0xFF && d[i + 1] == 0xDA) { hdr_end = i + 14 < d.size() ? i + 14 : d.size(); break; }
                }
                const bool pristine = it < (long) seeds.size(); // the first pass over the seeds: untouched, must be accepted
                const int edits = pristine ? 0 : 1 + (int) rnd(5);
                for (int e = 0; e < edits && hdr_end > 4; e++) {
                        const uint32_t mode = rnd(5), pos = 2 + rnd((uint32_t) hdr_end - 3);
                        static const uint8_t special[] = { 0, 0xFF, 0x7F, 0x80, 1, 0xC0, 0xC4, 0xDB, 0xDA, 0xDD, 0x11, 0x22 };
                        if (mode == 0) d[pos] = (uint8_t) rnd(256);
                        else if (mode == 1) d[pos] = special[rnd(sizeof special)];
                        else if (mode == 2) { const size_t n = 1 + rnd(8); d.erase(d.begin() + pos, d.begin() + (pos + n < d.size() ? pos + n : d.size())); }
                        else if (mode == 3) { for (uint32_t k = 1 + rnd(6); k > 0; k--) d.insert(d.begin() + pos, (uint8_t) rnd(256)); }
                        else if (pos + 1 < d.size()) { const uint16_t v = (uint16_t) rnd(65536); d[pos] = (uint8_t) (v >> 8); d[pos + 1] = (uint8_t) v; } // a length / dimension field
                        if (hdr_end > d.size()) hdr_end = d.size();
                }
                if (!pristine && rnd(2)) d.resize(2 + rnd((uint32_t) (d.size() < hdr_end + 40 ? d.size() : hdr_end + 40) - 2));
                uint8_t *exact = (uint8_t *) malloc(d.size()); // exactly as long as the stream: ASan's red zone starts behind the last byte
                memcpy(exact, d.data(), d.size());
                int w = 0, h = 0, sub = 0, rgb = 0, ri = 0;
                const int rc = ug_hip_jpeg_read_info(exact, d.size(), &w, &h, &sub, &rgb, &ri);
                free(exact);
                if (rc == UG_HIP_SUCCESS) {
                        accepted++;
                        if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (sub != 420 && sub != 422 && sub != 444) || ri < 0 || ri > 65535) {
                                fprintf(stderr, "accepted a header with impossible parameters: %dx%d sub %d ri %d\n", w, h, sub, ri);
                                return 1;
                        }
                } else {
                        refused++;
                }
                if (pristine && rc != UG_HIP_SUCCESS) {
                        fprintf(stderr, "seed %ld refused: %s\n", it, ug_hip_last_error_string());
                        return 1;
                }
        }
        printf("OK iterations=%ld accepted=%ld refused=%ld\n", iters, accepted, refused);
        return accepted > 0 && refused > 0 ? 0 : 1;
}
