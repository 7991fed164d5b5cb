// jpeg_decode.hip -- baseline JPEG decoder on gfx950: the receive side of the JPEG path (the rows behind
// src/video_decompress/gpujpeg.c:74-140,292-301, which hands the work to the external libgpujpeg).
//
//   host   the headers only (T.81 B.2): DQT, SOF0, DHT, DRI, SOS, Adobe APP14 -- a few microseconds; tables are uploaded when they change;
//   GPU 1  the entropy-coded data made plain: stuffed zeros and restart markers taken out, a table of where every restart segment starts
//          and ends (E.2.4) -- restart intervals are what makes the entropy-coded data parallel;
//   GPU 2  Huffman decoding (F.2.2), one lane per restart segment on an LDS copy of its bytes: 10-bit look-up for the short codes, a
//          branch-free form of the MAXCODE walk for the long ones; quantised coefficients are collected per block in LDS and written
//          out 128 bytes at a time;
//   GPU 2b scans of LONG segments (other senders' streams: no restart intervals, or a few per frame): self-synchronising parallel decoding, a lane per 1024 bits
//          of a segment -- 1080p without restart intervals 126 ms -> 0.85 ms; see the comment in front of sync_setup;
//   GPU 3  dequantisation + inverse DCT, one lane per 8x8 block: libjpeg's jidctint ("slow but accurate integer": Loeffler-Ligtenberg-
//          Moschytz, 13-bit constants, PASS1_BITS 2) -- integer arithmetic, so the component planes equal libjpeg's bit for bit;
//   GPU 4  planes -> the output codec with the pixel-format kernels the library already has (planar 4:2:2 / 4:2:0 -> UYVY as
//          from_planar.c does it, UYVY -> RGB / RGBA with pixfmt_conv.c's arithmetic), R,G,B planes packed directly.
// Bit-identical to oracle/jpeg_decode_oracle.c, which is pinned to libjpeg-turbo (tests/test_oracle_jpeg_decode.py).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include <vector>

#include "ug_common.h"

namespace {

constexpr int kLutBits = 10;  // AC tables: codes up to 10 bits are one look-up
constexpr int kDcLutBits = 9; // DC tables (the standard ones have nothing longer than 9 / 11 bits)

struct HuffHost {
        uint8_t bits[17] = {};
        uint8_t vals[256] = {};
        bool present = false;
};
// device form of one Huffman table
struct HuffDev {
        uint16_t lut[1 << kLutBits]; // (length << 8) | symbol for codes of at most kLutBits bits, 0 = longer
        uint32_t limit[17];          // the longer codes: (MAXCODE[n] + 1) << (16 - n), carried over lengths without codes
        int16_t offset[17];          // VALPTR[n] - MINCODE[n]
        uint8_t vals[256];
};
// the part of it that the longer codes need, as it sits in LDS
struct LongCodes {
        uint32_t limit[17];
        int16_t offset[17];
        uint8_t vals[256];
};

void build_dev(const HuffHost &h, HuffDev &d, int lut_bits)
{
        memset(&d, 0, sizeof d);
        if (!h.present) return;
        memcpy(d.vals, h.vals, sizeof d.vals);
        int code = 0, k = 0;
        uint32_t limit = 0;
        for (int l = 1; l <= 16; l++) {
                d.offset[l] = (int16_t) (k - code);
                for (int i = 0; i < h.bits[l]; i++, k++, code++) {
                        if (l <= lut_bits) {
                                const int lo = code << (lut_bits - l);
                                for (int f = 0; f < (1 << (lut_bits - l)); f++) d.lut[lo + f] = (uint16_t) (l << 8 | h.vals[k]);
                        }
                }
                if (h.bits[l]) limit = (uint32_t) code << (16 - l);
                d.limit[l] = limit;
                code <<= 1;
        }
}

struct Scan {
        int ns, comp[3], td[3], ta[3];
        size_t data_begin, data_end; // entropy-coded bytes [begin, end) in the stream (end = the marker that follows)
};

struct Header {
        int width = 0, height = 0, ncomp = 0, ri = 0, adobe = -1;
        int hs[3] = { 1, 1, 1 }, vs[3] = { 1, 1, 1 }, tq[3] = { 0, 0, 0 }, cid[3] = { 0, 0, 0 };
        int hmax = 1, vmax = 1, mcu_w = 0, mcu_h = 0;
        uint16_t qt[4][64]; // natural order
        HuffHost dc[4], ac[4];
        std::vector<Scan> scans;
        bool is_rgb() const { return ncomp == 3 && (adobe == 0 || (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B')); }
};

const uint8_t kZigzagHost[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
__device__ const uint8_t kZigzagDev[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

enum ParseMode {
        kHeadersOnly, // stop behind the first SOS header: everything a single-scan stream needs from the host
        kWalkScans,   // walk the entropy-coded data on the host to find where every scan ends and the next one starts (one scan per component)
};

// 0 ok, -1 not a baseline stream this decoder takes, -2 truncated
int parse(const uint8_t *data, size_t len, Header &h, ParseMode mode)
{
        if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
        memset(h.qt, 0, sizeof h.qt);
        for (auto &t : h.dc) t.present = false;
        for (auto &t : h.ac) t.present = false;
        size_t pos = 2;
        while (pos + 4 <= len) {
                if (data[pos] != 0xFF) return -1;
                const int m = data[pos + 1];
                if (m == 0xD9) break;
                if (m == 0xFF) { pos++; continue; } // fill byte
                const size_t seglen = (size_t) data[pos + 2] << 8 | data[pos + 3];
                const uint8_t *s = data + pos + 4;
                if (seglen < 2 || pos + 2 + seglen > len) return -2;
                if (m == 0xDB) {
                        for (size_t o = 0; o + 65 <= seglen - 2; o += 65) {
                                const int pq = s[o] >> 4, t = s[o] & 15;
                                if (pq != 0 || t > 3) return -1;
                                // One table set is uploaded for the whole frame.  A table redefined between the scans of a stream with one
                                // scan per component would have to be applied per scan, as libjpeg does: such streams are refused rather than
                                // decoded with the wrong table (UltraGrid's senders write all tables in front of the first scan).
                                for (const Scan &done : h.scans) {
                                        for (int k = 0; k < done.ns; k++) {
                                                if (h.tq[done.comp[k]] == t) return -1;
                                        }
                                }
                                for (int k = 0; k < 64; k++) h.qt[t][kZigzagHost[k]] = s[o + 1 + k];
                        }
                } else if (m == 0xC0) {
                        // one frame header per stream, in front of the first scan (T.81 B.2.1; libjpeg refuses both a duplicate SOF and one behind an SOS):
                        // a later one would change the geometry under the scans already recorded and under the caller's size check
                        if (h.width || !h.scans.empty()) return -1;
                        if (seglen < 8 || s[0] != 8) return -1;
                        h.hmax = h.vmax = 1;
                        h.height = s[1] << 8 | s[2];
                        h.width = s[3] << 8 | s[4];
                        h.ncomp = s[5];
                        if ((h.ncomp != 1 && h.ncomp != 3) || seglen < (size_t) 8 + 3 * h.ncomp || !h.width || !h.height) return -1;
                        if (h.width > 16384 || h.height > 16384) return -1; // twice 8K: nothing UltraGrid carries; bounds the work buffers a header can ask for
                        for (int c = 0; c < h.ncomp; c++) {
                                h.cid[c] = s[6 + 3 * c];
                                h.hs[c] = s[7 + 3 * c] >> 4;
                                h.vs[c] = s[7 + 3 * c] & 15;
                                h.tq[c] = s[8 + 3 * c];
                                if (h.hs[c] < 1 || h.hs[c] > 2 || h.vs[c] < 1 || h.vs[c] > 2 || h.tq[c] > 3) return -1;
                                h.hmax = h.hs[c] > h.hmax ? h.hs[c] : h.hmax;
                                h.vmax = h.vs[c] > h.vmax ? h.vs[c] : h.vmax;
                        }
                        if (h.ncomp == 1) h.hs[0] = h.vs[0] = h.hmax = h.vmax = 1; // one component is never interleaved: its factors mean nothing (T.81 A.2.2; libjpeg does the same)
                        h.mcu_w = (h.width + 8 * h.hmax - 1) / (8 * h.hmax);
                        h.mcu_h = (h.height + 8 * h.vmax - 1) / (8 * h.vmax);
                } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                        return -1; // extended / progressive / lossless / arithmetic: not baseline
                } else if (m == 0xC4) {
                        for (size_t o = 0; o + 17 <= seglen - 2;) {
                                const int tc = s[o] >> 4, th = s[o] & 15;
                                if (th > 3 || tc > 1) return -1;
                                for (const Scan &done : h.scans) { // see DQT above: no per-scan table sets
                                        for (int k = 0; k < done.ns; k++) {
                                                if ((tc ? done.ta[k] : done.td[k]) == th) return -1;
                                        }
                                }
                                HuffHost &t = tc ? h.ac[th] : h.dc[th];
                                int n = 0, code = 0;
                                t.bits[0] = 0;
                                for (int l = 1; l <= 16; l++) {
                                        n += (t.bits[l] = s[o + l]);
                                        code += t.bits[l];
                                        if (code > (1 << l)) return -1; // more codes of this length than there are (C.2): the look-ups are sized by this
                                        code <<= 1;
                                }
                                if (n > 256 || o + 17 + n > seglen - 2) return -1;
                                memset(t.vals, 0, sizeof t.vals);
                                memcpy(t.vals, s + o + 17, (size_t) n);
                                t.present = true;
                                o += 17 + (size_t) n;
                        }
                } else if (m == 0xDD) {
                        if (seglen < 4) return -1;
                        h.ri = s[0] << 8 | s[1];
                } else if (m == 0xEE && seglen >= 14 && memcmp(s, "Adobe", 5) == 0) {
                        h.adobe = s[11];
                } else if (m == 0xDA) {
                        if (!h.width || seglen < 8) return -1; // (seglen 2 would put s[0] behind the buffer)
                        Scan sc;
                        sc.ns = s[0];
                        if (sc.ns < 1 || sc.ns > h.ncomp || seglen < (size_t) 6 + 2 * sc.ns) return -1;
                        for (int k = 0; k < sc.ns; k++) {
                                sc.comp[k] = -1;
                                for (int c = 0; c < h.ncomp; c++) {
                                        if (h.cid[c] == s[1 + 2 * k]) sc.comp[k] = c;
                                }
                                sc.td[k] = s[2 + 2 * k] >> 4;
                                sc.ta[k] = s[2 + 2 * k] & 15;
                                if (sc.comp[k] < 0 || sc.td[k] > 3 || sc.ta[k] > 3 || !h.dc[sc.td[k]].present || !h.ac[sc.ta[k]].present) return -1;
                        }
                        if (sc.ns != 1 && sc.ns != h.ncomp) return -1;
                        sc.data_begin = pos + 2 + seglen;
                        if (mode == kHeadersOnly) {
                                sc.data_end = len;
                                h.scans.push_back(std::move(sc));
                                break;
                        }
                        // walk the entropy-coded data: RSTn markers start new segments, any other marker ends the scan
                        size_t q = sc.data_begin;
                        for (;;) {
                                const uint8_t *f = (const uint8_t *) memchr(data + q, 0xFF, len - q);
                                if (!f || f + 1 >= data + len) { q = len; break; }
                                q = (size_t) (f - data);
                                const int n = data[q + 1];
                                if (n == 0x00 || n == 0xFF) { q += n == 0 ? 2 : 1; continue; }
                                if (n >= 0xD0 && n <= 0xD7) {
                                        q += 2;
                                                        continue;
                                }
                                break; // a real marker
                        }
                        sc.data_end = q;
                        h.scans.push_back(std::move(sc));
                        pos = q;
                        continue;
                }
                pos += 2 + seglen;
        }
        if (!h.width || h.scans.empty()) return -1;
        return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ---- pass 1: the entropy-coded data made plain, on the GPU -----------------------------------------------------------------------------
// Inside entropy-coded data 0xFF is followed by 0x00 (a stuffed byte: the data byte is 0xFF), by D0..D7 (a restart marker: the next segment
// starts behind it) or by anything else (a marker that ends the scan, or damage).  Two kernels over the bytes of a scan, 4 KiB per
// workgroup, turn it into what the Huffman kernel wants to read: the same bytes with the stuffed zeros and the restart markers taken out
// ("clean" stream, positions counted from the start of the scan), plus, per restart segment, where it starts and where it ends in there
// (the end: the next restart marker, or an earlier marker of another kind, or the end of the data).  Pass A counts what every 4 KiB
// removes and how many restart markers it holds, pass B turns the counts into positions.
constexpr int kScanWG = 256, kScanBytesPerLane = 16, kScanChunk = kScanWG * kScanBytesPerLane;

struct PieceMasks {
        uint32_t drop, rst, marker; // per byte of the 16: taken out / first byte of a restart marker / first byte of any other marker
};
// the 16 bytes at pos (16-byte aligned) of a scan that occupies [begin, limit)
__device__ __forceinline__ PieceMasks classify(const uint8_t *__restrict__ stream, size_t pos, size_t begin, size_t limit, uint4 &bytes)
{
        PieceMasks m = { 0, 0, 0 };
        if (pos >= limit) return m;
        const uint4 q = *(const uint4 *) (stream + pos);
        bytes = q;
        const uint32_t before = pos > begin ? stream[pos - 1] : 0, after = stream[pos + 16]; // the buffer has slack behind the data
        const uint32_t w[4] = { q.x, q.y, q.z, q.w };
        uint32_t prev = before;
#pragma unroll
        for (int i = 0; i < 16; i++) {
                const uint32_t b = (w[i / 4] >> (8 * (i % 4))) & 0xff;
                const uint32_t next = i < 15 ? (w[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xff : after;
                const bool in = pos + i >= begin && pos + i < limit;
                const bool next_in = pos + i + 1 < limit;
                const bool ff = b == 0xFF;
                const bool rst1 = ff && next_in && (next & 0xF8) == 0xD0;
                const bool rst2 = prev == 0xFF && (b & 0xF8) == 0xD0 && pos + i > begin;
                const bool stuffed = prev == 0xFF && b == 0x00 && pos + i > begin;
                const bool other = ff && !rst1 && !(next_in && next == 0x00);
                m.drop |= (uint32_t) (in && (rst1 || rst2 || stuffed)) << i;
                m.rst |= (uint32_t) (in && rst1) << i;
                m.marker |= (uint32_t) (in && other) << i;
                prev = b;
        }
        return m;
}

__device__ __forceinline__ int wg_sum_256(int v, int *lds4)
{
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
        __syncthreads();
        const int t = lds4[0] + lds4[1] + lds4[2] + lds4[3];
        __syncthreads();
        return t;
}

// counts[2 * wg] = bytes the workgroup's 4 KiB lose, counts[2 * wg + 1] = restart markers in them
// also: seg_end[0 .. n_seg) = 0xFFFFFFFF, what pass B's atomicMin starts from
__global__ __launch_bounds__(kScanWG) void clean_count_kernel(const uint8_t *__restrict__ stream, size_t base, size_t begin, size_t limit, int *__restrict__ counts,
                                                              uint32_t *__restrict__ seg_end, int n_seg)
{
        __shared__ int part[4];
        for (int i = blockIdx.x * kScanWG + threadIdx.x; i < n_seg; i += gridDim.x * kScanWG) seg_end[i] = 0xFFFFFFFFu;
        const size_t pos = base + (size_t) blockIdx.x * kScanChunk + threadIdx.x * kScanBytesPerLane;
        uint4 bytes;
        const PieceMasks m = classify(stream, pos, begin, limit, bytes);
        const int dropped = wg_sum_256(__popc(m.drop), part), rsts = wg_sum_256(__popc(m.rst), part);
        if (threadIdx.x == 0) {
                counts[2 * blockIdx.x] = dropped;
                counts[2 * blockIdx.x + 1] = rsts;
        }
}

// seg_end holds 0xFFFFFFFF on entry (pass A); found[0] = number of segments (1 + restart markers); entries at index >= cap are not written
__global__ __launch_bounds__(kScanWG) void clean_place_kernel(const uint8_t *__restrict__ stream, size_t base, size_t begin, size_t limit, const int *__restrict__ counts,
                                                              uint8_t *__restrict__ clean, uint32_t *__restrict__ seg_start, uint32_t *__restrict__ seg_end, int cap,
                                                              int *__restrict__ found)
{
        __shared__ int part[4];
        __shared__ int wave_drop[4], wave_rst[4];
        int drop_before = 0, rst_before = 0;
        for (int j = threadIdx.x; j < (int) blockIdx.x; j += kScanWG) {
                drop_before += counts[2 * j];
                rst_before += counts[2 * j + 1];
        }
        drop_before = wg_sum_256(drop_before, part);
        rst_before = wg_sum_256(rst_before, part);
        const size_t pos = base + (size_t) blockIdx.x * kScanChunk + threadIdx.x * kScanBytesPerLane;
        uint4 bytes = make_uint4(0, 0, 0, 0);
        const PieceMasks m = classify(stream, pos, begin, limit, bytes);
        const int my_drop = __popc(m.drop), my_rst = __popc(m.rst);
        int inc_drop = my_drop, inc_rst = my_rst; // inclusive scans over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
                const int a = __shfl_up(inc_drop, o), c = __shfl_up(inc_rst, o);
                if ((int) (threadIdx.x & 63) >= o) {
                        inc_drop += a;
                        inc_rst += c;
                }
        }
        if ((threadIdx.x & 63) == 63) {
                wave_drop[threadIdx.x >> 6] = inc_drop;
                wave_rst[threadIdx.x >> 6] = inc_rst;
        }
        __syncthreads();
        int dropped = drop_before + inc_drop - my_drop, seg = rst_before + inc_rst - my_rst; // before this lane's first byte
        for (int w = 0; w < (int) (threadIdx.x >> 6); w++) {
                dropped += wave_drop[w];
                seg += wave_rst[w];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) seg_start[0] = 0;
        if (pos < limit) {
                // position in the clean stream of this lane's first byte, were it kept
                const long first = (long) pos - (long) begin - dropped; // negative only for the bytes in front of the scan, which are not written
                const int lo = pos < begin ? (int) (begin - pos) : 0, hi = limit - pos < 16 ? (int) (limit - pos) : 16;
                if (m.drop == 0 && lo == 0 && hi == 16) {
                        __builtin_memcpy(clean + first, &bytes, 16);
                } else {
                        const uint32_t w[4] = { bytes.x, bytes.y, bytes.z, bytes.w };
                        long at = first + lo; // bytes in front of the scan count as neither kept nor dropped
                        for (int i = lo; i < hi; i++) {
                                if (!(m.drop >> i & 1)) clean[at++] = (uint8_t) (w[i / 4] >> (8 * (i % 4)));
                        }
                }
                uint32_t events = m.rst | m.marker;
                while (events) { // rare: a lane with a marker in its bytes
                        const int i = __ffs(events) - 1;
                        events &= events - 1;
                        const int s_here = seg + __popc(m.rst & ((1u << i) - 1));
                        const uint32_t at = (uint32_t) ((long) pos + i - (long) begin - dropped - __popc(m.drop & ((1u << i) - 1)));
                        if (s_here < cap) atomicMin(seg_end + s_here, at); // whichever marker comes first ends the segment
                        if ((m.rst >> i & 1) && s_here + 1 < cap) seg_start[s_here + 1] = at;
                }
        }
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanWG - 1) {
                const int total_drop = dropped + my_drop, total_rst = seg + my_rst;
                found[0] = total_rst + 1;
                if (total_rst < cap) atomicMin(seg_end + total_rst, (uint32_t) ((long) limit - (long) begin - total_drop));
        }
}

// ---- pass 2: Huffman decoding --------------------------------------------------------------------------------------------------------
struct ScanDev {
        int ns, comp[3], td[3], ta[3], nbh[3], nbv[3], gw[3]; // blocks per unit and blocks per row of each component's grid
        int single, bw1, mcu_w, ri;
        int units;
        int16_t *coef[3];
        int n_dc, n_ac, dc_tab[3], ac_tab[3]; // the distinct tables of the scan ...
        int dc_slot[3], ac_slot[3];           // ... and which of them each component uses
};

// The bytes of one segment in the workgroup's LDS copy of the clean stream.  Every symbol: if 32 bits or fewer are left, the window takes
// the four bytes that were fetched from LDS one symbol earlier -- no byte-stuffing logic, no branches; behind the end of the segment the
// window fills with zero bits.
struct BitReader {
        uint32_t pos, end; // next unread byte and end of the segment, as offsets into the staged bytes
        uint32_t nxt_lo, nxt_hi; // the aligned words around pos
        unsigned long long acc; // bits are consumed from the top
        int cnt;
};

// offset of the tiles in the kernel's LDS, = what the tables in front of them take
__host__ __device__ constexpr int lds_tile_offset(int n_dc, int n_ac)
{
        return (n_dc * (2 << kDcLutBits) + n_ac * (2 << kLutBits) + (n_dc + n_ac) * (int) sizeof(LongCodes) + 15) & ~15;
}
constexpr size_t kMaxStage = 40 * 1024; // with the tables and tiles: under the 64 KiB a kernel gets without asking
constexpr int kTileWords = 36; // 32 words of coefficients, padded: 16-byte aligned rows that start in different LDS banks

// One lane per restart segment, `lanes` (8..64) of them per one-wave workgroup: with the usual few thousand segments per frame that is
// fewer lanes per wave than a wave has, on purpose -- the symbol loop runs as long as its slowest lane, the machine has 1024 SIMDs, and a
// frame has no more than a few hundred full waves of segments to offer.  The workgroup copies the stretch of the clean stream that its
// segments occupy into LDS first (they are adjacent), so the symbol loop never waits on global memory; lanes whose bytes did not fit read
// them from memory.  The lanes walk their segments block by block in step (every segment holds the same blocks in the same order), each
// collecting the coefficients of its current block, in zigzag order, in a private tile in LDS; when the block is done the wave writes the
// tiles out together, 128 contiguous bytes per block, zeros included -- so the coefficient planes need no clearing and no lane issues
// scattered 2-byte stores -- and leaves the tiles zeroed for the next block.  The coefficient planes are in zigzag order; the IDCT kernel
// undoes it with compile-time indices.
// amdgpu_waves_per_eu(1, 1): the launch is shaped for one wave per SIMD (above), but nothing makes the dispatcher spread 1 013 waves over 1 024 SIMDs
// that way -- two on one SIMD take turns at every instruction of a latency-bound loop.  The attribute enforces it: the compiler requests 264 registers
// for the 44 the kernel uses (.amdhsa_next_free_vgpr 257), so a second wave of THIS kernel does not fit a SIMD.  Whole 4K decode calls
// 142.4-143.2 -> 139.0-139.8 us (4:2:2), 162.0-162.5 -> 156.9 (4:2:0), interleaved A/B; (1, 2) and (2, 2) equal the default
// (profiles/r05_jpeg_decode_occupancy.txt).  Waves of other kernels (up to 248 registers) still share the SIMD; the Huffman kernels of two decoders
// running on two streams no longer do -- which costs nothing measurable: four decoders on four streams 90.2-91.7 us per frame without the attribute,
// 89.1-90.4 with it (three interleaved rounds).
__global__ __attribute__((amdgpu_waves_per_eu(1, 1))) __launch_bounds__(64) void huff_decode_kernel(const uint8_t *__restrict__ clean, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end,
                                                         int n_seg, const int *__restrict__ found, int lanes, int stage_bytes, ScanDev sp,
                                                         const HuffDev *__restrict__ tabs /* [0..3] DC, [4..7] AC */)
{
        // LDS, all of it sized at launch (lds_bytes() below): the scan's distinct DC look-ups (512 entries each), AC look-ups (1024 each),
        // their long-code tables, one tile per lane in use, the staged stretch of the stream
        extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
        __shared__ int tile_dst[64]; // where the lane's tile goes (in blocks), -1: nowhere
        uint16_t *const lut_dc = (uint16_t *) lds;
        uint16_t *const lut_ac = lut_dc + sp.n_dc * (1 << kDcLutBits);
        LongCodes *const longs = (LongCodes *) (lut_ac + sp.n_ac * (1 << kLutBits)); // DC tables first
        uint32_t *const tile = (uint32_t *) (lds + lds_tile_offset(sp.n_dc, sp.n_ac));
        uint8_t *const stage = (uint8_t *) (tile + lanes * kTileWords);
        const int lane = threadIdx.x;
        const int seg0 = blockIdx.x * lanes;
        const int n_found = min(n_seg, found[0]);
        // ---- the workgroup's stretch of the clean stream -> LDS ----
        uint32_t stage_begin = 0, staged = 0;
        bool all_staged = true; // workgroup-uniform: every byte any lane can ask for is in LDS
        if (seg0 < n_found) {
                const int last = min(seg0 + lanes, n_found) - 1;
                stage_begin = seg_start[seg0] & ~15u;
                const uint32_t stretch = seg_end[last] - stage_begin;
                staged = min((stretch + 16u + 15u) & ~15u, (uint32_t) stage_bytes); // 16 more: the look-ahead of the last segment stays inside
                all_staged = staged >= stretch + 12u;
                for (uint32_t o = lane * 16; o < staged; o += 64 * 16) *(uint4 *) (stage + o) = *(const uint4 *) (clean + stage_begin + o);
        }
        for (int j = 0; j < sp.n_dc + sp.n_ac; j++) {
                const bool dc = j < sp.n_dc;
                const HuffDev &t = tabs[dc ? sp.dc_tab[j] : 4 + sp.ac_tab[j - sp.n_dc]];
                const uint4 *src = (const uint4 *) t.lut;
                uint4 *dst = (uint4 *) (dc ? lut_dc + j * (1 << kDcLutBits) : lut_ac + (j - sp.n_dc) * (1 << kLutBits));
                for (int i = lane; i < (dc ? (2 << kDcLutBits) : (2 << kLutBits)) / 16; i += 64) dst[i] = src[i];
                LongCodes &lc = longs[j];
                for (int i = lane; i < 256; i += 64) lc.vals[i] = t.vals[i];
                if (lane < 17) {
                        lc.limit[lane] = t.limit[lane];
                        lc.offset[lane] = t.offset[lane];
                }
        }
        if (lane < lanes) {
#pragma unroll
                for (int j = 0; j < 8; j++) *(uint4 *) (tile + lane * kTileWords + 4 * j) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        const int seg = seg0 + lane;
        const bool mine = lane < lanes && seg < n_seg;
        const bool present = mine && seg < n_found; // a segment the stream does not have is an empty one: zero bits, decoded like any others
        BitReader br;
        br.pos = present ? seg_start[seg] - stage_begin : 0;
        br.end = present ? seg_end[seg] - stage_begin : 0;
        br.acc = 0;
        br.cnt = 0;
        const uint8_t *const far = clean + stage_begin; // for what did not fit in LDS
        // the two aligned words that hold the four bytes at `at`; put together when they are used, one symbol later
        auto fetch = [&](auto far_c, uint32_t at) {
                if constexpr (decltype(far_c)::value) {
                        if (__builtin_expect(at + 8 > staged, 0)) {
                                const uint32_t *two = (const uint32_t *) (far + (at & ~3u)); // the clean buffer has slack behind the data
                                br.nxt_lo = two[0];
                                br.nxt_hi = two[1];
                                asm volatile("" : "+v"(br.nxt_lo), "+v"(br.nxt_hi)); // completes here: the common path never waits on memory
                                return;
                        }
                }
                const uint32_t *two = (const uint32_t *) (stage + (at & ~3u));
                br.nxt_lo = two[0];
                br.nxt_hi = two[1];
        };
        // one Huffman symbol and the `symbol & 15` extra bits behind it (F.2.2.1, sign extension of Figure F.12): code (<= 16 bits) + extra bits
        // (<= 15) always fit in the top 32 bits of the window after the top-up
        auto symbol = [&](auto far_c, auto bits_c, const uint16_t *lut, const LongCodes &lc, int &value) -> int {
                constexpr int kBits = decltype(bits_c)::value;
                {
                        const bool need = br.cnt <= 32;
                        uint32_t w = __builtin_amdgcn_alignbyte(br.nxt_hi, br.nxt_lo, br.pos);
                        if (__builtin_expect(br.pos + 4 > br.end, 0)) { // the last bytes of the segment: zero bits behind them
                                const int left = (int) br.end - (int) br.pos;
                                w = left <= 0 ? 0 : w & (0xFFFFFFFFu >> (8 * (4 - left)));
                        }
                        const unsigned long long in = need ? __builtin_bswap32(w) : 0u;
                        br.acc |= in << ((32 - br.cnt) & 63);
                        br.cnt += need ? 32 : 0;
                        br.pos += need && br.pos < br.end ? 4 : 0; // never far behind the end: what is fetched there is masked anyway
                        fetch(far_c, br.pos);
                }
                const uint32_t hi = (uint32_t) (br.acc >> 32);
                const unsigned e = lut[hi >> (32 - kBits)];
                int l = (int) (e >> 8), sym = (int) (e & 0xff);
                if (__builtin_expect(e == 0, 0)) {
                        // a code longer than the look-up covers: codes of length n fill [.., limit[n]) of the 16-bit prefixes, limits ascending
                        // (F.2.2.3's MAXCODE walk without the loop)
                        const unsigned pk = hi >> 16;
                        l = kBits + 1;
#pragma unroll
                        for (int n = kBits + 1; n <= 16; n++) l += pk >= lc.limit[n];
                        if (l > 16) { // no such code (damaged data): the MAXCODE walk has read 17 bits by the time it gives up (libjpeg likewise), symbol 0
                                l = 17;
                                sym = 0;
                        } else {
                                sym = lc.vals[(int) (pk >> (16 - l)) + lc.offset[l] & 0xff];
                        }
                }
                const int sz = sym & 15;
                const int v = (int) (((hi << l) >> 1) >> (31 - sz)); // the sz bits behind the code; 0 for sz == 0
                value = v < ((1 << sz) >> 1) ? v + 1 - (1 << sz) : v;
                br.acc <<= l + sz;
                br.cnt -= l + sz;
                return sym;
        };
        int pred[3] = { 0, 0, 0 };
        const int per_seg = sp.ri && sp.ri < sp.units ? sp.ri : sp.units;
        const int u0 = seg * per_seg;
        int16_t *const my = (int16_t *) (tile + lane * kTileWords);
        const int rounds = (lanes + 7) / 8;
        // the walk over the segment, compiled twice: without the test for bytes outside LDS when the workgroup's whole stretch is staged (the
        // usual case; the choice is workgroup-uniform), with it otherwise
        auto walk = [&](auto far_c) {
                fetch(far_c, br.pos);
                for (int i = 0; i < per_seg; i++) {
                        const int u = u0 + i;
                        const bool active = mine && u < sp.units;
                        const int row_units = sp.single ? sp.bw1 : sp.mcu_w;
                        const int uy = u / row_units, ux = u - uy * row_units;
                        for (int k = 0; k < sp.ns; k++) {
                                for (int by = 0; by < sp.nbv[k]; by++) {
                                        for (int bx = 0; bx < sp.nbh[k]; bx++) {
                                                tile_dst[lane] = active ? (uy * sp.nbv[k] + by) * sp.gw[k] + ux * sp.nbh[k] + bx : -1;
                                                if (active) {
                                                        const uint16_t *const ac_lut = lut_ac + sp.ac_slot[k] * (1 << kLutBits);
                                                        const LongCodes &ac_long = longs[sp.n_dc + sp.ac_slot[k]];
                                                        int v;
                                                        symbol(far_c, std::integral_constant<int, kDcLutBits>(), lut_dc + sp.dc_slot[k] * (1 << kDcLutBits), longs[sp.dc_slot[k]], v);
                                                        pred[k] += v;
                                                        my[0] = (int16_t) pred[k];
                                                        for (int z = 1; z < 64; z++) {
                                                                const int rs = symbol(far_c, std::integral_constant<int, kLutBits>(), ac_lut, ac_long, v);
                                                                if ((rs & 15) == 0 && rs != 0xF0) break; // EOB
                                                                z += rs >> 4;
                                                                my[(rs & 15) && z < 64 ? z : 64] = (int16_t) v; // slot 64 is the tile's padding
                                                        }
                                                }
                                                __syncthreads();
                                                // 8 lanes per tile, 16 bytes each: 8 tiles per round
                                                for (int round = 0; round < rounds; round++) {
                                                        const int t = round * 8 + (lane >> 3), part = lane & 7;
                                                        if (t < lanes) {
                                                                const int dst = tile_dst[t];
                                                                uint4 *src = (uint4 *) (tile + t * kTileWords + part * 4);
                                                                const uint4 q = *src;
                                                                *src = make_uint4(0, 0, 0, 0);
                                                                if (dst >= 0) *(uint4 *) (sp.coef[k] + (size_t) dst * 64 + part * 8) = q;
                                                        }
                                                }
                                                __syncthreads();
                                        }
                                }
                        }
                }
        };
        if (all_staged) walk(std::false_type());
        else walk(std::true_type());
}

// ---- pass 2b: scans with LONG segments (no restart intervals, or few of them) -- self-synchronising parallel Huffman decoding ------------------------------------
// One segment = one lane in the kernel above: 126 ms for a 1080p frame of a sender that writes no restart markers (libjpeg's default), 16 ms with FFmpeg's eight
// slices (profiles/r06_decode_no_restart.txt).  Huffman-coded data synchronises itself: a decoder started at an arbitrary bit falls into step with the real code
// boundaries after a few symbols.  So (Klein & Wiseman 2003; Weissenberger & Schmidt 2018 / 2021 for JPEG): cut every segment into chunks of kSyncChunkBits, let one
// lane per chunk decode from a guessed state, hand every chunk's EXIT state (bit position, block of the unit, zigzag index -- where the sequential decoder would stand)
// to its right neighbour as that one's start, and decode again whoever's start changed, until nothing changes: a fixed point, and since the first chunk of every
// segment starts from the true state it is the sequential decoder's own chain of states -- for any data, damaged or not.  Then a prefix sum over the blocks completed per
// chunk tells every chunk which block it starts in, one more pass writes the coefficients (DC as differences), and a running sum per component in scan order, started
// anew at every restart interval, turns the differences into DC values.  Bit-identical to the one-lane walk; taken when the segments of a scan are long enough for it
// to be the faster way (the rule is where the choice is made), and only when every segment holds all its blocks (data that ends early goes the sequential way, whose zero-bit tail it would otherwise have to imitate).
constexpr int kSyncChunkBits = 1024;
constexpr int kSyncWG = 256;
constexpr int kSyncWarm = 16, kSyncOwn = kSyncWG - kSyncWarm; // chunks a workgroup of the settling kernel runs ahead of its own (see there); chunks it owns
constexpr size_t kSyncMinBytes = 4096;
constexpr int kSyncMaxUnitBlocks = 12; // blocks of one unit: 3 components of up to 2 x 2 (the layouts the output stage takes have at most 6)

__device__ __forceinline__ unsigned long long sync_pack(uint32_t p, int blk, int z) { return (unsigned long long) p << 16 | (unsigned) blk << 8 | (unsigned) z; }

// what the lanes index by the block they are in: kept in LDS (a lane-varying index into the kernel's arguments would send them through scratch memory)
struct SyncLds {
        uint16_t *lut_dc, *lut_ac;
        LongCodes *longs;
        int blk_k[kSyncMaxUnitBlocks], blk_by[kSyncMaxUnitBlocks], blk_bx[kSyncMaxUnitBlocks]; // the blocks of a unit in scan order: component of the scan, row, column
        int blk_dc[kSyncMaxUnitBlocks], blk_ac[kSyncMaxUnitBlocks];                             // their table slots
        int nbh[3], nbv[3], gw[3];
        int16_t *coef[3];
        int per_unit, row_units, n_dc;
};

// the scan's tables -> LDS (the layout of huff_decode_kernel); s is in LDS too, filled by lane 0
__device__ void sync_setup(uint8_t *lds, SyncLds *s, const ScanDev &sp, const HuffDev *__restrict__ tabs)
{
        const int tid = threadIdx.x;
        uint16_t *const lut_dc = (uint16_t *) lds, *const lut_ac = lut_dc + sp.n_dc * (1 << kDcLutBits);
        LongCodes *const longs = (LongCodes *) (lut_ac + sp.n_ac * (1 << kLutBits));
        for (int j = 0; j < sp.n_dc + sp.n_ac; j++) {
                const bool dc = j < sp.n_dc;
                const HuffDev &t = tabs[dc ? sp.dc_tab[j] : 4 + sp.ac_tab[j - sp.n_dc]];
                uint16_t *dst = dc ? lut_dc + j * (1 << kDcLutBits) : lut_ac + (j - sp.n_dc) * (1 << kLutBits);
                for (int i = tid; i < (dc ? (1 << kDcLutBits) : (1 << kLutBits)); i += blockDim.x) dst[i] = t.lut[i];
                LongCodes &lc = longs[j];
                for (int i = tid; i < 256; i += blockDim.x) lc.vals[i] = t.vals[i];
                if (tid < 17) {
                        lc.limit[tid] = t.limit[tid];
                        lc.offset[tid] = t.offset[tid];
                }
        }
        if (tid == 0) {
                s->lut_dc = lut_dc; s->lut_ac = lut_ac; s->longs = longs;
                int n = 0;
                for (int k = 0; k < sp.ns; k++) {
                        s->nbh[k] = sp.nbh[k]; s->nbv[k] = sp.nbv[k]; s->gw[k] = sp.gw[k]; s->coef[k] = sp.coef[k];
                        for (int by = 0; by < sp.nbv[k]; by++) {
                                for (int bx = 0; bx < sp.nbh[k]; bx++, n++) {
                                        if (n < kSyncMaxUnitBlocks) {
                                                s->blk_k[n] = k; s->blk_by[n] = by; s->blk_bx[n] = bx;
                                                s->blk_dc[n] = sp.dc_slot[k]; s->blk_ac[n] = sp.ac_slot[k];
                                        }
                                }
                        }
                }
                s->per_unit = n;
                s->row_units = sp.single ? sp.bw1 : sp.mcu_w;
                s->n_dc = sp.n_dc;
        }
        __syncthreads();
}

// The sequential decoder's steps from state (p, blk, z) until p >= stop_bits: p = bit position in the clean stream, blk = block of the unit, z = 0 in front of a DC
// symbol, else the zigzag index the next AC symbol starts from; bits from end_bits on (the end of the segment) read as zero.  nb counts the blocks completed.  WRITE:
// coefficients go to their planes (zigzag order, DC as the difference), blocks from number `first_block` on, those below `limit` only.  The symbol arithmetic is
// huff_decode_kernel's.
template <bool WRITE>
__device__ __forceinline__ void sync_decode(const uint8_t *__restrict__ clean, uint32_t end_bits, uint32_t stop_bits, uint32_t &p, int &blk, int &z, uint32_t &nb, const SyncLds &s,
                                            uint32_t first_block, uint32_t limit)
{
        const uint32_t *const words = (const uint32_t *) clean;
        uint32_t n = first_block;    // number of the block in work (WRITE)
        int16_t *dst = nullptr;      // its 64 coefficients, nullptr: not kept
        auto place = [&]() {
                if (!WRITE) return;
                dst = nullptr;
                if (n >= limit) return;
                const uint32_t u = n / (uint32_t) s.per_unit;
                const int k = s.blk_k[blk];
                const uint32_t uy = u / (uint32_t) s.row_units, ux = u - uy * (uint32_t) s.row_units;
                dst = s.coef[k] + ((size_t) (uy * s.nbv[k] + s.blk_by[blk]) * s.gw[k] + ux * s.nbh[k] + s.blk_bx[blk]) * 64;
        };
        place();
        while (p < stop_bits) {
                // 32 bits from bit p on, zero bits behind the end of the segment
                const uint32_t w0 = __builtin_bswap32(words[p >> 5]), w1 = __builtin_bswap32(words[(p >> 5) + 1]);
                uint32_t hi = (uint32_t) ((((unsigned long long) w0 << 32 | w1) << (p & 31)) >> 32);
                const uint32_t left = end_bits - p;
                if (left < 32) hi &= ~0u << (32 - left);
                const bool dc = z == 0;
                const uint16_t *lut = dc ? s.lut_dc + s.blk_dc[blk] * (1 << kDcLutBits) : s.lut_ac + s.blk_ac[blk] * (1 << kLutBits);
                const LongCodes &lc = s.longs[dc ? s.blk_dc[blk] : s.n_dc + s.blk_ac[blk]];
                const int bits = dc ? kDcLutBits : kLutBits;
                const unsigned e = lut[hi >> (32 - bits)];
                int l = (int) (e >> 8), sym = (int) (e & 0xff);
                if (e == 0) {
                        const unsigned pk = hi >> 16;
                        l = bits + 1;
                        for (int i = bits + 1; i <= 16; i++) l += pk >= lc.limit[i];
                        if (l > 16) {
                                l = 17;
                                sym = 0;
                        } else {
                                sym = lc.vals[(int) (pk >> (16 - l)) + lc.offset[l] & 0xff];
                        }
                }
                const int sz = sym & 15;
                const int v = (int) (((hi << l) >> 1) >> (31 - sz));
                const int value = v < ((1 << sz) >> 1) ? v + 1 - (1 << sz) : v;
                p += (uint32_t) (l + sz);
                if (dc) {
                        if (WRITE && dst) dst[0] = (int16_t) value;
                        z = 1;
                } else if (sz == 0 && sym != 0xF0) {
                        z = 64; // EOB
                } else {
                        z += sym >> 4;
                        if (WRITE && dst && sz && z < 64) dst[z] = (int16_t) value;
                        z++;
                }
                if (z >= 64) {
                        z = 0;
                        nb++;
                        blk = blk + 1 == s.per_unit ? 0 : blk + 1;
                        n++;
                        place();
                }
        }
}

struct SyncBuffers {
        unsigned long long *start, *exit; // per chunk: the state it was last decoded from, the state it left in
        uint32_t *nblk, *excl;            // blocks completed in the chunk; blocks completed in all the chunks in front of it (one more entry: the total)
        uint32_t *chunk_off;              // per segment: its first chunk (one more entry: the chunks in use)
        unsigned long long *wg_last;      // per workgroup: the exit state of its last chunk
        uint32_t *host;                   // mapped host memory: [0] a workgroup's last exit state changed, [1] every segment holds its blocks, [2] chunks in use
};
struct SyncGeom {
        int n_seg, seg_units, n_chunks; // segments the scan should have, units per segment, chunks the launches cover (an upper bound of those in use)
};

// chunks per segment -> first chunk of every segment (one workgroup, the segments in strides of 1024)
__global__ __launch_bounds__(1024) void sync_layout_kernel(const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end, const int *__restrict__ found, SyncGeom g,
                                                           SyncBuffers b)
{
        __shared__ uint32_t wave_tot[16];
        __shared__ uint32_t carry_s;
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int n_found = min(g.n_seg, found[0]);
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (int i0 = 0; i0 < g.n_seg; i0 += 1024) {
                const int i = i0 + tid;
                uint32_t v = 0;
                if (i < n_found) {
                        const uint32_t a = seg_start[i], e = seg_end[i];
                        v = e > a ? (8u * (e - a) + kSyncChunkBits - 1) / kSyncChunkBits : 0u;
                }
                uint32_t incl = v;
                for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += o;
                }
                if (lane == 63) wave_tot[wv] = incl;
                __syncthreads();
                uint32_t before = carry_s;
                for (int j = 0; j < wv; j++) before += wave_tot[j];
                if (i < g.n_seg) b.chunk_off[i] = before + incl - v;
                __syncthreads();
                if (tid == 1023) carry_s = before + incl;
                __syncthreads();
        }
        if (tid == 0) {
                b.chunk_off[g.n_seg] = carry_s;
                b.host[2] = carry_s;
        }
}

// chunk t -> its segment and its number inside it; false: not in use
__device__ __forceinline__ bool sync_locate(uint32_t t, const uint32_t *__restrict__ chunk_off, int n_seg, int &seg, uint32_t &j)
{
        if (t >= chunk_off[n_seg]) return false;
        int lo = 0, hi = n_seg; // the last segment whose first chunk is <= t (segments without chunks share their successor's first chunk: the last of them is the one with chunks)
        while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (chunk_off[mid] <= t) lo = mid;
                else hi = mid;
        }
        seg = lo;
        j = t - chunk_off[lo];
        return true;
}

// first = true: every chunk decodes from the guess "a block starts at my first bit" (for the first chunk of a segment that is the truth) and the workgroup settles its
// chunks among themselves.  first = false: the workgroup's first chunk takes the exit state of the workgroup in front (unless it starts a segment); if that is news, the
// workgroup settles again.  Run until no workgroup's last exit state changes (the host reads host[0]).
// A workgroup owns kSyncOwn chunks and, in the first pass, runs kSyncWarm lanes AHEAD of them over the last chunks of the workgroup in front: their only purpose is to
// hand the first owned chunk a start state that is in step already -- getting the bit position right takes a few symbols, getting the block of the unit right (the
// tables depend on it) several chunks.  With that the round across the workgroups finds nothing to change in most streams and costs microseconds; without, it cost as
// much as the first pass (every workgroup settled twice).
__global__ __launch_bounds__(kSyncWG) void sync_settle_kernel(const uint8_t *__restrict__ clean, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end,
                                                             ScanDev sp, const HuffDev *__restrict__ tabs, SyncGeom g, SyncBuffers b, int first)
{
        extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
        __shared__ SyncLds s;
        __shared__ unsigned long long s_exit[kSyncWG];
        sync_setup(lds, &s, sp, tabs);
        const int tid = threadIdx.x;
        const long tl = (long) blockIdx.x * kSyncOwn + tid - kSyncWarm; // lanes 0 .. kSyncWarm - 1: the chunks in front of the workgroup's own
        const uint32_t t = (uint32_t) (tl < 0 ? 0 : tl);
        const bool own = tid >= kSyncWarm;
        int seg = 0;
        uint32_t j = 0;
        // (the launch covers an upper bound: chunks behind the last one in use hold nothing and take no part; nor do the lanes ahead once the first pass is through)
        const bool live = tl >= 0 && (own || first) && sync_locate(t, b.chunk_off, g.n_seg, seg, j);
        const bool head = j == 0;                                       // the first chunk of its segment: its start state is known
        const uint32_t end_bits = live ? 8u * seg_end[seg] : 0u, c0 = live ? 8u * seg_start[seg] + j * (uint32_t) kSyncChunkBits : 0u;
        const uint32_t stop = min(c0 + (uint32_t) kSyncChunkBits, end_bits);
        unsigned long long st, ex = 0;
        uint32_t nb = 0;
        bool redo;
        if (first) {
                st = ex = sync_pack(c0, 0, 0);
                redo = live;
        } else {
                st = ex = 0;
                if (own) { st = b.start[t]; ex = b.exit[t]; nb = b.nblk[t]; }
                redo = false;
                if (tid == kSyncWarm && blockIdx.x > 0 && live && !head) {
                        const unsigned long long in = __atomic_load_n(b.wg_last + blockIdx.x - 1, __ATOMIC_RELAXED);
                        redo = in != st;
                        st = in;
                }
                if (!__syncthreads_or(redo)) return; // (uniform) the workgroup in front left where this one started from: nothing changes here
        }
        const unsigned long long last_before = first ? ~0ull : b.wg_last[blockIdx.x];
        for (int it = 0; it <= kSyncWG; it++) {
                if (redo) {
                        uint32_t p = (uint32_t) (st >> 16);
                        int blk = (int) (st >> 8 & 0xff), z = (int) (st & 0xff);
                        nb = 0;
                        sync_decode<false>(clean, end_bits, stop, p, blk, z, nb, s, 0, 0);
                        ex = sync_pack(p, blk, z);
                }
                s_exit[tid] = ex;
                __syncthreads();
                // (the first lane has nobody to its left; nor has the first owned one once the lanes ahead are out of the game: it got its state from the workgroup in front)
                const unsigned long long in = tid == 0 || !live || head || (!first && tid == kSyncWarm) ? st : s_exit[tid - 1];
                redo = in != st;
                st = in;
                if (!__syncthreads_or(redo)) break;
        }
        if (own) { b.start[t] = st; b.exit[t] = ex; b.nblk[t] = nb; }
        if (tid == kSyncWG - 1 && ex != last_before) {
                __atomic_store_n(b.wg_last + blockIdx.x, ex, __ATOMIC_RELAXED);
                if (!first) b.host[0] = 1;
        }
}

// blocks completed in front of every chunk; then: does every segment hold the blocks its units need?  (one workgroup, strides of 1024)
__global__ __launch_bounds__(1024) void sync_prefix_kernel(SyncBuffers b, SyncGeom g, int units, int per_unit, const int *__restrict__ found)
{
        __shared__ uint32_t wave_tot[16];
        __shared__ uint32_t carry_s;
        __shared__ int bad;
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        if (tid == 0) { carry_s = 0; bad = found[0] < g.n_seg; } // (a segment the stream does not have is an empty one to the sequential walk: zero bits)
        __syncthreads();
        for (int i0 = 0; i0 < g.n_chunks; i0 += 1024) {
                const int i = i0 + tid;
                const uint32_t v = i < g.n_chunks ? b.nblk[i] : 0;
                uint32_t incl = v;
                for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += o;
                }
                if (lane == 63) wave_tot[wv] = incl;
                __syncthreads();
                uint32_t before = carry_s;
                for (int j = 0; j < wv; j++) before += wave_tot[j];
                if (i < g.n_chunks) b.excl[i] = before + incl - v;
                __syncthreads();
                if (tid == 1023) carry_s = before + incl;
                __syncthreads();
        }
        if (tid == 0) b.excl[g.n_chunks] = carry_s;
        __threadfence();
        __syncthreads();
        for (int sg = tid; sg < g.n_seg; sg += 1024) {
                const int need = min(g.seg_units, units - sg * g.seg_units) * per_unit;
                const uint32_t have = b.excl[b.chunk_off[sg + 1]] - b.excl[b.chunk_off[sg]];
                if (have < (uint32_t) need) bad = 1;
        }
        __syncthreads();
        if (tid == 0) b.host[1] = !bad;
}

// every chunk once more, from its settled state, coefficients written
__global__ __launch_bounds__(kSyncWG) void sync_write_kernel(const uint8_t *__restrict__ clean, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end,
                                                            ScanDev sp, const HuffDev *__restrict__ tabs, SyncGeom g, SyncBuffers b)
{
        extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
        __shared__ SyncLds s;
        sync_setup(lds, &s, sp, tabs);
        const uint32_t t = blockIdx.x * kSyncWG + threadIdx.x;
        int seg = 0;
        uint32_t j = 0;
        if (!sync_locate(t, b.chunk_off, g.n_seg, seg, j)) return;
        const uint32_t end_bits = 8u * seg_end[seg];
        const uint32_t stop = min(8u * seg_start[seg] + (j + 1) * (uint32_t) kSyncChunkBits, end_bits);
        const unsigned long long st = b.start[t];
        uint32_t p = (uint32_t) (st >> 16), nb = 0;
        int blk = (int) (st >> 8 & 0xff), z = (int) (st & 0xff);
        const uint32_t seg_first = (uint32_t) seg * (uint32_t) g.seg_units * (uint32_t) s.per_unit;
        const uint32_t limit = (uint32_t) min((long) (seg + 1) * g.seg_units, (long) sp.units) * (uint32_t) s.per_unit; // blocks behind the segment's last unit are nobody's
        sync_decode<true>(clean, end_bits, stop, p, blk, z, nb, s, seg_first + b.excl[t] - b.excl[b.chunk_off[seg]], limit);
}

// DC differences -> DC values: a running sum per component over its blocks in scan order (unit after unit, the blocks of a unit row by row), started anew with every
// segment; blockIdx.x = component of the scan
__global__ __launch_bounds__(1024) void sync_dc_kernel(ScanDev sp, int seg_units)
{
        constexpr int kPer = 8; // consecutive blocks per lane: summed in registers, so that the workgroup-wide scan (barriers, shuffles) runs once per 8 192 blocks
        __shared__ int wave_v[16], wave_f[16];
        __shared__ int carry_s;
        const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int per = sp.nbh[k] * sp.nbv[k], row_units = sp.single ? sp.bw1 : sp.mcu_w;
        const long total = (long) sp.units * per;
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (long i0 = 0; i0 < total; i0 += 1024 * kPer) {
                int16_t *at[kPer];
                int fl[kPer], val[kPer];
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        const long i = i0 + (long) tid * kPer + e;
                        at[e] = nullptr;
                        fl[e] = 0; // 1: the sum starts anew here (the first block of the component in a segment)
                        if (i < total) {
                                const int u = (int) (i / per), j = (int) (i - (long) u * per);
                                const int by = j / sp.nbh[k], bx = j - by * sp.nbh[k];
                                const int uy = u / row_units, ux = u - uy * row_units;
                                at[e] = sp.coef[k] + ((size_t) (uy * sp.nbv[k] + by) * sp.gw[k] + ux * sp.nbh[k] + bx) * 64;
                                fl[e] = j == 0 && u % seg_units == 0;
                        }
                }
#pragma unroll
                for (int e = 0; e < kPer; e++) val[e] = at[e] ? (int) *at[e] : 0;
                // the lane's own blocks: running sums since the lane's first block or the last restart inside it
                int f = 0, v = 0;
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        v = fl[e] ? val[e] : v + val[e];
                        f |= fl[e];
                        val[e] = v;
                        fl[e] = f; // from here on: a restart lies at or in front of block e inside the lane
                }
                // segmented inclusive scan of the lanes' totals: (f, v) o (f', v') = (f | f', f' ? v' : v + v')
                for (int d = 1; d < 64; d <<= 1) {
                        const int of = __shfl_up(f, d, 64), ov = __shfl_up(v, d, 64);
                        if (lane >= d) {
                                if (!f) v += ov;
                                f |= of;
                        }
                }
                if (lane == 63) { wave_v[wv] = v; wave_f[wv] = f; }
                int pf = __shfl_up(f, 1, 64), pv = __shfl_up(v, 1, 64); // what the lanes in front of this one (in its wave) add up to
                if (lane == 0) { pf = 0; pv = 0; }
                __syncthreads();
                int cv = carry_s;
                for (int w = 0; w < wv; w++) cv = wave_f[w] ? wave_v[w] : cv + wave_v[w];
                const int before = pf ? pv : cv + pv; // the running sum in front of the lane's first block
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        if (at[e]) *at[e] = (int16_t) (fl[e] ? val[e] : before + val[e]);
                }
                __syncthreads();
                if (tid == 1023) carry_s = fl[kPer - 1] ? val[kPer - 1] : before + val[kPer - 1];
                __syncthreads();
        }
}

// jidctint.c; see oracle/jpeg_decode_oracle.c
#define CONST_BITS 13
#define PASS1_BITS 2
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ void idct_1d(const int (&s)[8], int (&o)[8])
{
        int z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
        z2 = s[2]; z3 = s[6];
        z1 = (z2 + z3) * 4433;
        tmp2 = z1 + z3 * (-15137);
        tmp3 = z1 + z2 * 6270;
        z2 = s[0]; z3 = s[4];
        tmp0 = (z2 + z3) * (1 << CONST_BITS);
        tmp1 = (z2 - z3) * (1 << CONST_BITS);
        tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
        tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
        z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3;
        o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
        o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1;
        o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

// one lane per block of one component's (MCU-padded) block grid
struct IdctJob {
        const int16_t *coef[3]; // zigzag order, as the Huffman kernel leaves them
        const uint16_t *qt[3];  // 64, natural order
        uint8_t *plane[3];
        int gw[3], pitch[3];
        long n_blocks[3];
};
// blockIdx.y = component
__global__ __launch_bounds__(256) void idct_kernel(IdctJob job)
{
        const int comp = blockIdx.y;
        const long b = (long) blockIdx.x * 256 + threadIdx.x;
        if (b >= job.n_blocks[comp]) return;
        const int16_t *__restrict__ coef = job.coef[comp];
        const uint16_t *__restrict__ qt = job.qt[comp];
        uint8_t *__restrict__ plane = job.plane[comp];
        const int gw = job.gw[comp], pitch = job.pitch[comp];
        const uint4 *src = (const uint4 *) (coef + b * 64);
        int v[64];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const uint4 q = src[i];
                const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                for (int k = 0; k < 4; k++) { // coefficient z of the zigzag sequence belongs at kZigzag[z]: indices known at compile time
                        const int n0 = kZigzagDev[8 * i + 2 * k], n1 = kZigzagDev[8 * i + 2 * k + 1];
                        v[n0] = (int) (int16_t) (w[k] & 0xffff) * (int) qt[n0];
                        v[n1] = (int) (int16_t) (w[k] >> 16) * (int) qt[n1];
                }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) { // pass 1: columns
                const int s[8] = { v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c] };
                int o[8];
                idct_1d(s, o);
#pragma unroll
                for (int k = 0; k < 8; k++) v[8 * k + c] = descale(o[k], CONST_BITS - PASS1_BITS);
        }
        const long by = b / gw, bx = b - by * gw;
        uint8_t *dst = plane + by * 8 * pitch + bx * 8;
#pragma unroll
        for (int r = 0; r < 8; r++) { // pass 2: rows
                const int s[8] = { v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7] };
                int o[8];
                idct_1d(s, o);
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                        const int x = descale(o[k], CONST_BITS + PASS1_BITS + 3) + 128;
                        const uint32_t px = (uint32_t) (x < 0 ? 0 : (x > 255 ? 255 : x));
                        if (k < 4) lo |= px << (8 * k);
                        else hi |= px << (8 * (k - 4));
                }
                *(uint2 *) (dst + (long) r * pitch) = make_uint2(lo, hi);
        }
}

// R, G, B planes -> packed RGB / RGBA (shifts as decoder_t has them)
__global__ void planar_rgb_pack_kernel(const uint8_t *__restrict__ r, const uint8_t *__restrict__ g, const uint8_t *__restrict__ b, int ppitch, uint8_t *__restrict__ dst,
                                       int dpitch, int width, int height, int rgba, int rs, int gs, int bs)
{
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
        if (x >= width || y >= height) return;
        const uint32_t R = r[(long) y * ppitch + x], G = g[(long) y * ppitch + x], B = b[(long) y * ppitch + x];
        if (rgba) {
                ((uint32_t *) (dst + (long) y * dpitch))[x] = (0xFFFFFFFFu ^ (0xFFu << rs) ^ (0xFFu << gs) ^ (0xFFu << bs)) | R << rs | G << gs | B << bs;
        } else {
                uint8_t *d = dst + (long) y * dpitch + 3 * x;
                d[0] = (uint8_t) R, d[1] = (uint8_t) G, d[2] = (uint8_t) B;
        }
}
// 4:4:4 Y, Cb, Cr planes -> UYVY: chroma of a pixel pair = (a + b) / 2, as UltraGrid's own 4:4:4 -> 4:2:2 converters do (vc_copylineY416toUYVY)
__global__ void yuv444p_to_uyvy_kernel(const uint8_t *__restrict__ yp, const uint8_t *__restrict__ cbp, const uint8_t *__restrict__ crp, int ppitch, uint8_t *__restrict__ dst,
                                       int dpitch, int width, int height)
{
        const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y; // x = pixel pair
        if (2 * x >= width || y >= height) return;
        const long o = (long) y * ppitch + 2 * x;
        const int x1 = 2 * x + 1 < width ? 1 : 0;
        const uint32_t u = (cbp[o] + cbp[o + x1]) / 2, v = (crp[o] + crp[o + x1]) / 2;
        ((uint32_t *) (dst + (long) y * dpitch))[x] = u | (uint32_t) yp[o] << 8 | v << 16 | (uint32_t) yp[o + x1] << 24;
}

struct Decoder {
        // device workspace, grown on demand
        uint8_t *stream = nullptr;
        size_t stream_cap = 0;
        uint8_t *clean = nullptr;   // the entropy-coded data of one scan without stuffed zeros and restart markers
        size_t clean_cap = 0;
        uint32_t *seg_start = nullptr, *seg_end = nullptr; // per restart segment, in the clean stream
        size_t seg_start_cap = 0, seg_end_cap = 0;
        int *scan_counts = nullptr; // [0]: segments found, [2 + 2 i], [3 + 2 i]: bytes taken out of / restart markers in the i-th 4 KiB of the scan
        size_t scan_cap = 0;
        HuffDev *tabs = nullptr;     // 8 tables
        uint16_t *qt = nullptr;      // 4 x 64
        int16_t *coef[3] = { nullptr, nullptr, nullptr };
        uint8_t *plane[3] = { nullptr, nullptr, nullptr };
        size_t coef_cap[3] = { 0, 0, 0 }, plane_cap[3] = { 0, 0, 0 };
        uint8_t *tmp = nullptr;      // intermediate packed frame (UYVY or RGB) when the output needs a second conversion
        size_t tmp_cap = 0;
        // scans without restart intervals (pass 2b): per-chunk states and counts, the mapped words the host reads between the launches
        unsigned long long *sync_start = nullptr, *sync_exit = nullptr, *sync_wg_last = nullptr;
        uint32_t *sync_nblk = nullptr, *sync_base = nullptr, *sync_chunk_off = nullptr;
        size_t sync_start_cap = 0, sync_exit_cap = 0, sync_wg_cap = 0, sync_nblk_cap = 0, sync_base_cap = 0, sync_chunk_off_cap = 0;
        uint32_t *sync_host = nullptr, *sync_host_dev = nullptr;
        // pinned staging for the tables; they are uploaded when they differ from the last frame's
        void *pinned = nullptr;
        HuffHost dc_now[4], ac_now[4];
        uint16_t qt_now[4][64];
        bool tables_valid = false;
        Header hdr;
        int plane_pitch[3] = { 0, 0, 0 };
        hipEvent_t uploaded = nullptr; // the pinned staging area may be rewritten once this has happened
        bool upload_pending = false;
};

bool grow(void **p, size_t *cap, size_t need)
{
        if (*cap >= need) return true;
        if (*p) (void) hipFree(*p);
        *p = nullptr;
        *cap = 0;
        if (hipMalloc(p, need + 64) != hipSuccess) return false;
        *cap = need;
        return true;
}

} // namespace

extern "C" {

typedef struct ug_hip_jpeg_decoder ug_hip_jpeg_decoder;

void ug_hip_jpeg_decoder_destroy(ug_hip_jpeg_decoder *dec);

int ug_hip_jpeg_decoder_create(ug_hip_jpeg_decoder **out)
{
        if (!out) return UG_HIP_EINVAL;
        Decoder *d = new Decoder();
        hipError_t err = hipMalloc((void **) &d->tabs, 8 * sizeof(HuffDev));
        if (err == hipSuccess) err = hipMalloc((void **) &d->qt, 4 * 64 * sizeof(uint16_t));
        if (err == hipSuccess) err = hipHostMalloc(&d->pinned, 8 * sizeof(HuffDev) + sizeof d->qt_now, hipHostMallocDefault);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d->uploaded, hipEventDisableTiming);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_jpeg_decoder_create");
                ug_hip_jpeg_decoder_destroy((ug_hip_jpeg_decoder *) d); // whatever was allocated so far
                return UG_HIP_ERUNTIME;
        }
        *out = (ug_hip_jpeg_decoder *) d;
        return UG_HIP_SUCCESS;
}

void ug_hip_jpeg_decoder_destroy(ug_hip_jpeg_decoder *dec)
{
        Decoder *d = (Decoder *) dec;
        if (!d) return;
        for (void *p : { (void *) d->stream, (void *) d->clean, (void *) d->seg_start, (void *) d->seg_end, (void *) d->scan_counts, (void *) d->tabs, (void *) d->qt, (void *) d->coef[0], (void *) d->coef[1], (void *) d->coef[2],
                         (void *) d->plane[0], (void *) d->plane[1], (void *) d->plane[2], (void *) d->tmp, (void *) d->sync_start, (void *) d->sync_exit, (void *) d->sync_wg_last,
                         (void *) d->sync_nblk, (void *) d->sync_base, (void *) d->sync_chunk_off }) {
                if (p) (void) hipFree(p);
        }
        if (d->sync_host) (void) hipHostFree(d->sync_host);
        if (d->pinned) (void) hipHostFree(d->pinned);
        if (d->uploaded) (void) hipEventDestroy(d->uploaded);
        delete d;
}

int ug_hip_jpeg_read_info(const void *jpeg_host, size_t len, int *width, int *height, int *subsampling, int *is_rgb, int *restart_interval)
{
        Header h;
        const int rc = parse((const uint8_t *) jpeg_host, len, h, kHeadersOnly);
        if (rc) {
                ug::set_last_error_msg(rc == -2 ? "ug_hip_jpeg_read_info: truncated stream" : "ug_hip_jpeg_read_info: not a baseline JPEG stream this decoder takes");
                return UG_HIP_EUNSUPP;
        }
        if (width) *width = h.width;
        if (height) *height = h.height;
        if (subsampling) *subsampling = h.ncomp == 1 ? 400 : (h.hs[0] == 2 ? (h.vs[0] == 2 ? 420 : 422) : 444);
        if (is_rgb) *is_rgb = h.is_rgb();
        if (restart_interval) *restart_interval = h.ri;
        return UG_HIP_SUCCESS;
}

int ug_hip_jpeg_decoder_decode(ug_hip_jpeg_decoder *dec, const void *jpeg_host, size_t len, ug_pixfmt_t out, void *dst_dev, int dst_pitch, int rshift, int gshift,
                               int bshift, ug_hip_stream_t stream)
{
        return ug_hip_jpeg_decoder_decode_sized(dec, jpeg_host, len, 0, 0, out, dst_dev, dst_pitch, rshift, gshift, bshift, stream);
}

int ug_hip_jpeg_decoder_decode_sized(ug_hip_jpeg_decoder *dec, const void *jpeg_host, size_t len, int expect_width, int expect_height, ug_pixfmt_t out,
                                     void *dst_dev, int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream)
{
        Decoder *d = (Decoder *) dec;
        if (!d || !jpeg_host || (!dst_dev && out != UG_PF_NONE) || expect_width < 0 || expect_height < 0) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: bad arguments");
                return UG_HIP_EINVAL;
        }
        Header &h = d->hdr;
        h = Header();
        int prc = parse((const uint8_t *) jpeg_host, len, h, kHeadersOnly);
        // one scan that carries every component (what UltraGrid's senders emit): its restart markers are found on the GPU, the host reads the
        // headers only; streams with one scan per component are walked on the host
        const bool gpu_scan = prc == 0 && h.scans[0].ns == h.ncomp;
        if (prc == 0 && !gpu_scan) {
                h = Header();
                prc = parse((const uint8_t *) jpeg_host, len, h, kWalkScans);
        }
        if (prc) {
                ug::set_last_error_msg(prc == -2 ? "ug_hip_jpeg_decoder_decode: truncated stream" : "ug_hip_jpeg_decoder_decode: not a baseline JPEG stream this decoder takes");
                return UG_HIP_EUNSUPP;
        }
        if (len > 0xFFFFFFF0u) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: stream too long");
                return UG_HIP_EUNSUPP;
        }
        // the caller sized dst_dev for this picture: the headers as THIS parse read them must agree, whatever an earlier parse said
        if ((expect_width && h.width != expect_width) || (expect_height && h.height != expect_height)) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: the stream's picture size is not the size the destination was made for");
                return UG_HIP_EINVAL;
        }
        for (int c = 1; c < h.ncomp; c++) { // the sampling layouts the output stage knows: 4:4:4, 4:2:2, 4:2:0
                if (h.hs[c] != 1 || h.vs[c] != 1) {
                        ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: unsupported sampling factors");
                        return UG_HIP_EUNSUPP;
                }
        }
        hipStream_t st = (hipStream_t) stream;
        // ---- workspace ----
        struct ScanPlan {
                ScanDev sp;
                int n_seg;
        };
        ScanPlan plan[3];
        long gw[3], gh[3];
        for (int c = 0; c < h.ncomp; c++) {
                gw[c] = (long) h.mcu_w * h.hs[c];
                gh[c] = (long) h.mcu_h * h.vs[c];
        }
        if (gw[0] * gh[0] > 0x7FFFFFFF / 64 || h.scans.size() > 3) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: picture too large");
                return UG_HIP_EUNSUPP;
        }
        int max_seg = 1;
        for (size_t i = 0; i < h.scans.size(); i++) {
                const Scan &sc = h.scans[i];
                ScanDev &sp = plan[i].sp;
                sp = ScanDev();
                sp.ns = sc.ns;
                sp.single = sc.ns == 1 && h.ncomp > 1;
                sp.mcu_w = h.mcu_w;
                sp.ri = h.ri;
                for (int k = 0; k < sc.ns; k++) {
                        const int c = sc.comp[k];
                        sp.comp[k] = c;
                        sp.td[k] = sc.td[k];
                        sp.ta[k] = sc.ta[k];
                        sp.nbh[k] = sp.single ? 1 : h.hs[c];
                        sp.nbv[k] = sp.single ? 1 : h.vs[c];
                        sp.gw[k] = (int) gw[c];
                        // the distinct tables of the scan (usually two of each: luma, chroma)
                        int j = 0;
                        while (j < sp.n_dc && sp.dc_tab[j] != sc.td[k]) j++;
                        if (j == sp.n_dc) sp.dc_tab[sp.n_dc++] = sc.td[k];
                        sp.dc_slot[k] = j;
                        j = 0;
                        while (j < sp.n_ac && sp.ac_tab[j] != sc.ta[k]) j++;
                        if (j == sp.n_ac) sp.ac_tab[sp.n_ac++] = sc.ta[k];
                        sp.ac_slot[k] = j;
                }
                if (sp.single) { // a non-interleaved scan walks the component's own block grid, ceil(size / 8) blocks (T.81 A.2.2)
                        const int c = sc.comp[0];
                        sp.bw1 = ((h.width * h.hs[c] + h.hmax - 1) / h.hmax + 7) / 8;
                        const int bh1 = ((h.height * h.vs[c] + h.vmax - 1) / h.vmax + 7) / 8;
                        sp.units = sp.bw1 * bh1;
                } else {
                        sp.bw1 = 1;
                        sp.units = h.mcu_w * h.mcu_h;
                }
                // as many segments as the restart interval accounts for (a stream may carry fewer or more markers than it should)
                plan[i].n_seg = h.ri ? (sp.units + h.ri - 1) / h.ri : 1;
                max_seg = plan[i].n_seg > max_seg ? plan[i].n_seg : max_seg;
        }
        const unsigned max_grid = (unsigned) ((len + kScanChunk - 1) / kScanChunk) + 1;
        bool ok = grow((void **) &d->stream, &d->stream_cap, len + 32) && grow((void **) &d->clean, &d->clean_cap, len + 64) &&
                  grow((void **) &d->seg_start, &d->seg_start_cap, (size_t) max_seg * sizeof(uint32_t)) &&
                  grow((void **) &d->seg_end, &d->seg_end_cap, (size_t) max_seg * sizeof(uint32_t)) &&
                  grow((void **) &d->scan_counts, &d->scan_cap, (2 * (size_t) max_grid + 2) * sizeof(int));
        for (int c = 0; c < h.ncomp && ok; c++) {
                d->plane_pitch[c] = (int) (gw[c] * 8);
                ok = grow((void **) &d->coef[c], &d->coef_cap[c], (size_t) (gw[c] * gh[c]) * 128) && grow((void **) &d->plane[c], &d->plane_cap[c], (size_t) (gw[c] * gh[c]) * 64);
        }
        if (!ok) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: out of device memory");
                return UG_HIP_ERUNTIME;
        }
        // ---- tables (when they differ from the last frame's) and the stream -> device ----
        if (!d->tables_valid || memcmp(d->dc_now, h.dc, sizeof h.dc) != 0 || memcmp(d->ac_now, h.ac, sizeof h.ac) != 0 || memcmp(d->qt_now, h.qt, sizeof h.qt) != 0) {
                if (d->upload_pending) { // an earlier call's asynchronous copy reads the staging area: let it finish before it is rewritten
                        UG_HIP_TRY(hipEventSynchronize(d->uploaded));
                        d->upload_pending = false;
                }
                HuffDev *tabs_h = (HuffDev *) d->pinned;
                for (int t = 0; t < 4; t++) {
                        build_dev(h.dc[t], tabs_h[t], kDcLutBits);
                        build_dev(h.ac[t], tabs_h[4 + t], kLutBits);
                }
                uint16_t *qt_h = (uint16_t *) (tabs_h + 8);
                memcpy(qt_h, h.qt, sizeof h.qt);
                UG_HIP_TRY(hipMemcpyAsync(d->tabs, tabs_h, 8 * sizeof(HuffDev), hipMemcpyHostToDevice, st));
                UG_HIP_TRY(hipMemcpyAsync(d->qt, qt_h, sizeof h.qt, hipMemcpyHostToDevice, st));
                UG_HIP_TRY(hipEventRecord(d->uploaded, st));
                d->upload_pending = true;
                memcpy(d->dc_now, h.dc, sizeof h.dc);
                memcpy(d->ac_now, h.ac, sizeof h.ac);
                memcpy(d->qt_now, h.qt, sizeof h.qt);
                d->tables_valid = true;
        }
        UG_HIP_TRY(hipMemcpyAsync(d->stream, jpeg_host, len, hipMemcpyHostToDevice, st));
        if (!gpu_scan) { // a scan of one component leaves the padding blocks of the MCU grid untouched
                for (int c = 0; c < h.ncomp; c++) UG_HIP_TRY(hipMemsetAsync(d->coef[c], 0, (size_t) (gw[c] * gh[c]) * 128, st));
        }
        // ---- scan by scan: clean stream + segment table, then Huffman decoding ----
        for (size_t i = 0; i < h.scans.size(); i++) {
                const Scan &sc = h.scans[i];
                ScanDev &sp = plan[i].sp;
                for (int k = 0; k < sc.ns; k++) sp.coef[k] = d->coef[sc.comp[k]];
                const int n_seg = plan[i].n_seg;
                const size_t base = sc.data_begin & ~(size_t) 15;
                const unsigned grid = (unsigned) ((sc.data_end - base + kScanChunk - 1) / kScanChunk);
                if (grid == 0) continue; // no data at all: the planes stay as they are
                hipLaunchKernelGGL(clean_count_kernel, dim3(grid), dim3(kScanWG), 0, st, d->stream, base, sc.data_begin, sc.data_end, d->scan_counts + 2, d->seg_end,
                                   n_seg);
                hipLaunchKernelGGL(clean_place_kernel, dim3(grid), dim3(kScanWG), 0, st, d->stream, base, sc.data_begin, sc.data_end, d->scan_counts + 2, d->clean,
                                   d->seg_start, d->seg_end, n_seg, d->scan_counts);
                // Lanes per workgroup: a wave executes the same instructions whatever the number of its lanes in use, so the fewest waves that
                // still give every SIMD one is the fastest launch -- measured at 4K (8 100 segments): whole frame with 16 / 8 / 4 / 2 / 1 lanes per wave
                // 149 / 150 / 164 / 218 / 291 us (this kernel: 62 us at 8 lanes = 1 013 waves, 76 us at 4 = two waves per SIMD); more than 64 lanes' worth of segments per
                // SIMD (8K and up) simply fills the waves.  The stretch of the stream a workgroup stages is sized for twice the average segment.
                // UG_JPEG_DEC_LANES=<n> overrides (experiments).
                const size_t avg = (sc.data_end - sc.data_begin) / (size_t) n_seg + 1;
                auto stage_for = [&](int lanes) {
                        size_t stage = (2 * avg * (size_t) lanes + 256 + 255) & ~(size_t) 255;
                        return stage < 512 ? (size_t) 512 : (stage > kMaxStage ? kMaxStage : stage);
                };
                auto lds_for = [&](int lanes) { return (size_t) lds_tile_offset(sp.n_dc, sp.n_ac) + (size_t) lanes * kTileWords * 4 + stage_for(lanes) + 16; };
                int lanes = 8;
                while (lanes < 64 && (n_seg + lanes - 1) / lanes > 1024) lanes *= 2;
                static const int forced_lanes = getenv("UG_JPEG_DEC_LANES") ? atoi(getenv("UG_JPEG_DEC_LANES")) : 0;
                if (forced_lanes >= 1 && forced_lanes <= 64) lanes = forced_lanes;
                const size_t stage = stage_for(lanes);
                auto one_lane_per_segment = [&]() {
                        hipLaunchKernelGGL(huff_decode_kernel, dim3((unsigned) ((n_seg + lanes - 1) / lanes)), dim3(64), lds_for(lanes), st, d->clean, d->seg_start, d->seg_end, n_seg,
                                           d->scan_counts, lanes, (int) stage, sp, d->tabs);
                };
                // ---- a scan of LONG segments (no restart intervals, or few): self-synchronising parallel decoding (pass 2b) -- synchronises with the host between its launches ----
                static const bool sync_off = getenv("UG_JPEG_DEC_SYNC") != nullptr && getenv("UG_JPEG_DEC_SYNC")[0] == '0';
                int per_unit = 0;
                for (int k = 0; k < sp.ns; k++) per_unit += sp.nbh[k] * sp.nbv[k];
                const size_t scan_bytes = sc.data_end - sc.data_begin;
                // Which way is faster (profiles/r06_decode_no_restart.txt): a lane walks its segment at ~0.5 us per byte, and the segments run side by side; the parallel
                // decode costs ~0.55 ms of launches, host synchronisations and settling however small the picture, plus what grows with the picture (clearing the
                // planes, the DC sums: ~0.06 us per 1000 pixels), and hardly depends on the length of the stream.  4K: segments from ~2 KiB; 1080p: from ~1.3 KiB.
                const bool long_segments = 0.5 * (double) (scan_bytes / (size_t) n_seg) > 550.0 + 6e-5 * (double) h.width * (double) h.height;
                if (scan_bytes < kSyncMinBytes || !long_segments || scan_bytes >= ((size_t) 1 << 28) || per_unit > kSyncMaxUnitBlocks ||
                    (long) sp.units * per_unit >= (1L << 31) || sync_off) {
                        one_lane_per_segment();
                        continue;
                }
                // (an upper bound of the chunks: the clean stream is no longer than the scan, and every segment's last chunk may be a partial one)
                const size_t chunks_ub = (scan_bytes * 8 + kSyncChunkBits - 1) / kSyncChunkBits + (size_t) n_seg;
                const int n_wg = (int) ((chunks_ub + kSyncOwn - 1) / kSyncOwn), n_wg_write = (int) ((chunks_ub + kSyncWG - 1) / kSyncWG); // (the settling kernel's workgroups own kSyncOwn chunks each)
                const size_t padded = std::max((size_t) n_wg * kSyncOwn, (size_t) n_wg_write * kSyncWG);
                const int per_seg = sp.ri && sp.ri < sp.units ? sp.ri : sp.units; // units per segment (huff_decode_kernel's rule)
                const SyncGeom geom = { n_seg, per_seg, n_wg * kSyncOwn }; // (the chunks the settling kernel gives a block count)
                if (!d->sync_host) {
                        UG_HIP_TRY(hipHostMalloc((void **) &d->sync_host, 64, hipHostMallocMapped));
                        UG_HIP_TRY(hipHostGetDevicePointer((void **) &d->sync_host_dev, d->sync_host, 0));
                }
                if (!grow((void **) &d->sync_start, &d->sync_start_cap, padded * 8) || !grow((void **) &d->sync_exit, &d->sync_exit_cap, padded * 8) ||
                    !grow((void **) &d->sync_nblk, &d->sync_nblk_cap, padded * 4) || !grow((void **) &d->sync_base, &d->sync_base_cap, (padded + 1) * 4) ||
                    !grow((void **) &d->sync_chunk_off, &d->sync_chunk_off_cap, ((size_t) n_seg + 1) * 4) || !grow((void **) &d->sync_wg_last, &d->sync_wg_cap, (size_t) n_wg * 8)) {
                        ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: out of device memory");
                        return UG_HIP_ERUNTIME;
                }
                const SyncBuffers sb = { d->sync_start, d->sync_exit, d->sync_nblk, d->sync_base, d->sync_chunk_off, d->sync_wg_last, d->sync_host_dev };
                const size_t sync_lds = (size_t) lds_tile_offset(sp.n_dc, sp.n_ac);
                hipLaunchKernelGGL(sync_layout_kernel, dim3(1), dim3(1024), 0, st, d->seg_start, d->seg_end, d->scan_counts, geom, sb);
                hipLaunchKernelGGL(sync_settle_kernel, dim3((unsigned) n_wg), dim3(kSyncWG), sync_lds, st, d->clean, d->seg_start, d->seg_end, sp, d->tabs, geom, sb, 1);
                bool settled = n_wg == 1;
                for (int round = 0; round < n_wg + 1 && !settled; round++) { // (a change travels at least one workgroup per round: n_wg rounds at the very worst; usually one or two)
                        UG_HIP_TRY(hipStreamSynchronize(st)); // the earlier launch is through: the flag may be cleared
                        d->sync_host[0] = 0;
                        hipLaunchKernelGGL(sync_settle_kernel, dim3((unsigned) n_wg), dim3(kSyncWG), sync_lds, st, d->clean, d->seg_start, d->seg_end, sp, d->tabs, geom, sb, 0);
                        UG_HIP_TRY(hipStreamSynchronize(st));
                        settled = d->sync_host[0] == 0;
                }
                hipLaunchKernelGGL(sync_prefix_kernel, dim3(1), dim3(1024), 0, st, sb, geom, sp.units, per_unit, d->scan_counts);
                UG_HIP_TRY(hipStreamSynchronize(st));
                if (!settled || !d->sync_host[1]) { // a segment's data ends before its units do (or the states never settled): the sequential walk, zero-bit tail and all
                        one_lane_per_segment();
                        continue;
                }
                if (gpu_scan) { // (the other scans' planes were cleared above) the write pass stores the coefficients that are there, not the zeros between them
                        for (int k = 0; k < sc.ns; k++) UG_HIP_TRY(hipMemsetAsync(sp.coef[k], 0, (size_t) (gw[sc.comp[k]] * gh[sc.comp[k]]) * 128, st));
                }
                hipLaunchKernelGGL(sync_write_kernel, dim3((unsigned) n_wg_write), dim3(kSyncWG), sync_lds, st, d->clean, d->seg_start, d->seg_end, sp, d->tabs, geom, sb);
                hipLaunchKernelGGL(sync_dc_kernel, dim3((unsigned) sc.ns), dim3(1024), 0, st, sp, per_seg);
        }
        // ---- dequantisation + IDCT ----
        {
                IdctJob job = {};
                long most = 0;
                for (int c = 0; c < h.ncomp; c++) {
                        job.coef[c] = d->coef[c];
                        job.qt[c] = d->qt + 64 * h.tq[c];
                        job.plane[c] = d->plane[c];
                        job.gw[c] = (int) gw[c];
                        job.pitch[c] = d->plane_pitch[c];
                        job.n_blocks[c] = gw[c] * gh[c];
                        most = job.n_blocks[c] > most ? job.n_blocks[c] : most;
                }
                hipLaunchKernelGGL(idct_kernel, dim3((unsigned) ((most + 255) / 256), (unsigned) h.ncomp), dim3(256), 0, st, job);
        }
        UG_HIP_LAUNCH_CHECK();
        if (out == UG_PF_NONE) return UG_HIP_SUCCESS; // planes only (tests)
        // ---- planes -> output codec ----
        const int w = h.width, hh = h.height;
        const bool rgb = h.is_rgb();
        if (!dst_pitch) dst_pitch = ug::linesize(out, w);
        if (h.ncomp == 1) {
                // greyscale (GPUJPEG_U8, video_decompress/gpujpeg.c:239-241): a Y'CbCr picture whose chroma is nowhere off its zero -- two planes of 128 beside
                // the decoded one, then the 4:4:4 path (I420: the luma plane and 128s)
                if (out == UG_PF_I420) {
                        const int cw = (w + 1) / 2, ch = (hh + 1) / 2;
                        uint8_t *o = (uint8_t *) dst_dev;
                        UG_HIP_TRY(hipMemcpy2DAsync(o, w, d->plane[0], d->plane_pitch[0], w, hh, hipMemcpyDeviceToDevice, st));
                        UG_HIP_TRY(hipMemsetAsync(o + (size_t) w * hh, 128, 2 * (size_t) cw * ch, st));
                        return UG_HIP_SUCCESS;
                }
                const size_t bytes = (size_t) (gw[0] * gh[0]) * 64;
                for (int c = 1; c < 3; c++) {
                        if (!grow((void **) &d->plane[c], &d->plane_cap[c], bytes)) {
                                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: out of device memory");
                                return UG_HIP_ERUNTIME;
                        }
                        d->plane_pitch[c] = d->plane_pitch[0];
                        UG_HIP_TRY(hipMemsetAsync(d->plane[c], 128, bytes, st));
                }
        }
        auto need_tmp = [&](ug_pixfmt_t f) {
                return grow((void **) &d->tmp, &d->tmp_cap, (size_t) ug::linesize(f, w) * hh + 64);
        };
        if (rgb) {
                if (h.hs[0] != 1 || h.vs[0] != 1) {
                        ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: subsampled R,G,B streams are not supported");
                        return UG_HIP_EUNSUPP;
                }
                if (out == UG_PF_RGB || out == UG_PF_RGBA) {
                        hipLaunchKernelGGL(planar_rgb_pack_kernel, dim3((unsigned) ((w + 255) / 256), (unsigned) hh), dim3(256), 0, st, d->plane[0], d->plane[1], d->plane[2],
                                           d->plane_pitch[0], (uint8_t *) dst_dev, dst_pitch, w, hh, out == UG_PF_RGBA, rshift, gshift, bshift);
                        UG_HIP_LAUNCH_CHECK();
                        return UG_HIP_SUCCESS;
                }
                if (out == UG_PF_UYVY) { // through packed RGB and vc_copylineRGBtoUYVY's arithmetic
                        if (!need_tmp(UG_PF_RGB)) return UG_HIP_ERUNTIME;
                        hipLaunchKernelGGL(planar_rgb_pack_kernel, dim3((unsigned) ((w + 255) / 256), (unsigned) hh), dim3(256), 0, st, d->plane[0], d->plane[1], d->plane[2],
                                           d->plane_pitch[0], d->tmp, 3 * w, w, hh, 0, 0, 8, 16);
                        return ug_hip_pixfmt_convert(UG_PF_RGB, UG_PF_UYVY, d->tmp, dst_dev, w, hh, 0, dst_pitch, 0, 8, 16, stream);
                }
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: unsupported output codec");
                return UG_HIP_EUNSUPP;
        }
        // Y, Cb, Cr (coded as they came: BT.709 limited-range samples in UltraGrid's streams, gpujpeg.cpp:303-305)
        const int sub = h.hs[0] == 2 ? (h.vs[0] == 2 ? 420 : 422) : (h.vs[0] == 1 ? 444 : 0);
        if (!sub) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: unsupported sampling factors");
                return UG_HIP_EUNSUPP;
        }
        if (out == UG_PF_I420) {
                if (sub != 420) {
                        ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: I420 output needs a 4:2:0 stream");
                        return UG_HIP_EUNSUPP;
                }
                const int cw = (w + 1) / 2, ch = (hh + 1) / 2;
                uint8_t *o = (uint8_t *) dst_dev;
                UG_HIP_TRY(hipMemcpy2DAsync(o, w, d->plane[0], d->plane_pitch[0], w, hh, hipMemcpyDeviceToDevice, st));
                UG_HIP_TRY(hipMemcpy2DAsync(o + (size_t) w * hh, cw, d->plane[1], d->plane_pitch[1], cw, ch, hipMemcpyDeviceToDevice, st));
                UG_HIP_TRY(hipMemcpy2DAsync(o + (size_t) w * hh + (size_t) cw * ch, cw, d->plane[2], d->plane_pitch[2], cw, ch, hipMemcpyDeviceToDevice, st));
                return UG_HIP_SUCCESS;
        }
        if (out != UG_PF_UYVY && out != UG_PF_RGB && out != UG_PF_RGBA) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: unsupported output codec");
                return UG_HIP_EUNSUPP;
        }
        uint8_t *uyvy = (uint8_t *) dst_dev;
        int uyvy_pitch = dst_pitch;
        if (out != UG_PF_UYVY) {
                if (!need_tmp(UG_PF_UYVY)) return UG_HIP_ERUNTIME;
                uyvy = d->tmp;
                uyvy_pitch = ug::linesize(UG_PF_UYVY, w);
        }
        int rc;
        if (sub == 422) {
                rc = ug_hip_yuv422p_to_uyvy(d->plane[0], d->plane_pitch[0], d->plane[1], d->plane_pitch[1], d->plane[2], d->plane_pitch[2], uyvy, uyvy_pitch, w, hh, stream);
        } else if (sub == 420) {
                rc = ug_hip_yuv420p_to_uyvy(d->plane[0], d->plane_pitch[0], d->plane[1], d->plane_pitch[1], d->plane[2], d->plane_pitch[2], uyvy, uyvy_pitch, w, hh, stream);
        } else {
                hipLaunchKernelGGL(yuv444p_to_uyvy_kernel, dim3((unsigned) (((w + 1) / 2 + 255) / 256), (unsigned) hh), dim3(256), 0, st, d->plane[0], d->plane[1], d->plane[2],
                                   d->plane_pitch[0], uyvy, uyvy_pitch, w, hh);
                rc = UG_HIP_SUCCESS;
        }
        if (rc != UG_HIP_SUCCESS || out == UG_PF_UYVY) return rc;
        return ug_hip_pixfmt_convert(UG_PF_UYVY, out, uyvy, dst_dev, w, hh, uyvy_pitch, dst_pitch, rshift, gshift, bshift, stream);
}

// the component planes of the last decode (device memory, MCU-padded): for tests and for callers that want planar output
int ug_hip_jpeg_decoder_plane(const ug_hip_jpeg_decoder *dec, int component, const void **plane_dev, int *pitch, int *width, int *height)
{
        const Decoder *d = (const Decoder *) dec;
        if (!d || component < 0 || component >= d->hdr.ncomp || !plane_dev) return UG_HIP_EINVAL;
        *plane_dev = d->plane[component];
        if (pitch) *pitch = d->plane_pitch[component];
        if (width) *width = (d->hdr.width * d->hdr.hs[component] + d->hdr.hmax - 1) / d->hdr.hmax;
        if (height) *height = (d->hdr.height * d->hdr.vs[component] + d->hdr.vmax - 1) / d->hdr.vmax;
        return UG_HIP_SUCCESS;
}

} // extern "C"
