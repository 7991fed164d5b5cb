// jpeg_decode.hip -- baseline JPEG decoder on gfx950: the receive side of the JPEG path (the rows behind
// src/video_decompress/gpujpeg.c:74-140,292-301, which hands the work to the external libgpujpeg).
//
//   host   marker syntax (T.81 B.2): DQT, SOF0, DHT, DRI, SOS, Adobe APP14; the restart segments of every scan are located by their RSTn
//          markers (E.2.4) -- restart intervals are what makes the entropy-coded data parallel;
//   GPU 1  Huffman decoding (F.2.2), one lane per restart segment: 9-bit look-up for the short codes, the canonical MAXCODE walk for the
//          long ones, byte stuffing removed on the fly; quantised coefficients are collected per block in LDS and written out 128 bytes at a time;
//   GPU 2  dequantisation + inverse DCT, one lane per 8x8 block: libjpeg's jidctint ("slow but accurate integer": Loeffler-Ligtenberg-
//          Moschytz, 13-bit constants, PASS1_BITS 2) -- integer arithmetic, so the component planes equal libjpeg's bit for bit;
//   GPU 3  planes -> the output codec with the pixel-format kernels the library already has (planar 4:2:2 / 4:2:0 -> UYVY as
//          from_planar.c does it, UYVY -> RGB / RGBA with pixfmt_conv.c's arithmetic), R,G,B planes packed directly.
// Bit-identical to oracle/jpeg_decode_oracle.c, which is pinned to libjpeg-turbo (tests/test_oracle_jpeg_decode.py).
#include <string.h>

#include <vector>

#include "ug_common.h"

namespace {

constexpr int kLutBits = 9;

struct HuffHost {
        uint8_t bits[17] = {};
        uint8_t vals[256] = {};
        bool present = false;
};
// device form of one Huffman table
struct HuffDev {
        uint16_t lut[1 << kLutBits]; // (length << 8) | symbol for codes of at most kLutBits bits, 0 = longer
        int maxcode[18];             // per length; -1 = no code of that length
        int mincode[17];
        int valptr[17];
        uint8_t vals[256];
};

void build_dev(const HuffHost &h, HuffDev &d)
{
        memset(&d, 0, sizeof d);
        for (int l = 0; l < 18; l++) d.maxcode[l] = -1;
        if (!h.present) return;
        memcpy(d.vals, h.vals, sizeof d.vals);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
                d.valptr[l] = k;
                d.mincode[l] = code;
                for (int i = 0; i < h.bits[l]; i++, k++, code++) {
                        if (l <= kLutBits) {
                                const int lo = code << (kLutBits - l);
                                for (int f = 0; f < (1 << (kLutBits - l)); f++) d.lut[lo + f] = (uint16_t) (l << 8 | h.vals[k]);
                        }
                }
                d.maxcode[l] = h.bits[l] ? code - 1 : -1;
                code <<= 1;
        }
        d.maxcode[17] = 0x7fffffff;
}

struct Scan {
        int ns, comp[3], td[3], ta[3];
        size_t data_begin, data_end; // entropy-coded bytes [begin, end) in the stream (end = the marker that follows)
        std::vector<uint32_t> seg_off; // start of every restart segment
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
        kHeadersOnly, // stop behind the first SOS header: everything a single-scan stream needs from the host (its restart markers are found on the GPU)
        kWalkScans,   // walk the entropy-coded data of every scan on the host and record where each restart segment starts
};

// 0 ok, -1 not a baseline stream this decoder takes, -2 truncated
int parse(const uint8_t *data, size_t len, Header &h, ParseMode mode)
{
        const bool want_segments = mode == kWalkScans;
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
                                for (int k = 0; k < 64; k++) h.qt[t][kZigzagHost[k]] = s[o + 1 + k];
                        }
                } else if (m == 0xC0) {
                        if (seglen < 8 || s[0] != 8) return -1;
                        h.height = s[1] << 8 | s[2];
                        h.width = s[3] << 8 | s[4];
                        h.ncomp = s[5];
                        if ((h.ncomp != 1 && h.ncomp != 3) || seglen < (size_t) 8 + 3 * h.ncomp || !h.width || !h.height) return -1;
                        for (int c = 0; c < h.ncomp; c++) {
                                h.cid[c] = s[6 + 3 * c];
                                h.hs[c] = s[7 + 3 * c] >> 4;
                                h.vs[c] = s[7 + 3 * c] & 15;
                                h.tq[c] = s[8 + 3 * c];
                                if (h.hs[c] < 1 || h.hs[c] > 2 || h.vs[c] < 1 || h.vs[c] > 2 || h.tq[c] > 3) return -1;
                                h.hmax = h.hs[c] > h.hmax ? h.hs[c] : h.hmax;
                                h.vmax = h.vs[c] > h.vmax ? h.vs[c] : h.vmax;
                        }
                        h.mcu_w = (h.width + 8 * h.hmax - 1) / (8 * h.hmax);
                        h.mcu_h = (h.height + 8 * h.vmax - 1) / (8 * h.vmax);
                } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                        return -1; // extended / progressive / lossless / arithmetic: not baseline
                } else if (m == 0xC4) {
                        for (size_t o = 0; o + 17 <= seglen - 2;) {
                                const int tc = s[o] >> 4, th = s[o] & 15;
                                if (th > 3 || tc > 1) return -1;
                                HuffHost &t = tc ? h.ac[th] : h.dc[th];
                                int n = 0;
                                t.bits[0] = 0;
                                for (int l = 1; l <= 16; l++) n += (t.bits[l] = s[o + l]);
                                if (n > 256 || o + 17 + n > seglen - 2) return -1;
                                memset(t.vals, 0, sizeof t.vals);
                                memcpy(t.vals, s + o + 17, (size_t) n);
                                t.present = true;
                                o += 17 + (size_t) n;
                        }
                } else if (m == 0xDD) {
                        h.ri = s[0] << 8 | s[1];
                } else if (m == 0xEE && seglen >= 14 && memcmp(s, "Adobe", 5) == 0) {
                        h.adobe = s[11];
                } else if (m == 0xDA) {
                        if (!h.width) return -1;
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
                        if (want_segments) sc.seg_off.push_back((uint32_t) q);
                        for (;;) {
                                const uint8_t *f = (const uint8_t *) memchr(data + q, 0xFF, len - q);
                                if (!f || f + 1 >= data + len) { q = len; break; }
                                q = (size_t) (f - data);
                                const int n = data[q + 1];
                                if (n == 0x00 || n == 0xFF) { q += n == 0 ? 2 : 1; continue; }
                                if (n >= 0xD0 && n <= 0xD7) {
                                        q += 2;
                                        if (want_segments) sc.seg_off.push_back((uint32_t) q);
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
// ---- restart markers located on the GPU (single-scan streams) ---------------------------------------------------------------------
// Inside entropy-coded data 0xFF is always followed by 0x00 (stuffing), another 0xFF (fill) or a marker's second byte, so a byte pair
// FF D0..D7 is a restart marker wherever it stands.  Pass 1 counts the markers of every 4 KiB of the stream, pass 2 turns the counts into
// positions in stream order: seg_off[0] = first byte of the scan, seg_off[1 + i] = the byte behind the i-th marker.
constexpr int kScanWG = 256, kScanBytesPerLane = 16, kScanChunk = kScanWG * kScanBytesPerLane;

// bit i set: bytes pos+i, pos+i+1 are a restart marker that lies in [begin, len); pos is 16-byte aligned
__device__ __forceinline__ uint32_t rst_mask(const uint8_t *__restrict__ stream, size_t pos, size_t begin, size_t len)
{
        if (pos >= len) return 0;
        const uint4 q = *(const uint4 *) (stream + pos);
        const uint32_t w[5] = { q.x, q.y, q.z, q.w, stream[pos + 16] }; // the buffer is allocated with slack behind len
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
                const uint32_t b0 = (w[i / 4] >> (8 * (i % 4))) & 0xff, b1 = (w[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xff;
                const bool hit = b0 == 0xFF && (b1 & 0xF8) == 0xD0 && pos + i >= begin && pos + i + 1 < len;
                m |= (uint32_t) hit << i;
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

__global__ __launch_bounds__(kScanWG) void rst_count_kernel(const uint8_t *__restrict__ stream, size_t base, size_t begin, size_t len, int *__restrict__ counts)
{
        __shared__ int part[4];
        const size_t pos = base + (size_t) blockIdx.x * kScanChunk + threadIdx.x * kScanBytesPerLane;
        const int t = wg_sum_256(__popc(rst_mask(stream, pos, begin, len)), part);
        if (threadIdx.x == 0) counts[blockIdx.x] = t;
}

// found[0] = number of segments located (1 + markers); at most cap offsets are written
__global__ __launch_bounds__(kScanWG) void rst_place_kernel(const uint8_t *__restrict__ stream, size_t base, size_t begin, size_t len, const int *__restrict__ counts,
                                                            uint32_t *__restrict__ seg_off, int cap, int *__restrict__ found)
{
        __shared__ int part[4];
        __shared__ int wave_tot[4];
        int before = 0;
        for (int j = threadIdx.x; j < (int) blockIdx.x; j += kScanWG) before += counts[j];
        before = wg_sum_256(before, part);
        const size_t pos = base + (size_t) blockIdx.x * kScanChunk + threadIdx.x * kScanBytesPerLane;
        uint32_t m = rst_mask(stream, pos, begin, len);
        const int mine = __popc(m);
        int incl = mine; // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(incl, o);
                if ((int) (threadIdx.x & 63) >= o) incl += n;
        }
        if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        int idx = 1 + before + incl - mine;
        for (int w = 0; w < (int) (threadIdx.x >> 6); w++) idx += wave_tot[w];
        while (m) {
                const int i = __ffs(m) - 1;
                m &= m - 1;
                if (idx < cap) seg_off[idx] = (uint32_t) (pos + i + 2);
                idx++;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) seg_off[0] = (uint32_t) begin;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanWG - 1) found[0] = idx;
}

// ---- Huffman decoding ----------------------------------------------------------------------------------------------------------------
struct ScanDev {
        int ns, comp[3], td[3], ta[3], nbh[3], nbv[3], gw[3]; // blocks per unit and blocks per row of each component's grid
        int single, bw1, mcu_w, ri;
        long units;
        int16_t *coef[3];
        unsigned scan_end;
};

// what the codes longer than the look-up need, in LDS
struct LongCodes {
        uint32_t limit[17]; // (MAXCODE[n] + 1) << (16 - n), carried over lengths without codes
        int16_t offset[17]; // VALPTR[n] - MINCODE[n]
        uint8_t vals[256];
};

// The entropy-coded bytes of one segment.  The bit window is topped up 32 bits at a time from a word that was loaded one refill earlier
// (the common case: four bytes, none of them 0xFF); a word with 0xFF in it, or the last bytes of the data, go through the byte-wise path
// that removes the stuffing and stops at a marker.  Behind a marker or the end of the data the window fills with zero bits.
struct BitReader {
        const uint8_t *base;     // the stream (the same for every lane)
        uint32_t pos, end;       // next unread byte, end of the data
        uint32_t nxt;            // the four bytes at pos, loaded ahead
        bool over;
        unsigned long long acc;  // bits are consumed from the top
        int cnt;
        __device__ __forceinline__ uint32_t load32(uint32_t at) const
        {
                uint32_t w;
                __builtin_memcpy(&w, base + at, 4); // the stream buffer has slack behind the data
                return w;
        }
        __device__ __forceinline__ void open(const uint8_t *stream, uint32_t begin, uint32_t finish)
        {
                base = stream;
                pos = begin;
                end = finish;
                over = begin >= finish;
                acc = 0;
                cnt = 0;
                nxt = load32(pos);
        }
        __device__ __forceinline__ void refill()
        {
                if (cnt > 32) return; // a code (<= 16 bits) and its extra bits (<= 15) fit in what is left
                do {
                        if (over) break;
                        uint32_t w = nxt;
                        const bool has_ff = ((~w - 0x01010101u) & w & 0x80808080u) != 0;
                        if (pos + 4 <= end && !has_ff) {
                                acc |= (unsigned long long) __builtin_bswap32(w) << (32 - cnt);
                                cnt += 32;
                                pos += 4;
                        } else { // byte by byte, out of the same register
                                int avail = end - pos < 4 ? (int) (end - pos) : 4;
                                while (avail > 0) {
                                        const uint32_t byte = w & 0xff;
                                        int used = 1;
                                        if (byte == 0xFF) {
                                                if (avail < 2) {
                                                        if (pos + 1 >= end) over = true; // the data ends in 0xFF
                                                        break;                            // else: the byte behind it is in the next word
                                                }
                                                if ((w >> 8 & 0xff) != 0) { // a marker: the segment is over
                                                        over = true;
                                                        break;
                                                }
                                                used = 2; // a stuffed zero: the data byte is 0xFF
                                        }
                                        acc |= (unsigned long long) byte << (56 - cnt);
                                        cnt += 8;
                                        w >>= 8 * used;
                                        avail -= used;
                                        pos += used;
                                }
                                if (pos >= end) over = true;
                        }
                        nxt = load32(pos);
                } while (cnt <= 32);
                if (over && cnt <= 32) cnt = 64; // zero bits from here on (the window's low bits are zero already)
        }
};

// One Huffman symbol and the `symbol & 15` extra bits behind it (F.2.2.1, sign extension of Figure F.12): returns the symbol, the extended
// value in `value`.  Works on the top 32 bits of the window: code (<= 16) + extra bits (<= 15) always fit.
__device__ __forceinline__ int decode_symbol(BitReader &br, const LongCodes &lc, const uint16_t *lut, int &value)
{
        const uint32_t hi = (uint32_t) (br.acc >> 32);
        const unsigned e = lut[hi >> (32 - kLutBits)];
        int l = (int) (e >> 8), sym = (int) (e & 0xff);
        if (__builtin_expect(e == 0, 0)) {
                // a code longer than the look-up covers: codes of length n fill [.., limit[n]) of the 16-bit prefixes, limits ascending (F.2.2.3's
                // MAXCODE walk without the loop)
                const unsigned pk = hi >> 16;
                l = kLutBits + 1;
#pragma unroll
                for (int n = kLutBits + 1; n <= 16; n++) l += pk >= lc.limit[n];
                if (l > 16) { // corrupt data: consume the bits, decode nothing
                        l = 16;
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
}

// One lane per restart segment, one wave per workgroup.  The lanes of a wave walk their segments block by block in step (every segment
// holds the same blocks in the same order), each lane collecting the coefficients of its current block, in zigzag order, in a private
// tile in LDS; when the block is done the wave writes the 64 tiles out together, 128 contiguous bytes per block, zeros included -- so the
// coefficient planes need no clearing and no lane issues scattered 2-byte stores -- and leaves the tiles zeroed for the next block.
// The coefficient planes are in zigzag order; the IDCT kernel undoes it with compile-time indices.
constexpr int kTileWords = 36; // 32 words of coefficients, padded: 16-byte aligned rows that start in different LDS banks

__global__ __launch_bounds__(64) void huff_decode_kernel(const uint8_t *__restrict__ stream, const uint32_t *__restrict__ seg_off, int n_seg,
                                                         const int *__restrict__ found /* segments located on the GPU, or null: all n_seg are there */, ScanDev sp,
                                                         const HuffDev *__restrict__ tabs /* [0..3] DC, [4..7] AC */)
{
        __shared__ uint16_t luts[6][1 << kLutBits]; // the scan's DC tables then its AC tables
        __shared__ LongCodes longs[6];
        __shared__ __attribute__((aligned(16))) uint32_t tile[64 * kTileWords];
        __shared__ long tile_dst[64]; // where the lane's tile goes (in coefficients), -1: nowhere
        const int lane = threadIdx.x;
        for (int k = 0; k < sp.ns; k++) {
                for (int i = lane; i < (1 << kLutBits); i += 64) {
                        luts[k][i] = tabs[sp.td[k]].lut[i];
                        luts[3 + k][i] = tabs[4 + sp.ta[k]].lut[i];
                }
                for (int half = 0; half < 2; half++) {
                        const HuffDev &t = tabs[half ? 4 + sp.ta[k] : sp.td[k]];
                        LongCodes &lc = longs[3 * half + k];
                        for (int i = lane; i < 256; i += 64) lc.vals[i] = t.vals[i];
                        if (lane == 0) {
                                uint32_t limit = 0;
                                for (int n = 1; n <= 16; n++) {
                                        if (t.maxcode[n] >= 0) limit = (uint32_t) (t.maxcode[n] + 1) << (16 - n);
                                        lc.limit[n] = limit;
                                        lc.offset[n] = (int16_t) (t.valptr[n] - t.mincode[n]);
                                }
                        }
                }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) *(uint4 *) (tile + lane * kTileWords + 4 * j) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const int seg = blockIdx.x * 64 + lane;
        const bool present = seg < n_seg && (!found || seg < found[0]); // a segment the stream does not have decodes to zero blocks
        BitReader br;
        br.open(stream, present ? seg_off[seg] : 0u, present ? sp.scan_end : 0u);
        int pred[3] = { 0, 0, 0 };
        const long per_seg = sp.ri ? sp.ri : sp.units;
        const long u0 = (long) seg * per_seg;
        int16_t *const my = (int16_t *) (tile + lane * kTileWords);
        for (long i = 0; i < per_seg; i++) {
                const long u = u0 + i;
                const bool active = seg < n_seg && u < sp.units;
                const long ux = sp.single ? u % sp.bw1 : u % sp.mcu_w, uy = sp.single ? u / sp.bw1 : u / sp.mcu_w;
                for (int k = 0; k < sp.ns; k++) {
                        const LongCodes &tdc = longs[k], &tac = longs[3 + k];
                        for (int by = 0; by < sp.nbv[k]; by++) {
                                for (int bx = 0; bx < sp.nbh[k]; bx++) {
                                        tile_dst[lane] = active ? ((uy * sp.nbv[k] + by) * sp.gw[k] + ux * sp.nbh[k] + bx) * 64 : -1;
                                        if (active && present) {
                                                int v;
                                                br.refill();
                                                decode_symbol(br, tdc, luts[k], v);
                                                pred[k] += v;
                                                my[0] = (int16_t) pred[k];
                                                for (int z = 1; z < 64; z++) {
                                                        br.refill();
                                                        const int rs = decode_symbol(br, tac, luts[3 + k], v);
                                                        if ((rs & 15) == 0 && rs != 0xF0) break; // EOB
                                                        z += rs >> 4;
                                                        my[(rs & 15) && z < 64 ? z : 64] = (int16_t) v; // slot 64 is the tile's padding
                                                }
                                        }
                                        __syncthreads();
                                        // 8 lanes per tile, 16 bytes each: 8 tiles per round
#pragma unroll
                                        for (int round = 0; round < 8; round++) {
                                                const int t = round * 8 + (lane >> 3), part = lane & 7;
                                                const long dst = tile_dst[t];
                                                uint4 *src = (uint4 *) (tile + t * kTileWords + part * 4);
                                                const uint4 q = *src;
                                                *src = make_uint4(0, 0, 0, 0);
                                                if (dst >= 0) *(uint4 *) (sp.coef[k] + dst + part * 8) = q;
                                        }
                                        __syncthreads();
                                }
                        }
                }
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
__global__ __launch_bounds__(256) void idct_kernel(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qt /* 64, natural order */, int gw, long n_blocks,
                                                   uint8_t *__restrict__ plane, int pitch)
{
        const long b = (long) blockIdx.x * 256 + threadIdx.x;
        if (b >= n_blocks) return;
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
        uint32_t *seg_off = nullptr;
        size_t seg_cap = 0;
        int *scan_counts = nullptr; // [0]: segments located by the GPU marker scan, [1..]: markers per 4 KiB of the stream
        size_t scan_cap = 0;
        HuffDev *tabs = nullptr;     // 8 tables
        uint16_t *qt = nullptr;      // 4 x 64
        int16_t *coef[3] = { nullptr, nullptr, nullptr };
        uint8_t *plane[3] = { nullptr, nullptr, nullptr };
        size_t coef_cap[3] = { 0, 0, 0 }, plane_cap[3] = { 0, 0, 0 };
        uint8_t *tmp = nullptr;      // intermediate packed frame (UYVY or RGB) when the output needs a second conversion
        size_t tmp_cap = 0;
        // pinned staging for the tables and segment offsets
        void *pinned = nullptr;
        size_t pinned_cap = 0;
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

int ug_hip_jpeg_decoder_create(ug_hip_jpeg_decoder **out)
{
        if (!out) return UG_HIP_EINVAL;
        Decoder *d = new Decoder();
        hipError_t err = hipMalloc((void **) &d->tabs, 8 * sizeof(HuffDev));
        if (err == hipSuccess) err = hipMalloc((void **) &d->qt, 4 * 64 * sizeof(uint16_t));
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d->uploaded, hipEventDisableTiming);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_jpeg_decoder_create");
                delete d;
                return UG_HIP_ERUNTIME;
        }
        *out = (ug_hip_jpeg_decoder *) d;
        return UG_HIP_SUCCESS;
}

void ug_hip_jpeg_decoder_destroy(ug_hip_jpeg_decoder *dec)
{
        Decoder *d = (Decoder *) dec;
        if (!d) return;
        for (void *p : { (void *) d->stream, (void *) d->seg_off, (void *) d->scan_counts, (void *) d->tabs, (void *) d->qt, (void *) d->coef[0], (void *) d->coef[1], (void *) d->coef[2],
                         (void *) d->plane[0], (void *) d->plane[1], (void *) d->plane[2], (void *) d->tmp }) {
                if (p) (void) hipFree(p);
        }
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
        Decoder *d = (Decoder *) dec;
        if (!d || !jpeg_host || (!dst_dev && out != UG_PF_NONE)) {
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
        for (int c = 1; c < h.ncomp; c++) { // the sampling layouts the output stage knows: 4:4:4, 4:2:2, 4:2:0
                if (h.hs[c] != 1 || h.vs[c] != 1) {
                        ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: unsupported sampling factors");
                        return UG_HIP_EUNSUPP;
                }
        }
        hipStream_t st = (hipStream_t) stream;
        // ---- workspace ----
        const long mcus = (long) h.mcu_w * h.mcu_h;
        const long gpu_expect = h.ri ? (mcus + h.ri - 1) / h.ri : 1;
        const size_t scan_base = gpu_scan ? h.scans[0].data_begin & ~(size_t) 15 : 0;
        const unsigned scan_grid = gpu_scan ? (unsigned) ((len - scan_base + kScanChunk - 1) / kScanChunk) : 0;
        size_t n_seg_total = 0; // offsets the host uploads
        for (const Scan &sc : h.scans) n_seg_total += sc.seg_off.size();
        bool ok = grow((void **) &d->stream, &d->stream_cap, len + 32) &&
                  grow((void **) &d->seg_off, &d->seg_cap, (gpu_scan ? (size_t) gpu_expect : n_seg_total) * sizeof(uint32_t)) &&
                  grow((void **) &d->scan_counts, &d->scan_cap, ((size_t) scan_grid + 1) * sizeof(int));
        long gw[3], gh[3];
        for (int c = 0; c < h.ncomp && ok; c++) {
                gw[c] = (long) h.mcu_w * h.hs[c];
                gh[c] = (long) h.mcu_h * h.vs[c];
                d->plane_pitch[c] = (int) (gw[c] * 8);
                ok = grow((void **) &d->coef[c], &d->coef_cap[c], (size_t) (gw[c] * gh[c]) * 128) && grow((void **) &d->plane[c], &d->plane_cap[c], (size_t) (gw[c] * gh[c]) * 64);
        }
        const size_t pin_need = 8 * sizeof(HuffDev) + sizeof h.qt + n_seg_total * sizeof(uint32_t);
        if (ok && d->pinned_cap < pin_need) {
                if (d->upload_pending) (void) hipEventSynchronize(d->uploaded);
                d->upload_pending = false;
                if (d->pinned) (void) hipHostFree(d->pinned);
                d->pinned = nullptr;
                d->pinned_cap = 0;
                ok = hipHostMalloc(&d->pinned, pin_need + 4096, hipHostMallocDefault) == hipSuccess;
                if (ok) d->pinned_cap = pin_need + 4096;
        }
        if (!ok) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: out of device memory");
                return UG_HIP_ERUNTIME;
        }
        // ---- tables, segment offsets, stream -> device ----
        if (d->upload_pending) { // the previous call's asynchronous copies read the staging area: let them finish before it is rewritten
                UG_HIP_TRY(hipEventSynchronize(d->uploaded));
                d->upload_pending = false;
        }
        HuffDev *tabs_h = (HuffDev *) d->pinned;
        for (int t = 0; t < 4; t++) {
                build_dev(h.dc[t], tabs_h[t]);
                build_dev(h.ac[t], tabs_h[4 + t]);
        }
        uint16_t *qt_h = (uint16_t *) (tabs_h + 8);
        memcpy(qt_h, h.qt, sizeof h.qt);
        uint32_t *seg_h = (uint32_t *) (qt_h + 4 * 64);
        {
                size_t k = 0;
                for (const Scan &sc : h.scans) {
                        memcpy(seg_h + k, sc.seg_off.data(), sc.seg_off.size() * sizeof(uint32_t));
                        k += sc.seg_off.size();
                }
        }
        UG_HIP_TRY(hipMemcpyAsync(d->tabs, tabs_h, 8 * sizeof(HuffDev), hipMemcpyHostToDevice, st));
        UG_HIP_TRY(hipMemcpyAsync(d->qt, qt_h, sizeof h.qt, hipMemcpyHostToDevice, st));
        if (n_seg_total) UG_HIP_TRY(hipMemcpyAsync(d->seg_off, seg_h, n_seg_total * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        UG_HIP_TRY(hipMemcpyAsync(d->stream, jpeg_host, len, hipMemcpyHostToDevice, st));
        UG_HIP_TRY(hipEventRecord(d->uploaded, st));
        d->upload_pending = true;
        if (gpu_scan) {
                const Scan &sc = h.scans[0];
                hipLaunchKernelGGL(rst_count_kernel, dim3(scan_grid), dim3(kScanWG), 0, st, d->stream, scan_base, sc.data_begin, len, d->scan_counts + 1);
                hipLaunchKernelGGL(rst_place_kernel, dim3(scan_grid), dim3(kScanWG), 0, st, d->stream, scan_base, sc.data_begin, len, d->scan_counts + 1, d->seg_off,
                                   (int) gpu_expect, d->scan_counts);
        } else { // a scan of one component leaves the padding blocks of the MCU grid untouched, a short stream whole segments
                for (int c = 0; c < h.ncomp; c++) UG_HIP_TRY(hipMemsetAsync(d->coef[c], 0, (size_t) (gw[c] * gh[c]) * 128, st));
        }
        // ---- Huffman decoding, scan by scan ----
        size_t seg_base = 0;
        for (const Scan &sc : h.scans) {
                ScanDev sp = {};
                sp.ns = sc.ns;
                sp.single = sc.ns == 1 && h.ncomp > 1;
                sp.mcu_w = h.mcu_w;
                sp.ri = h.ri;
                sp.scan_end = (unsigned) sc.data_end;
                for (int k = 0; k < sc.ns; k++) {
                        const int c = sc.comp[k];
                        sp.comp[k] = c;
                        sp.td[k] = sc.td[k];
                        sp.ta[k] = sc.ta[k];
                        sp.nbh[k] = sp.single ? 1 : h.hs[c];
                        sp.nbv[k] = sp.single ? 1 : h.vs[c];
                        sp.gw[k] = (int) gw[c];
                        sp.coef[k] = d->coef[c];
                }
                if (sp.single) { // a non-interleaved scan walks the component's own block grid, ceil(size / 8) blocks (T.81 A.2.2)
                        const int c = sc.comp[0];
                        sp.bw1 = ((h.width * h.hs[c] + h.hmax - 1) / h.hmax + 7) / 8;
                        const int bh1 = ((h.height * h.vs[c] + h.vmax - 1) / h.vmax + 7) / 8;
                        sp.units = (long) sp.bw1 * bh1;
                } else {
                        sp.bw1 = 1;
                        sp.units = (long) h.mcu_w * h.mcu_h;
                }
                // only as many segments as the restart interval accounts for (a stream may carry fewer or more markers than it should)
                const long expect = h.ri ? (sp.units + h.ri - 1) / h.ri : 1;
                const int n_seg = gpu_scan ? (int) expect : (int) (expect < (long) sc.seg_off.size() ? expect : (long) sc.seg_off.size());
                hipLaunchKernelGGL(huff_decode_kernel, dim3((unsigned) ((n_seg + 63) / 64)), dim3(64), 0, st, d->stream, d->seg_off + seg_base, n_seg,
                                   gpu_scan ? d->scan_counts : nullptr, sp, d->tabs);
                seg_base += sc.seg_off.size();
        }
        // ---- dequantisation + IDCT ----
        for (int c = 0; c < h.ncomp; c++) {
                const long nb = gw[c] * gh[c];
                hipLaunchKernelGGL(idct_kernel, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, st, d->coef[c], d->qt + 64 * h.tq[c], (int) gw[c], nb, d->plane[c],
                                   d->plane_pitch[c]);
        }
        UG_HIP_LAUNCH_CHECK();
        if (out == UG_PF_NONE) return UG_HIP_SUCCESS; // planes only (tests)
        // ---- planes -> output codec ----
        const int w = h.width, hh = h.height;
        const bool rgb = h.is_rgb();
        if (!dst_pitch) dst_pitch = ug::linesize(out, w);
        if (h.ncomp != 3) {
                ug::set_last_error_msg("ug_hip_jpeg_decoder_decode: greyscale streams have no output mapping here");
                return UG_HIP_EUNSUPP;
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
