// jpeg_entropy.hip -- baseline JPEG entropy coding + JFIF container on gfx950, so that the JPEG path produces a
// decodable stream.  In UltraGrid this is the second half of gpujpeg_encoder_encode()
// (src/video_compress/gpujpeg.cpp:624, external libgpujpeg); the object shape below mirrors the call sites
// gpujpeg_encoder_create / _encode / _destroy (gpujpeg.cpp:353,624,639).
//
// Stream: SOI, APP0 (JFIF), DQT x2, SOF0 (8-bit, 3 components, Y 2x2 (4:2:0) or 2x1 (4:2:2) / 1x1 / 1x1), DHT x4 (T.81 Annex K.3
// tables), DRI, SOS (interleaved), entropy-coded segments of `restart_interval` MCUs separated by RSTm, EOI.
// Restart intervals make the scan data-parallel: every segment starts byte-aligned with DC predictors reset, so
// segments are coded independently -- one lane per BLOCK, whole segments per workgroup, fused with the forward DCT for UYVY / RGB input
// (jpeg_code_kernel, the default) or one wave per segment (entropy_wave_kernel, long restart intervals) -- and placed by their sizes.
// The byte stream is identical to the test writer tests/jpeg_bitstream.py, which Pillow/libjpeg decodes.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ug_common.h"
#include "jpeg_fdct_device.h"

namespace {

#include "jpeg_huffman_tables.h"

// zig-zag positions of the quantiser table entries in a DQT segment
const uint8_t kZigHost[64] = {
        0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

// ---------------------------------------------------------------------------------------------------------------
// Wave-cooperative Huffman coding: one wave per restart segment, one LANE PER COEFFICIENT of the current block.
//   1. lane k loads zz[k]; lane 0 turns it into the DC difference;
//   2. zero runs come from the ballot of non-zero AC lanes (distance to the previous set bit), so every emitting lane
//      builds its own bit string -- up to 3 ZRL codes + run/size code + value bits (+ EOB on the last emitter), <= 63 bits;
//   3. an exclusive wave scan of the string lengths gives every lane its bit position; lanes OR their strings into a
//      64-word LDS window (ds_or_b32), and the complete 32-bit words are flushed, byte-swapped, with one coalesced store;
//   4. the partial word is carried into the next block; the segment ends padded with 1-bits.
// The scan data written here is NOT yet byte-stuffed: 0xFF bytes are counted per segment and the 0x00 bytes are inserted
// by the compaction pass, which has to move every byte anyway.
// ---------------------------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes with DPP moves (no LDS traffic): Hillis-Steele inside the 16-lane rows, then the
// gfx9 row broadcasts carry the row totals across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_inclusive_scan(int v, int)
{
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31
        return v;
}

__device__ __forceinline__ int count_ff_bytes(uint32_t w)
{
        return ((w & 0xffu) == 0xffu) + ((w & 0xff00u) == 0xff00u) + ((w & 0xff0000u) == 0xff0000u) + ((w >> 24) == 0xffu);
}

constexpr int kRawBytesPerBlock = 224; // 56 words >= (31 carried + 64 x 27) bits
// The position of a segment in the final stream = header + the sizes of all segments before it.  The coders add every segment's final
// size to the total of its chunk of kChunk segments; a compaction wave then sums the chunk totals below its chunk and the segment
// sizes below it inside the chunk (two short wave reductions) instead of a separate single-workgroup prefix-sum launch.
constexpr int kChunk = 256;
// Each chunk total sits in a cache line of its own: thousands of atomics into ONE 128-byte line serialise at a single L2 channel
// (measured: the coder went from 21 to 80 us with the totals packed).
constexpr int kChunkStride = 32; // uint32 words

// A batch of frames in one launch: blockIdx.y = frame, every per-frame array of the coder `stride` elements (of its own type) behind the
// previous frame's.  One frame: all zero, gridDim.y = 1.
struct BatchStride {
        long coef_y, coef_c;  // int16 elements between the frames' coefficient arrays (luma; each chroma plane)
        long raw_words;       // the unstuffed segment data
        long seg;             // seg_len / seg_ff entries
        long tot_words;       // chunk totals
        long out_bytes;       // final streams (compaction only)
};
constexpr int kMaxBatch = 16; // frames per encode_batch call (the pinned length words of an encoder)

__global__ __launch_bounds__(256) void entropy_wave_kernel(const int16_t *__restrict__ cy, const int16_t *__restrict__ cb,
                                                           const int16_t *__restrict__ cr, int mcu_w, int n_mcu, int hs, int vs /* sampling factors of component 0: 2x2 (4:2:0), 2x1 (4:2:2), 1x1 (4:4:4) */,
                                                           int ctab /* Huffman table set of components 1,2: 1 = chroma (YCbCr), 0 = same as component 0 (RGB) */,
                                                           int nc /* components behind component 0 in the MCU: 2, or 0 for one scan of a non-interleaved stream (cy = that component) */,
                                                           int tab0 /* Huffman table set of component 0 */, int ri, int n_seg,
                                                           uint32_t *__restrict__ raw, int cap_words, uint32_t *__restrict__ seg_len,
                                                           uint32_t *__restrict__ seg_ff /* final size of the segment */,
                                                           uint32_t *__restrict__ chunk_tot /* sums of seg_ff over chunks of kChunk segments */, BatchStride bs)
{
        __shared__ uint32_t ac_tab[2][256], dc_tab[2][12];
        __shared__ uint32_t win[4][68];
        cy += blockIdx.y * bs.coef_y; cb += blockIdx.y * bs.coef_c; cr += blockIdx.y * bs.coef_c;
        raw += blockIdx.y * bs.raw_words; seg_len += blockIdx.y * bs.seg; seg_ff += blockIdx.y * bs.seg; chunk_tot += blockIdx.y * bs.tot_words;
        for (int i = threadIdx.x; i < 512; i += 256) ac_tab[i >> 8][i & 255] = kAcTab[i >> 8][i & 255];
        if (threadIdx.x < 24) dc_tab[threadIdx.x / 12][threadIdx.x % 12] = kDcTab[threadIdx.x / 12][threadIdx.x % 12];
        __syncthreads();
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int seg = blockIdx.x * 4 + wv;
        if (seg >= n_seg) return; // wave-uniform
        uint32_t *buf = win[wv];
        uint32_t *out = raw + (size_t) seg * cap_words;
        uint32_t carry_word = 0; // partial word, bits left-aligned
        int carry_bits = 0, wbase = 0, ff = 0;
        int pred[3] = { 0, 0, 0 };
        const int ybl = hs * vs, per_mcu = ybl + nc;
        const int m0 = seg * ri, n_blk = per_mcu * (min(n_mcu, (seg + 1) * ri) - m0);
        // Walk the blocks of the segment in scan order (per MCU: Y00 Y01 [Y10 Y11] Cb Cr) with incrementally updated
        // wave-uniform indices (one division per segment), always one block ahead: the load of block t+1 is issued before
        // block t is coded.
        int mx = m0 % mcu_w, my = m0 / mcu_w, m = m0, b_next = 0;
        auto next_ptr = [&]() { // pointer of block (m, b_next), then advance
                const int yrow = vs * my + (hs == 2 ? b_next >> 1 : 0), ycol = hs * mx + (hs == 2 ? b_next & 1 : 0); // block of component 0
                const int16_t *p = b_next < ybl ? cy + 64 * ((long) yrow * (hs * mcu_w) + ycol)
                                                : (b_next == ybl ? cb : cr) + 64L * m;
                if (++b_next == per_mcu) {
                        b_next = 0;
                        m++;
                        if (++mx == mcu_w) { mx = 0; my++; }
                }
                return p;
        };
        int v_next = next_ptr()[lane];
        int b = -1;
        for (int t = 0; t < n_blk; t++) {
                {
                        b = b == per_mcu - 1 ? 0 : b + 1;
                        const int comp = b < ybl ? tab0 : ctab, pi = b < ybl ? 0 : b - ybl + 1;
                        int v = v_next;
                        if (t + 1 < n_blk) v_next = next_ptr()[lane];
                        const int dc = __builtin_amdgcn_readfirstlane(v);
                        if (lane == 0) v = dc - pred[pi];
                        pred[pi] = dc;
                        const unsigned long long acmask = __ballot(lane > 0 && v != 0);
                        const int a = v < 0 ? -v : v;
                        const int size = a ? 32 - __builtin_clz((unsigned) a) : 0;
                        const uint32_t vbits = (uint32_t) (v < 0 ? v + (1 << size) - 1 : v) & ((1u << size) - 1);
                        // zero run before this lane = distance to the previous non-zero AC lane (or to the DC lane)
                        const unsigned long long below = (acmask | 1ull) & ((1ull << lane) - 1ull);
                        const int prev = lane ? 63 - __builtin_clzll(below) : 0;
                        const int run = lane ? lane - prev - 1 : 0;
                        const uint32_t e = lane == 0 ? dc_tab[comp][size] : ac_tab[comp][((run & 15) << 4) | size];
                        const bool emits = lane == 0 || ((acmask >> lane) & 1ull);
                        const int last = 63 - __builtin_clzll(acmask | 1ull); // wave-uniform
                        // Common case (no zero run longer than 15 in the block): every string fits 32 bits -- Huffman code (<= 16)
                        // + value bits (<= 11) -- once the EOB is emitted by the otherwise idle lane last + 1 instead of being
                        // appended to the last coefficient's string.  Blocks with ZRL symbols take the general 64-bit path.
                        const bool fast = !__any(emits && run > 15);
                        unsigned long long bits = 0;
                        uint32_t bits32 = 0;
                        int nb;
                        if (fast) {
                                const bool eob_lane = lane == last + 1; // exists iff last < 63
                                const uint32_t eob = ac_tab[comp][0x00];
                                bits32 = eob_lane ? (eob & 0xffff) : (((e & 0xffff) << size) | vbits);
                                nb = eob_lane ? (int) (eob >> 16) : (int) (e >> 16) + size;
                                if (!(emits || eob_lane)) nb = 0;
                        } else {
                                bits = ((unsigned long long) (e & 0xffff) << size) | vbits;
                                nb = (int) (e >> 16) + size;
                                if (lane > 0 && run > 15) { // 1..3 ZRL symbols in front
                                        const uint32_t z = ac_tab[comp][0xF0];
                                        const unsigned long long zc = z & 0xffff;
                                        const int zl = (int) (z >> 16), nz = run >> 4;
                                        unsigned long long zz3 = zc;
                                        if (nz > 1) zz3 = (zz3 << zl) | zc;
                                        if (nz > 2) zz3 = (zz3 << zl) | zc;
                                        bits |= zz3 << nb;
                                        nb += nz * zl;
                                }
                                if (lane == last && last < 63) { // EOB after the last non-zero coefficient
                                        const uint32_t eob = ac_tab[comp][0x00];
                                        bits = (bits << (eob >> 16)) | (eob & 0xffff);
                                        nb += (int) (eob >> 16);
                                }
                                if (!emits) nb = 0;
                        }
                        const int incl = wave_inclusive_scan(nb, lane);
                        const int total = __builtin_amdgcn_readlane(incl, 63);
                        const int pos = carry_bits + incl - nb;
                        // window: word 0 starts with the carried partial word
                        buf[lane] = lane == 0 ? carry_word : 0;
                        if (lane < 4) buf[64 + lane] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (nb) {
                                const int j = pos >> 5, o = pos & 31;
                                if (fast) {
                                        const uint32_t s32 = bits32 << (32 - nb); // left-aligned string
                                        atomicOr(&buf[j], s32 >> o);
                                        if (o + nb > 32) atomicOr(&buf[j + 1], s32 << (32 - o));
                                } else {
                                        const unsigned long long s64 = bits << (64 - nb);
                                        atomicOr(&buf[j], (uint32_t) (s64 >> (32 + o)));
                                        if (o + nb > 32) atomicOr(&buf[j + 1], (uint32_t) (s64 >> o));
                                        if (o + nb > 64) atomicOr(&buf[j + 2], (uint32_t) (s64 << (32 - o)));
                                }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int tot = carry_bits + total, nw = tot >> 5;
                        const uint32_t word = buf[lane];
                        if (lane < nw) {
                                out[wbase + lane] = __builtin_bswap32(word); // stream order in memory
                        }
                        if (lane < nw) ff += count_ff_bytes(word); // per lane; reduced once at the end of the segment
                        carry_word = __builtin_amdgcn_readlane(word, nw & 63);
                        carry_bits = tot & 31;
                        wbase += nw;
                        __builtin_amdgcn_wave_barrier();
                }
        }
        // pad the last partial byte with 1-bits (T.81 F.1.2.3) and emit the tail bytes
        int tail = 0;
        if (carry_bits) {
                tail = (carry_bits + 7) >> 3;
                const int pad = 8 * tail - carry_bits;
                carry_word |= ((1u << pad) - 1u) << (32 - carry_bits - pad);
                if (lane == 0) {
                        out[wbase] = __builtin_bswap32(carry_word);
                        for (int i = 0; i < tail; i++) ff += ((carry_word >> (24 - 8 * i)) & 0xff) == 0xff;
                }
        }
        ff = __builtin_amdgcn_readlane(wave_inclusive_scan(ff, lane), 63);
        if (lane == 0) {
                seg_len[seg] = (uint32_t) (4 * wbase + tail);
                seg_ff[seg] = (uint32_t) (4 * wbase + tail + ff + 2); // bytes this segment occupies in the final stream (stuffed + marker)
                atomicAdd(&chunk_tot[(seg / kChunk) * kChunkStride], (uint32_t) (4 * wbase + tail + ff + 2));
        }
}

// LDS window of the block-parallel coder: 64 bytes per block on average (a 4K q75 frame needs ~10); a wave whose segments
// need more goes through its window in several passes
constexpr int kWinWordsPerBlock = 16;

// coefficients per skip group of the walks (a group that is zero in all 64 blocks of the wave costs one scalar branch)
#ifndef UG_JPEG_WALK_GROUP
#define UG_JPEG_WALK_GROUP 8 // 4 measures the same on 4K S2 content at q75 (23.7 vs 23.1 us)
#endif
constexpr int kWalkGroup = UG_JPEG_WALK_GROUP;

// Per-lane bit accumulator of the emission walk.  Branch-free: a lane that appends nothing shifts in zero bits, and a lane whose
// accumulator has not filled a word ORs 0 into its current window word -- every lane runs the same instructions.
// MULTI = false: the whole segment fits its window, `word` walks through it.  MULTI = true (a segment longer than its window:
// near-lossless quality on noise): the window shows words [lo_idx, lo_idx + cap) of the segment in this pass; words outside it
// go to a spare word nobody reads.
template <bool MULTI>
struct BitSink {
        uint32_t hi, lo;   // 64-bit accumulator, left-aligned: the next bit goes to position 63 - fill
        uint32_t fill;     // < 32 between appends
        uint32_t *word;    // !MULTI: window word the top half of the accumulator belongs to; MULTI: the segment's window
        uint32_t idx, lo_idx, cap; // MULTI: segment-relative index of that word; the range the window shows
        uint32_t *spare;
        __device__ __forceinline__ uint32_t *target() const
        {
                if (!MULTI) return word;
                const uint32_t rel = idx - lo_idx;
                return rel < cap ? word + rel : spare;
        }
        __device__ __forceinline__ void append(uint32_t str, uint32_t n) // n <= 27 bits; n == 0 (with str == 0) appends nothing
        {
                const uint32_t t = fill + n;
                const unsigned long long sh = (unsigned long long) str << ((64u - t) & 63u); // t == 0: str == 0, any shift will do
                hi |= (uint32_t) (sh >> 32);
                lo |= (uint32_t) sh;
                const uint32_t fl = t >> 5, m = 0u - fl; // fl = 1: the top word is complete
                atomicOr(target(), hi & m);
                hi = (lo & m) | (hi & ~m);
                lo &= ~m;
                fill = t & 31u;
                if (MULTI) idx += fl;
                else word += fl;
        }
        __device__ __forceinline__ void finish() { atomicOr(target(), hi); } // the last, partial word (0 if the block ended on a word boundary)
};

__device__ __forceinline__ int coef_at(const uint32_t (&w)[32], int k)
{
        return (k & 1) ? (int) w[k >> 1] >> 16 : (int) (w[k >> 1] << 16) >> 16;
}

// LDS copy of an AC table as the block-parallel coder reads it: entry of symbol (run << 4 | size) = code << size | (code length + size) << 27 --
// the bits of the value go straight under the code and the total length needs no addition; the entries of EOB (0x00) and ZRL (0xF0), which
// are emitted explicitly, and of the symbols T.81 does not define are zero: a zero coefficient (size 0) looks up an empty code and appends
// nothing without any predication.
__device__ __forceinline__ uint32_t packed_ac_entry(uint32_t e /* code | length << 16 */, int sym)
{
        const uint32_t size = (uint32_t) sym & 15u;
        return (e == 0 || sym == 0x00 || sym == 0xF0) ? 0u : ((e & 0xffffu) << size) | (((e >> 16) + size) << 27);
}
constexpr uint32_t kCodeMask = (1u << 27) - 1u;

// The general path's pass over the 63 AC coefficients of every lane's block (static register indices): appends the codes to `sink`, which
// writes straight into the segment's window.  The pass is cut into groups of 8 coefficients: a group that is zero in all 64 blocks is
// skipped with one scalar branch, and inside a group there is no control flow at all; every coefficient may have 1..3 ZRL symbols in front.
template <class Sink>
__device__ __forceinline__ void walk_block(const uint32_t (&w)[32], const uint32_t *tab, uint32_t zrl, uint32_t eob, Sink &sink)
{
        const uint32_t zl = zrl >> 16, zc = zrl & 0xffffu;
        uint32_t run = 0;
#pragma unroll
        for (int g = 0; g < 64 / kWalkGroup; g++) {
                uint32_t any = g == 0 ? w[0] & 0xffff0000u : w[kWalkGroup / 2 * g]; // the DC value is not an AC coefficient
#pragma unroll
                for (int i = 1; i < kWalkGroup / 2; i++) any |= w[kWalkGroup / 2 * g + i];
                if (__ballot(any != 0) == 0) { // wave-uniform: nobody has a coefficient in this group
                        run += g == 0 ? kWalkGroup - 1 : kWalkGroup;
                        continue;
                }
#pragma unroll
                for (int k = g == 0 ? 1 : kWalkGroup * g; k < kWalkGroup * (g + 1); k++) {
                        const int v = coef_at(w, k);
                        const uint32_t neg = (uint32_t) (v >> 31);
                        const uint32_t a = ((uint32_t) v ^ neg) - neg;
                        const uint32_t size = 32u - (uint32_t) __clz((int) a); // 0 for a zero coefficient
                        const uint32_t e = tab[((run & 15u) << 4) | size];
#pragma unroll
                        for (uint32_t z = 0; z < 3; z++) { // 1..3 ZRL symbols in front of this coefficient
                                const bool on = size != 0 && run > 15 + 16 * z;
                                sink.append(on ? zc : 0u, on ? zl : 0u);
                        }
                        const uint32_t vb = __builtin_amdgcn_ubfe((uint32_t) v + neg, 0, size); // v < 0: the low bits of v - 1 (T.81 F.1.2.1)
                        sink.append((e & kCodeMask) | vb, e >> 27);
                        run = size ? 0u : run + 1u;
                }
        }
        sink.append(run ? eob & 0xffffu : 0u, run ? eob >> 16 : 0u); // EOB after the last non-zero coefficient (not when position 63 is coded)
}

// ---------------------------------------------------------------------------------------------------------------
// Block-parallel Huffman coding with byte stuffing and stream placement (round 4; the default whenever a restart segment has at most 256
// blocks).  One LANE PER BLOCK, a workgroup = the whole segments that fit its lanes; the workgroup finishes its part of the JPEG stream
// itself -- code, byte stuffing, RSTm markers:
//   1. every lane gets its block as 32 registers (64 int16, zig-zag order): from the coefficient arrays in HBM (SRC = 0: lanes in scan order,
//      8 lanes share a 128-byte line, rows handed to their owners through LDS) or, fused (SRC = 420 / 422 / 444 / 1420), from the workgroup's
//      own forward DCT + quantiser of 32 (64) consecutive MCUs of the UYVY (RGB, I420) frame, made in FRAME order -- a wave = a luma block row,
//      or the chroma blocks, of the MCUs -- and coded by the lane that made it: the coefficients never exist in HBM and never change lanes;
//      what the scan order decides goes through LDS arrays indexed by the block's scan index `sid`;
//   2. the DC predictor = the DC value at sid - 1 / - 3 / - blocks per MCU (the previous block of the same component);
//   3. ONE walk over the 63 AC coefficients (static register indices; groups of 8 and single positions that are zero in all 64 blocks of the
//      wave cost a scalar branch; the rare coefficient behind a zero run longer than 15 is met by a wave-uniform branch at its position)
//      appends code + value bits to a 64-bit accumulator whose words go to the lane's PRIVATE string in LDS (PrivSink)
//      and adds up the length -- rounds 2-3 walked twice, a length pass and an emission pass;
//   4. a prefix sum of the lengths in scan order, made segment-relative, gives every block its bit position; every lane shifts its private
//      string there and ORs it into the segment's window (v_alignbit + ds_or_b32);
//   5. byte stuffing and write-out by all lanes at once: K consecutive window words per lane, the bytes each will write counted (0x00 after
//      every 0xFF, T.81 B.1.1.5; RSTm / EOI behind each segment; the last byte's padding ORed in on the way), one prefix sum over the
//      workgroup = every position in the workgroup's stretch of the stream, then the bytes;
//   6. placement.  Two launches (the default from two frames per call up): the stretch goes into a slot of the workgroup's own, its byte
//      count into wg_bytes, and jpeg_gather_kernel moves the
//      stretches to their places.  One launch (one-frame calls; the fallback when a slot is too small; UG_JPEG_LOOKBACK=1): a decoupled
//      look-back over the workgroups of the frame (one 64-bit status word per workgroup: generation of the call | aggregate / inclusive prefix |
//      bytes) gives the workgroup its position in the stream, the lanes write there; the last segment's last lane reports the stream length to
//      pinned host memory, workgroup 0 lays down the header.
// A block whose code exceeds its 512-bit private string, or a segment beyond its window (near-lossless quality on noise), sends its workgroup
// down the general path: emission straight into the windows at the known bit positions (BitSink), in several passes when a segment exceeds its
// window, counted first and emitted again for the write-out (segment by segment, a wave each: count_pass / place / write_pass).  The stream is
// byte-identical to the wave-per-segment coder + compaction.
// ---------------------------------------------------------------------------------------------------------------
#ifndef UG_JPEG_SKIP_POSITIONS
#define UG_JPEG_SKIP_POSITIONS 1 // inside a group that is not empty, still skip the positions that are zero in all 64 blocks of the wave
#endif
constexpr int kPrivWords = 16;         // private string of a block: 512 bits (a 4K q75 frame needs ~80)
constexpr int kPrivStride = 17;        // + one dump word for what does not fit; odd stride: the lanes' rows start in different banks

// the lane's private bit string in LDS (words of `base`, MSB first).  New bits enter a 64-bit accumulator at the bottom.  `pos` is the bit
// address (relative to `base`) of the next bit MINUS 32: pos >> 5 is the word in front of the one being filled, pos & 31 the bits the latter
// has.  Every append stores acc >> (pos & 31) -- one v_alignbit -- at word pos >> 5: when the append completed a word, that is the word; when
// it did not, it is the word completed before, written once more with the same bits (the accumulator still holds them; in front of the
// string's first word that is the dump word of the row below, or the pad word in front of the rows).  No control flow, no masking, no separate
// fill / index bookkeeping: 6 operations (round 4's first form kept fill and index apart and took 9).
struct PrivSink {
        unsigned long long acc;
        uint32_t pos;
        uint32_t *base;
        uint32_t first; // word index of the row's first word (the dump word: first + kPrivWords)
        // CLAMP = false: the caller knows the word cannot lie outside the row (at most 8 complete words at the head of a group of 8 coefficients without ZRL symbols)
        template <bool CLAMP = true>
        __device__ __forceinline__ void append(uint32_t str, uint32_t n) // n <= 27 bits; n == 0 (with str == 0) appends nothing
        {
                acc = (acc << n) | str;
                pos += n;
                const uint32_t word = __builtin_amdgcn_alignbit((uint32_t) (acc >> 32), (uint32_t) acc, pos);
                base[CLAMP ? min(pos >> 5, first + (uint32_t) kPrivWords) : pos >> 5] = word;
        }
        __device__ __forceinline__ uint32_t bits() const { return pos + 32u - 32u * first; }
        __device__ __forceinline__ uint32_t words() const { return bits() >> 5; } // complete words so far
        __device__ __forceinline__ void finish() // the last, partial word, left-aligned (no partial word: a value beyond the string's end, never read)
        {
                base[min((pos >> 5) + 1u, first + (uint32_t) kPrivWords)] = (uint32_t) acc << ((32u - pos) & 31u);
        }
};

// 8 coefficients of the one-walk coder.  ZRL: some block of the wave may meet a coefficient behind a zero run longer than 15 in this group
// (then a wave-uniform branch at that position puts the 1..3 ZRL symbols in front of it); CLAMP: some private string may reach its end here.
template <bool ZRL, bool CLAMP>
__device__ __forceinline__ void walk_group_private(const uint32_t (&w)[32], const int g, const uint32_t *tab, uint32_t zl, uint32_t zc, uint32_t &run16, PrivSink &sink)
{
#pragma unroll
        for (int k = g == 0 ? 1 : kWalkGroup * g; k < kWalkGroup * (g + 1); k++) {
                const int v = coef_at(w, k);
                const bool nz = v != 0;
#if UG_JPEG_SKIP_POSITIONS
                if (__ballot(nz) == 0) { // wave-uniform: this position is zero in all 64 blocks (two operations against ~20)
                        run16 += 16u;
                        continue;
                }
#endif
                // T.81 F.1.2.1 without the absolute value: t = v (v >= 0) or v - 1 (v < 0) has |v|'s bit count in front of its sign run and
                // its low `size` bits are the value bits; v_ffbh_i32 counts the sign run (-1 for t = 0: the saturating subtraction makes that size 0)
                const int t = v + (v >> 31);
                uint32_t sign_run;
                asm("v_ffbh_i32 %0, %1" : "=v"(sign_run) : "v"(t));
                const uint32_t size = __builtin_elementwise_sub_sat(32u, sign_run);
                const uint32_t e = tab[(run16 & 0xF0u) | size];       // packed_ac_entry: a zero coefficient appends nothing
                if (ZRL && k > 16 && __ballot(nz && run16 > 15u * 16u) != 0) { // (a run of 16 zeros ends at position 17 at the earliest)
                        const uint32_t zr = nz ? run16 >> 8 : 0u; // one ZRL, then the other two together (<= 22 bits)
                        sink.append<true>(zr ? zc : 0u, zr ? zl : 0u);
                        const uint32_t two = zr > 2u ? (zc << zl) | zc : zc;
                        sink.append<true>(zr > 1u ? two : 0u, zr > 1u ? (zr - 1u) * zl : 0u);
                }
                const uint32_t vb = __builtin_amdgcn_ubfe((uint32_t) t, 0, size);
                sink.append<CLAMP>((e & kCodeMask) | vb, e >> 27);
                run16 = nz ? 0u : run16 + 16u;
        }
}

// The AC part of every lane's block, appended to `sink` (whose bits() then says how long the block's code is, EOB included).  Groups of
// kWalkGroup coefficients and single positions that are zero in all 64 blocks of the wave cost a scalar branch; what a group has to be able to
// do -- ZRL symbols, strings that end -- is decided per group, wave-uniformly, so that the usual group carries neither test.
__device__ __forceinline__ void walk_private(const uint32_t (&w)[32], const uint32_t *tab, uint32_t zrl, uint32_t eob, PrivSink &sink)
{
        const uint32_t zl = zrl >> 16, zc = zrl & 0xffffu;
        uint32_t run16 = 0; // 16 x the zero run: the table row's offset as it is
#pragma unroll
        for (int g = 0; g < 64 / kWalkGroup; g++) {
                uint32_t any = g == 0 ? w[0] & 0xffff0000u : w[kWalkGroup / 2 * g]; // the DC value is not an AC coefficient
#pragma unroll
                for (int i = 1; i < kWalkGroup / 2; i++) any |= w[kWalkGroup / 2 * g + i];
                if (__ballot(any != 0) == 0) { // wave-uniform: nobody has a coefficient in this group
                        run16 += 16u * (g == 0 ? kWalkGroup - 1 : kWalkGroup);
                        continue;
                }
                // a zero run longer than 15 can end inside this group only in a lane that enters it with run + (kWalkGroup - 1) > 15
                if (kWalkGroup * (g + 1) > 17 && __ballot(any != 0 && run16 + 16u * (uint32_t) (kWalkGroup - 1) > 16u * 15u) != 0) {
                        walk_group_private<true, true>(w, g, tab, zl, zc, run16, sink);
                } else if (__ballot(sink.words() > (uint32_t) (kPrivWords - kWalkGroup)) != 0) { // 8 codes of <= 27 bits: at most 8 more words
                        walk_group_private<false, true>(w, g, tab, zl, zc, run16, sink);
                } else {
                        walk_group_private<false, false>(w, g, tab, zl, zc, run16, sink);
                }
        }
        sink.append<true>(run16 ? eob & 0xffffu : 0u, run16 ? eob >> 16 : 0u); // EOB after the last non-zero coefficient (not when position 63 is coded)
}

__device__ __forceinline__ int count_ff_valid(uint32_t word, int valid) // 0xFF bytes among the first `valid` (stream order) bytes of a window word
{
        int c = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) c += (t < valid && ((word >> (24 - 8 * t)) & 0xffu) == 0xffu) ? 1 : 0;
        return c;
}

// ---- position of a workgroup's bytes in the stream: decoupled look-back over the workgroups of the frame ----
// status word = generation of the call << 34 | state << 32 | bytes ; state 1: the workgroup's own byte count, 2: the count of all
// workgroups up to and including it.  Words of other generations (earlier calls, other batch sizes) read as "not there yet", so the
// array is never cleared between calls.  Workgroups are dispatched in index order, so the predecessors a workgroup waits for are
// running or done.  Device-scope atomic loads / stores: the word carries its own data, no other memory has to be ordered with it.
constexpr unsigned long long status_word(uint32_t gen, uint32_t state, uint32_t bytes) { return ((unsigned long long) gen << 34) | ((unsigned long long) state << 32) | bytes; }

// A wait that has not ended after kSpinLimit polls (~0.1 s; the whole kernel takes tens of microseconds) is given up and reported through *stuck
// (the host fails the call): a defect must not be able to hang the GPU.
constexpr int kSpinLimit = 1 << 19;
__device__ __forceinline__ uint32_t lookback_exclusive(unsigned long long *status, int wg, uint32_t aggregate, uint32_t gen, int lane, uint32_t *stuck) // one wave
{
        if (wg == 0) {
                if (lane == 0) __hip_atomic_store(&status[0], status_word(gen, 2, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0;
        }
        if (lane == 0) __hip_atomic_store(&status[wg], status_word(gen, 1, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (int pos = wg - 1;; pos -= 64) { // 64 predecessors at a time, lane 0 = the nearest
                const int idx = pos - lane;
                unsigned long long s, full;
                int last;
                // wait only for what the sum needs: the predecessors up to the nearest one that already knows its inclusive prefix (measured: waiting
                // for all 64 of the window to have reported cost 12 k of a workgroup's 48 k cycles -- one late workgroup among 64 held up all behind it)
                for (int polls = 0;; polls++) {
                        s = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : status_word(gen, 2, 0);
                        const uint32_t state = (uint32_t) (s >> 34) == gen ? (uint32_t) (s >> 32) & 3u : 0u;
                        const unsigned long long there = __ballot(state != 0);
                        full = __ballot(state == 2u);
                        last = full ? __builtin_ctzll(full) : 63; // the nearest predecessor that knows its inclusive prefix ends the walk
                        const unsigned long long need = last == 63 ? ~0ull : (2ull << last) - 1ull;
                        if ((there & need) == need) break;
                        if (polls == kSpinLimit) {
                                if (lane == 0) *stuck = 1u;
                                if (state == 0) s = status_word(gen, 2, 0);
                                break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                }
                const int v = lane <= last ? (int) (uint32_t) s : 0;
                excl += (uint32_t) __builtin_amdgcn_readlane(wave_inclusive_scan(v, lane), 63);
                if (full) break;
        }
        if (lane == 0) __hip_atomic_store(&status[wg], status_word(gen, 2, excl + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return excl;
}

// The flat form, for one-frame calls (round 5).  All workgroups of one frame are resident at once (a 4K 4:2:0 frame: 1 013 workgroups, the chip
// holds 1 536) and finish coding within a few microseconds of each other, so with the windowed walk above the LAST workgroups resolve their prefix
// window after window -- up to n_wg / 64 = 16 dependent memory round trips (measured with UG_JPEG_PROF: 11.4 k clocks of waiting per workgroup on
// average at one frame per call against 0.4 k at eight).  Here nobody publishes an inclusive prefix: every workgroup leaves its own byte count and sums
// ALL its predecessors' counts itself, 64 * U words per round with the U loads of a lane in flight together -- two rounds for the last workgroup of
// a 4K frame instead of sixteen; n_wg^2 / 2 eight-byte loads per frame (4 MB at 4K), all of them L2 hits.
template <int U>
__device__ __forceinline__ uint32_t lookback_exclusive_flat(unsigned long long *status, int wg, uint32_t aggregate, uint32_t gen, int lane, uint32_t *stuck) // one wave
{
        if (lane == 0) __hip_atomic_store(&status[wg], status_word(gen, 1, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t sum = 0;
        for (int base = 0; base < wg; base += 64 * U) {
                unsigned long long s[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                        const int idx = base + 64 * u + lane;
                        s[u] = idx < wg ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : status_word(gen, 1, 0);
                }
                for (int polls = 0;; polls++) {
                        bool missing = false;
#pragma unroll
                        for (int u = 0; u < U; u++) missing = missing || (uint32_t) (s[u] >> 34) != gen || ((uint32_t) (s[u] >> 32) & 3u) == 0u;
                        if (__ballot(missing) == 0) break;
                        if (polls == kSpinLimit) {
                                if (lane == 0) *stuck = 1u;
                                break;
                        }
                        __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int u = 0; u < U; u++) {
                                const int idx = base + 64 * u + lane;
                                if ((uint32_t) (s[u] >> 34) != gen || ((uint32_t) (s[u] >> 32) & 3u) == 0u) {
                                        s[u] = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                        }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                        if ((uint32_t) (s[u] >> 34) == gen && ((uint32_t) (s[u] >> 32) & 3u) != 0u) sum += (uint32_t) s[u];
                }
        }
        return (uint32_t) __builtin_amdgcn_readlane(wave_inclusive_scan((int) sum, lane), 63);
}

struct CodeArgs {
        // scan geometry
        int mcu_w, n_mcu, hs, vs, ctab, ri, n_seg, S /* blocks per full segment */, G /* segments per workgroup (SRC = 0) */, n_wg /* workgroups per frame */;
        // SRC = 0 only: nc = chroma blocks of an MCU -- 2, or 0 for the scan of ONE component (a non-interleaved scan, T.81 A.2.2: its MCU is one block, cy = that
        // component's blocks); tab0 = the Huffman table pair of the blocks b < hs * vs (0; 1 for the Cb / Cr scans of a non-interleaved YCbCr stream)
        int nc, tab0;
        // SRC = 0, one-launch placement only: the stream of frame f goes on at byte base[f] - 2 of its buffer -- behind the previous scan of a non-interleaved stream,
        // over the EOI that scan ended with (NULL: at byte 0).  The word is the length the previous scan's launch left in (mapped, pinned) host memory: kernels of one
        // stream run in order, so it is final when this launch reads it.
        const uint32_t *base;
        // SRC = 0: quantised blocks in HBM, per-frame strides in int16 elements
        const int16_t *cy, *cb, *cr;
        long coef_y, coef_c;
        // SRC = 420 / 422: the UYVY frame(s) and the quantiser (luma 64 divisors, chroma 64)
        const uint8_t *src;
        int pitch, width, height;
        size_t src_stride;
        // the stream(s)
        uint8_t *out;
        size_t out_stride, capacity;
        const uint8_t *header;
        int header_len;
        uint32_t *total_pinned;         // pinned host memory mapped into the device: the lengths need no copy back; word kMaxBatch = "a wait was given up"
        unsigned long long *status;     // n_status words per frame
        long n_status;
        uint32_t gen;
        uint32_t *ticket;               // workgroups take their index from here, in the order they start (0 before and after every launch)
        int flat;                       // one-launch placement: 1 = every workgroup sums all its predecessors' byte counts itself (lookback_exclusive_flat; one-frame calls)
        // two-launch placement (the default): the workgroup leaves its finished bytes in a slot of its own and its byte count in wg_bytes; the gather
        // launch behind it moves the stretches to their places.  slots == NULL: one launch, the position comes from the look-back (status, gen)
        uint8_t *slots;
        size_t slot_bytes;
        uint32_t *wg_bytes;
        unsigned long long *prof;       // UG_JPEG_PROF=1: per workgroup, kProfPhases clock deltas + a mark (averaged and printed when the encoder is destroyed); else NULL
        // divisions by the scan's constants as multiplications: x / d = (x * m16) >> 16 with m16 = 65536 / d + 1 where x * d < 2^16 (lane and block
        // numbers by S <= 256 and by the blocks of an MCU), = the high word of x * m32 with m32 = 2^32 / d + 1 where x * d < 2^32 (window words by S;
        // MCU numbers by the MCUs of a row -- the host takes the fused kernels only where that holds and the row has more than one MCU)
        uint32_t S_m16, per_mcu_m16, S_m32, mcu_w_m32;
};
__device__ __forceinline__ int div16(int x, uint32_t m16) { return (int) (__umul24((uint32_t) x, m16) >> 16); }
__device__ __forceinline__ int div32(int x, uint32_t m32) { return (int) __umulhi((uint32_t) x, m32); }
constexpr int kProfPhases = 10;

// WAVES = waves per workgroup.  SRC = 0: a workgroup codes the G = 64 * WAVES / S whole segments that fit its lanes (the host picks the WAVES
// that leaves the fewest lanes idle).  SRC = 420 (WAVES = 3) / 422 (WAVES = 2) / 444 (WAVES = 3) / 1420 (planar I420, WAVES = 3): a workgroup = 32 (444: 64) consecutive MCUs of the
// scan = 32 / ri whole segments (the restart interval must divide 32 / 64); UYVY: a 16-byte aligned frame of width % 16 == 0.
// phase clock of the profiling runs: thread 0 of every workgroup adds the time since the previous mark to phase `i` (a.prof == NULL: nothing)
#define UG_PHASE(i)                                                                                                      \
        if (a.prof != nullptr && threadIdx.x == 0) {                                                                     \
                const unsigned long long now_ = __builtin_readcyclecounter();                                            \
                a.prof[((size_t) blockIdx.y * gridDim.x + blockIdx.x) * (kProfPhases + 1) + i] = now_ - prof_t; /* a slot per workgroup: no contention */ \
                prof_t = now_;                                                                                           \
        }

// The register allocator's occupancy target (amdgpu_waves_per_eu = the MINIMUM it has to keep; the workgroup's LDS caps the resident waves at
// 4.5 per SIMD anyway).  Round 5, interleaved A/B over every call form (profiles/r05_jpeg_code_occupancy.txt): with a target of 3 instead of 5 the
// 4:2:0 kernel keeps its 95 VGPRs but is scheduled for latency instead of occupancy -- 12.1 -> 11.4 us per 4K frame at 8 per call, one frame
// per call unchanged (31.9); 4:2:2 gains as much in batches (15.4 -> 14.3) and LOSES as much one frame per call (35.0 -> 37.6), so the batch
// launches (BATCH = true, two frames or more per call) take the target-3 instantiation and the one-frame call keeps 4; RGB 4:4:4, I420 and
// the unfused variants are indifferent and stay at 4.
template <int WAVES, int SRC, bool BATCH = false>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(SRC == 420 || (SRC == 422 && BATCH) ? 3 : 4))) void jpeg_code_kernel(const CodeArgs a, const float *__restrict__ div /* the quantiser (fused variants): a parameter of its own, restrict, so that its 128 words are scalar loads */)
{
        constexpr int W = 64 * WAVES;
        // The look-back below waits for workgroups with smaller indices.  Index = blockIdx.x within the frame blockIdx.y: the dispatcher starts the workgroups of a grid in
        // index order (per XCD, which is all the argument needs: the lowest unfinished index is always running or next in line for a slot that
        // only lower indices hold).  Should a wait ever be given up (kSpinLimit), the encoder switches to a.ticket != nullptr for good: the index
        // is then a ticket drawn when the workgroup STARTS, so that every index it waits for belongs to a workgroup that is already running whatever
        // the start order (one atomic round trip, ~2 us, at the head of every workgroup: measured 8 % of the kernel).  The workgroup that draws
        // the last ticket puts the counter back to 0 for the next launch.
        unsigned long long prof_t = a.prof != nullptr ? __builtin_readcyclecounter() : 0ull;
        __shared__ uint32_t lds_ticket;
        int frame = (int) blockIdx.y, wg = (int) blockIdx.x; // grid = (workgroups per frame, frames): x runs fastest, the frames follow each other
        if (a.ticket != nullptr) { // wave-uniform
                if (threadIdx.x == 0) {
                        const uint32_t t = atomicAdd(a.ticket, 1u);
                        if (t == gridDim.x * gridDim.y - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lds_ticket = t;
                }
                __syncthreads();
                const uint32_t index = lds_ticket;
                frame = (int) (index / (uint32_t) a.n_wg);
                wg = (int) (index - (uint32_t) frame * (uint32_t) a.n_wg);
        }
        __shared__ uint32_t ac_tab[2][256], dc_tab[2][12];
        // one buffer, two lives: (SRC = 0) the staging rows of the block loads (per wave 32 rows of 8 x 16 B, 144 B apart), then (behind 4 pad
        // words) the private strings, 17 W words, and behind them the segments' windows (kWin words per block + one spare word per lane)
        constexpr int kStageRow = 9; // uint4 per row
        constexpr int kStageWords = 32 * kStageRow * 4; // per wave (SRC = 0)
        // window words per block: 16 = the private strings' size; the 4:2:0 fused kernel takes 12 (384 bits per block on a segment's average, 5 x
        // what a 4K q75 frame needs; beyond: the general path) -- that is what lets a sixth workgroup onto the CU
        constexpr int kWin = SRC == 420 || SRC == 1420 || SRC == 444 ? 12 : kWinWordsPerBlock;
        constexpr int kBufWords = (kPrivStride + kWin + 1) * W + 4; // (+ the pad in front of the private strings: 4 words, the windows stay 16-byte aligned)
        static_assert(kBufWords >= WAVES * kStageWords, "both lives must fit");
        __shared__ __attribute__((aligned(16))) uint32_t buf[kBufWords];
        constexpr int kMaxSeg = W / 3 + 1; // segments per workgroup: a segment has at least 3 blocks (4:4:4, restart interval 1)
        __shared__ int lds_wave_total[WAVES], lds_seg_bits[kMaxSeg], lds_flag[2];
        __shared__ uint32_t lds_base;
        __shared__ int lds_nb[SRC == 0 ? 1 : W]; // fused: the code lengths, by scan index (the lanes hold their blocks in frame order)
        uint32_t *const priv = buf + 4; // buf[3]: where a string's first append rewrites "the word before" (PrivSink)
        uint32_t *const win = priv + kPrivStride * W;
        // per segment, the general path only -- which has no use for the private strings: their place
        uint32_t *const lds_seg_ff = priv, *const lds_seg_off = priv + kMaxSeg, *const lds_seg_done = priv + 2 * kMaxSeg;
        static_assert(3 * kMaxSeg <= kPrivStride * W, "the general path's per-segment words fit the private strings' place");
        // the DC values, then the inclusive bit positions, live in the spare words behind the windows (which only the general path's emission
        // uses, later): with them the 4:2:0 kernel stays under 26 KB, the size at which six workgroups fit a CU's 160 KB
        __shared__ int lds_pos0[SRC == 0 ? W : 1];
        // (SRC = 0 has no barrier between the DC reads and the positions' writes: a place of their own there)
        int *const lds_dc = (int *) (win + W * kWin), *const lds_incl = SRC == 0 ? lds_pos0 : lds_dc;
        const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform by construction: say so, or the waves' branches compile as divergent ones)
        // (the Huffman tables are copied to LDS further down, behind the block loads: nothing ahead of the first barrier needs them, and their
        // memory round trip then runs beside the pixels' instead of in front of it)
        auto init_tables = [&]() {
                for (int i = tid; i < 512; i += W) {
                        const int sym = i & 255;
                        ac_tab[i >> 8][sym] = packed_ac_entry(kAcTab[i >> 8][sym], sym);
                }
                if (tid < 24) dc_tab[tid / 12][tid % 12] = kDcTab[tid / 12][tid % 12];
                if (tid < 2) lds_flag[tid] = 0;
        };
        const int nc = SRC == 0 ? a.nc : 2; // (the fused variants code three components: a constant there)
        const int ybl = a.hs * a.vs, per_mcu = ybl + nc, S = a.S, ri = a.ri;
        // ---- which segments, which block ----
        int seg0, nseg_wg, m0 = 0; // m0 (fused): the workgroup's first MCU, raster order
        if (SRC == 0) {
                seg0 = wg * a.G;
                nseg_wg = min(a.G, a.n_seg - seg0);
        } else {
                // the workgroup's lanes = kMcus consecutive MCUs of the scan (raster order, whatever row they lie in: a run that reaches the end
                // of an MCU row goes on in the next) = kMcus / ri whole segments
                constexpr int kMcus = SRC == 444 ? 64 : 32; // 192 (420, 444) / 128 (422) blocks
                m0 = kMcus * wg;
                seg0 = wg * a.G; // G = kMcus / ri
                nseg_wg = min(a.G, a.n_seg - seg0);
        }
        // which block this lane CODES: `sid`, its index in scan order among the workgroup's blocks.  SRC = 0: the lanes are in scan order.  Fused:
        // every lane codes the block it made -- the lanes are in FRAME order (a wave = a luma block row, or the chroma blocks, of the 32 MCUs;
        // 4:4:4: a component of 64 MCUs) and stay there: what the scan order decides -- the DC predictions, the bit positions -- goes through
        // small LDS arrays indexed by sid, the coefficients never change lanes.  (Round 4's first form handed the blocks over to scan-order lanes
        // through LDS, 128 B per lane and four barriers: a tenth of the workgroup's time; and a wave of chroma blocks alone, or luma blocks alone,
        // skips more of the zero positions than a mixed one.)  The fused variants work the identity out behind their front end -- values that
        // would otherwise sit in registers through the forward DCT, where the 4:2:0 kernel has none to spare at five waves per SIMD
        int sid = tid, sl, j, m_first, n_blk, ml, b, comp;
        bool active;
        auto identify = [&]() {
                sl = div16(sid, a.S_m16); j = sid - sl * S;   // segment of the workgroup, block of the segment
                m_first = (seg0 + sl) * ri;
                n_blk = sl < nseg_wg ? per_mcu * (min(a.n_mcu, m_first + ri) - m_first) : 0;
                active = j < n_blk;
                ml = div16(j, a.per_mcu_m16); b = j - ml * per_mcu; // MCU of the segment, block of the MCU
                comp = b < ybl ? (SRC == 0 ? a.tab0 : 0) : a.ctab;
        };
        uint32_t w[32];
        if (SRC == 0) {
                identify();
                init_tables();
                const int16_t *const cy = a.cy + frame * a.coef_y, *const cb = a.cb + frame * a.coef_c, *const cr = a.cr + frame * a.coef_c;
                const int m = m_first + ml;
                const int my = m / a.mcu_w, mx = m - my * a.mcu_w;
                const int yrow = a.vs * my + (a.hs == 2 ? b >> 1 : 0), ycol = a.hs * mx + (a.hs == 2 ? b & 1 : 0);
                const int16_t *p = b < ybl ? cy + 64 * ((long) yrow * (a.hs * a.mcu_w) + ycol) : (b == ybl ? cb : cr) + 64L * m;
                if (!active) p = cy;
                // A block is one 128-byte line.  If every lane fetched its own block 16 bytes at a time, each of the 8 load
                // instructions of the wave would touch 64 different lines and use an eighth of each: 8x the traffic between L2 and
                // the CU (measured: the kernel then spends two thirds of its time waiting for these loads).  So 8 lanes share a
                // block -- an instruction fetches 8 whole lines -- and the rows are handed to their owners through LDS.
                const long off = (const char *) p - (const char *) cy;
                const int off_lo = (int) off, off_hi = (int) (off >> 32);
                uint4 *const stage = (uint4 *) (buf + wv * kStageWords);
                // two halves of 32 blocks, so that the staging rows take 4.5 KB per wave instead of 9
#pragma unroll
                for (int half = 0; half < 2; half++) {
                        uint4 t[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                                const int owner = 32 * half + (lane >> 3) + 8 * i; // lane (of this wave) that owns the block this lane helps to fetch
                                const unsigned lo = (unsigned) __builtin_amdgcn_ds_bpermute(4 * owner, off_lo);
                                const long hi = __builtin_amdgcn_ds_bpermute(4 * owner, off_hi);
                                t[i] = ((const uint4 *) ((const char *) cy + ((hi << 32) | (long) lo)))[lane & 7];
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) stage[((lane >> 3) + 8 * i) * kStageRow + (lane & 7)] = t[i];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        if ((lane >> 5) == half) {
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                        uint4 r = stage[(lane & 31) * kStageRow + i];
                                        if (!active) r = make_uint4(0, 0, 0, 0); // idle lanes must not keep the wave from skipping zero positions
                                        w[4 * i] = r.x; w[4 * i + 1] = r.y; w[4 * i + 2] = r.z; w[4 * i + 3] = r.w;
                                }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                }
        } else {
                // ---- fused front end: the arithmetic of uyvy_jpeg_fast_kernel (jpeg_fdct.hip), lane = block in FRAME order: the luma waves take
                // one luma block row of the workgroup's 32 MCUs each (64 blocks), the last wave 32 Cb blocks (lanes 0-31) + 32 Cr blocks (lanes 32-63) ----
                constexpr int kLumaWaves = SRC == 420 || SRC == 1420 ? 2 : 1;
                const uint8_t *const src = a.src + (size_t) frame * a.src_stride;
                const int height = a.height, pitch = a.pitch;
                if (SRC == 444) {
                        // packed RGB, components kept as R, G, B (gpujpeg.cpp:303-305): 64 MCUs = 64 8x8 pixel blocks per workgroup; wave c makes the blocks
                        // of component c (the arithmetic of rgb_jpeg444_kernel, jpeg_fdct.hip; the three waves read the same pixels, HBM sees them once)
                        const int mm = m0 + lane;
                        const int by = div32(mm, a.mcu_w_m32), bx = mm - by * a.mcu_w;
                        const bool valid = mm < a.n_mcu;
                        if (valid) {
                                uint32_t raw[8][6];
                                const bool interior = 8 * bx + 8 <= a.width && 8 * by + 8 <= height && !(pitch & 3) && !(3 & (uintptr_t) src);
                                if (interior) {
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                const uint32_t *p = (const uint32_t *) (src + (long) (8 * by + r) * pitch + 24 * bx);
#pragma unroll
                                                for (int k = 0; k < 6; k++) raw[r][k] = p[k];
                                        }
                                } else { // edge replication, byte by byte into the same register layout
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                const uint8_t *row = src + (long) min(8 * by + r, height - 1) * pitch;
#pragma unroll
                                                for (int k = 0; k < 6; k++) raw[r][k] = 0;
#pragma unroll
                                                for (int c = 0; c < 8; c++) {
                                                        const uint8_t *px = row + 3L * min(8 * bx + c, a.width - 1);
#pragma unroll
                                                        for (int comp2 = 0; comp2 < 3; comp2++) {
                                                                const int bi = 3 * c + comp2;
                                                                raw[r][bi >> 2] |= (uint32_t) px[comp2] << (8 * (bi & 3));
                                                        }
                                                }
                                        }
                                }
                                float q[64];
                                auto take = [&](auto comp_c) { // static byte positions: one copy of the code per component, selected per WAVE
                                        constexpr int cc = decltype(comp_c)::value;
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
#pragma unroll
                                                for (int c = 0; c < 8; c++) {
                                                        const int bi = 3 * c + cc;
                                                        q[8 * r + c] = (float) ((raw[r][bi >> 2] >> (8 * (bi & 3))) & 0xff);
                                                }
                                        }
                                };
                                if (wv == 0) take(std::integral_constant<int, 0>{});
                                else if (wv == 1) take(std::integral_constant<int, 1>{});
                                else take(std::integral_constant<int, 2>{});
                                ug_jpeg::fdct8x8(q);
                                ug_jpeg::quant_pack(q, div, w); // R, G and B are all quantised with table 0
                        } else {
#pragma unroll
                                for (int i = 0; i < 32; i++) w[i] = 0;
                        }
                } else if (SRC == 1420) {
                        // planar 4:2:0 (I420: Y, U, V planes back to back, tightly packed; GPUJPEG_420_U8_P0P1P2, gpujpeg.cpp:335): the samples as they are
                        const int cw = (a.width + 1) / 2, ch = (height + 1) / 2;
                        const uint8_t *const up = src + (size_t) a.width * height, *const vp = up + (size_t) cw * ch;
                        float q[64];
                        bool valid;
                        if (wv < kLumaWaves) {
                                const int mm = m0 + (lane >> 1), my = div32(mm, a.mcu_w_m32), mx = mm - my * a.mcu_w;
                                const int bx = 2 * mx + (lane & 1), brow = 2 * my + wv;
                                valid = mm < a.n_mcu;
                                if (valid) {
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                const uint2 v2 = *(const uint2 *) (src + (long) min(8 * brow + r, height - 1) * a.width + 8 * bx);
#pragma unroll
                                                for (int x = 0; x < 4; x++) {
                                                        q[8 * r + x] = (float) ((v2.x >> (8 * x)) & 0xff);
                                                        q[8 * r + 4 + x] = (float) ((v2.y >> (8 * x)) & 0xff);
                                                }
                                        }
                                        ug_jpeg::fdct8x8(q);
                                        ug_jpeg::quant_pack(q, div, w);
                                }
                        } else {
                                const int c = lane >> 5, m = lane & 31; // 0 = Cb, 1 = Cr ; MCU of the workgroup
                                const int mm = m0 + m, my = div32(mm, a.mcu_w_m32), mx = mm - my * a.mcu_w;
                                valid = mm < a.n_mcu;
                                if (valid) {
                                        const uint8_t *const plane = c ? vp : up;
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                const uint2 v2 = *(const uint2 *) (plane + (long) min(8 * my + r, ch - 1) * cw + 8 * mx);
#pragma unroll
                                                for (int x = 0; x < 4; x++) {
                                                        q[8 * r + x] = (float) ((v2.x >> (8 * x)) & 0xff);
                                                        q[8 * r + 4 + x] = (float) ((v2.y >> (8 * x)) & 0xff);
                                                }
                                        }
                                        ug_jpeg::fdct8x8(q);
                                        ug_jpeg::quant_pack(q, div + 64, w);
                                }
                        }
                        if (!valid) {
#pragma unroll
                                for (int i = 0; i < 32; i++) w[i] = 0;
                        }
                } else {
                        float q[64];
                        bool valid; // lanes past the last MCU of the picture hold no block
                        if (wv < kLumaWaves) {
                                const int mm = m0 + (lane >> 1), my = div32(mm, a.mcu_w_m32), mx = mm - my * a.mcu_w; // this lane's MCU
                                const int bx = 2 * mx + (lane & 1);        // luma block column
                                const int brow = kLumaWaves * my + wv;     // luma block row
                                valid = mm < a.n_mcu;
                                if (valid) {
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                const int y = min(8 * brow + r, height - 1);
                                                const uint4 v4 = *(const uint4 *) (src + (long) y * pitch + 16 * bx);
                                                const uint32_t ww[4] = { v4.x, v4.y, v4.z, v4.w };
#pragma unroll
                                                for (int k = 0; k < 4; k++) {
                                                        q[8 * r + 2 * k] = (float) ((ww[k] >> 8) & 0xff);
                                                        q[8 * r + 2 * k + 1] = (float) (ww[k] >> 24);
                                                }
                                        }
                                        ug_jpeg::fdct8x8(q);
                                        ug_jpeg::quant_pack(q, div, w);
                                        // (the two front ends end differently on purpose: left alike, the compiler sinks their common tail -- forward DCT
                                        // and quantiser -- behind the branch with the quantiser table chosen at run time, 128 scalar loads from computed
                                        // addresses and SGPR spills: 1 890 instructions per wave instead of 1 340)
                                        asm volatile("; luma blocks made" : "+v"(w[31]));
                                }
                        } else {
                                const int c = lane >> 5, m = lane & 31; // 0 = Cb, 1 = Cr ; MCU of the workgroup
                                const int mm = m0 + m, my = div32(mm, a.mcu_w_m32), mx = mm - my * a.mcu_w;
                                valid = mm < a.n_mcu;
                                if (valid) {
#pragma unroll
                                        for (int r = 0; r < 8; r++) {
                                                int y0, y1;
                                                if (SRC == 420) {
                                                        const int cy2 = min(8 * my + r, (height + 1) / 2 - 1); // edge replication on the chroma plane
                                                        y0 = 2 * cy2; y1 = min(2 * cy2 + 1, height - 1);             // odd height: last line doubled
                                                } else {
                                                        y0 = y1 = min(8 * my + r, height - 1);
                                                }
                                                // this lane's half of the MCU's 32-byte row piece (Cb lanes the first 16 bytes, Cr lanes the second)
                                                const uint4 a0 = *(const uint4 *) (src + (long) y0 * pitch + 32 * mx + 16 * c);
                                                uint32_t wa[4] = { a0.x, a0.y, a0.z, a0.w };
                                                if (SRC == 420) { // (a + b + 1) / 2 of uyvy_to_i420 (to_planar.c:364-367), all four bytes of a word at once
                                                        const uint4 c0 = *(const uint4 *) (src + (long) y1 * pitch + 32 * mx + 16 * c);
                                                        wa[0] = ug_jpeg::avg_bytes(wa[0], c0.x); wa[1] = ug_jpeg::avg_bytes(wa[1], c0.y);
                                                        wa[2] = ug_jpeg::avg_bytes(wa[2], c0.z); wa[3] = ug_jpeg::avg_bytes(wa[3], c0.w);
                                                } // else uyvy_to_i422 (video_codec.c:949-969): samples as they are
                                                ug_jpeg::chroma_row_from_uyvy(wa, q + 8 * r);
                                        }
                                        ug_jpeg::fdct8x8(q);
                                        ug_jpeg::quant_pack(q, div + 64, w);
                                        asm volatile("; chroma blocks made" : "+v"(w[31]));
                                }
                        }
                        if (!valid) {
#pragma unroll
                                for (int i = 0; i < 32; i++) w[i] = 0;
                        }
                }
                init_tables();
                sid = SRC == 444 ? 3 * lane + wv
                                 : (wv < kLumaWaves ? per_mcu * (lane >> 1) + (kLumaWaves == 2 ? 2 * wv : 0) + (lane & 1) : per_mcu * (lane & 31) + ybl + (lane >> 5));
                identify();
                UG_PHASE(0) // pixels -> quantised block (wave 0's view, like all the marks)
        }
        // DC difference: the previous block of the same component sits `back` lanes below (luma: the previous luma block of the
        // scan -- one lane back, or across the two chroma blocks of the previous MCU; chroma: one MCU back); none at a segment start
        const int dc = coef_at(w, 0);
        lds_dc[sid] = dc;
        __syncthreads(); // tables, DC values; (SRC = 0) the staging rows have been read
        UG_PHASE(1) // tables + DC values (SRC = 0: the block loads)
        const int back = b < ybl ? (b > 0 ? 1 : 1 + nc) : per_mcu;
        const bool has_pred = b < ybl ? j > 0 : ml > 0;
        const int diff = dc - (has_pred ? lds_dc[max(sid - back, 0)] : 0);
        const uint32_t dneg = (uint32_t) (diff >> 31), da = ((uint32_t) diff ^ dneg) - dneg;
        const uint32_t dsize = 32u - (uint32_t) __clz((int) da);
        const uint32_t de = dc_tab[comp][dsize];
        const uint32_t zrl = kAcTab[comp][0xF0], eob = kAcTab[comp][0x00];
        const uint32_t *const tab = ac_tab[comp];
        const uint32_t dc_vb = ((uint32_t) diff + dneg) & ((1u << dsize) - 1u);
        const uint32_t dc_str = active ? ((de & 0xffffu) << dsize) | dc_vb : 0u, dc_n = active ? (de >> 16) + dsize : 0u;
        // the windows start out zero (this lane's 16 words; SRC = 0: the staging rows they overlay have been read: the barrier above)
        {
                uint4 *const z = (uint4 *) (win + tid * kWin);
#pragma unroll
                for (int i = 0; i < kWin / 4; i++) z[i] = make_uint4(0, 0, 0, 0);
        }
        // ---- the walk: code into the private string, length as a by-product ----
        uint32_t *const row = priv + tid * kPrivStride;
        uint32_t nbits = 0;
        if (active) { // idle lanes stay out of the walk
                const uint32_t first = 4u + (uint32_t) tid * kPrivStride;
                PrivSink sink = { 0ull, 32u * first - 32u, buf, first };
                sink.append<false>(dc_str, dc_n);
                walk_private(w, tab, zrl, eob, sink);
                nbits = sink.bits();
                sink.finish();
                if (nbits > 32u * kPrivWords) lds_flag[0] = 1; // does not fit its private string: the general path for this workgroup
        }
        UG_PHASE(2) // the walk
        // ---- bit position of every block inside its segment: prefix sum over the workgroup, made segment-relative ----
        int len_scan = (int) nbits; // the length of the block whose scan index is tid
        if (SRC != 0) {
                lds_nb[sid] = (int) nbits;
                __syncthreads();
                len_scan = lds_nb[tid];
        }
        const int incl_w = wave_inclusive_scan(len_scan, lane);
        lds_incl[tid] = incl_w; // inclusive bit position inside the wave, by scan index (fused: the DC values were read a barrier ago)
        if (lane == 63) lds_wave_total[wv] = incl_w;
        __syncthreads();
        int wave_total[WAVES];
#pragma unroll
        for (int k = 0; k < WAVES; k++) wave_total[k] = lds_wave_total[k];
        auto incl_at = [&](int i) { // inclusive bit position of scan index i in the workgroup: the waves in front are added by whoever asks
                int v = lds_incl[i];
#pragma unroll
                for (int k = 0; k < WAVES - 1; k++) v += k < (i >> 6) ? wave_total[k] : 0;
                return v;
        };
        const int excl = incl_at(sid) - (int) nbits;
        const int first_tid = min(sl * S, W - 1); // scan index of the segment's first block
        const int seg_base = first_tid ? incl_at(first_tid - 1) : 0;
        const int seg_bits = sl < nseg_wg ? incl_at(min(first_tid + max(n_blk, 1) - 1, W - 1)) - seg_base : 0; // total bits of this lane's segment
        if (j == 0 && sl < nseg_wg) lds_seg_bits[sl] = seg_bits;
        uint32_t *const mywin = win + first_tid * kWin; // the segment's window: kWin words per block of the segment
        const int cap = S * kWin;                               // words of a segment's window
        const int seg_words = (seg_bits + 31) >> 5;
        const int p0 = excl - seg_base;                          // bit position of this lane's block in its segment
        if (seg_words > cap) { // a segment beyond its window (possible where the windows are smaller than the private strings): general path, several passes
                lds_flag[0] = 1;
                atomicMax(&lds_flag[1], seg_words);
        }
        if (kWin < kWinWordsPerBlock) __syncthreads(); // (with full-size windows no segment of blocks that fit their strings can exceed them)
        const bool general = lds_flag[0] != 0;
        // zero the words the segment uses in a pass (+ one for the padding), cooperatively: lane j takes words j, j + S, ...
        auto zero_window = [&](int lo_idx) {
                if (sl < nseg_wg) {
                        const int nw = min(cap, seg_words + 1 - lo_idx);
                        for (int i = j; i < nw; i += S) mywin[i] = 0;
                }
                __syncthreads();
        };
        // the general path's emission of one pass: straight into the windows, which show words [lo_idx, lo_idx + cap) of every segment
        auto emit_general = [&](int lo_idx) {
                // everything the walk derives from the coefficients is invariant over the passes; left alone, the compiler
                // hoists all of it out of the pass loops (4 values x 63 coefficients: 237 VGPRs for a path that almost never runs)
#pragma unroll
                for (int i = 0; i < 32; i++) asm volatile("" : "+v"(w[i]));
                zero_window(lo_idx);
                if (active) { // words outside this pass's window go to a spare word of the lane's own (no hot spot)
                        BitSink<true> sink = { 0, 0, (uint32_t) p0 & 31u, mywin, (uint32_t) (p0 >> 5), (uint32_t) lo_idx, (uint32_t) cap, win + W * kWin + tid };
                        sink.append(dc_str, dc_n);
                        walk_block(w, tab, zrl, eob, sink);
                        sink.finish();
                }
                __syncthreads();
        };
        // pad the last byte of segment s2 with 1-bits (T.81 F.1.2.3) if this pass's window shows it (the wave that reads the segment afterwards)
        auto pad_segment = [&](int s2, int lo_idx) {
                const int bits = lds_seg_bits[s2];
                const int padw = (bits >> 5) - lo_idx; // window word that holds the last, partial byte
                if (lane == 0 && (bits & 7) && padw >= 0 && padw < cap) {
                        const int pad = 8 - (bits & 7);
                        win[s2 * S * kWin + padw] |= ((1u << pad) - 1u) << (32 - (bits & 31) - pad);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
        };
        // count the 0xFF bytes of the window words of this pass
        auto count_pass = [&](int lo_idx) {
#pragma unroll 1
                for (int s2 = wv; s2 < nseg_wg; s2 += WAVES) {
                        pad_segment(s2, lo_idx);
                        const int bits = lds_seg_bits[s2];
                        const uint32_t *const sw = win + s2 * S * kWin;
                        const int nbytes = (bits + 7) >> 3;
                        const int nw = min(cap, ((nbytes + 3) >> 2) - lo_idx); // words of the segment in this pass
                        int ff = 0;
                        for (int i = lane; i < nw; i += 64) ff += count_ff_valid(sw[i], min(4, nbytes - 4 * (lo_idx + i)));
                        const int ff_all = __builtin_amdgcn_readlane(wave_inclusive_scan(ff, lane), 63);
                        if (lane == 0) lds_seg_ff[s2] += (uint32_t) ff_all;
                }
        };
        // final sizes -> positions of the segments inside the workgroup's stretch, the stretch's position in the stream (first wave)
        auto place = [&]() {
                __syncthreads();
                if (wv == 0) {
                        uint32_t carry = 0;
#pragma unroll 1
                        for (int base = 0; base < nseg_wg; base += 64) {
                                const int s2 = base + lane;
                                const int sz = s2 < nseg_wg ? ((lds_seg_bits[s2] + 7) >> 3) + (int) lds_seg_ff[s2] + 2 : 0; // stuffed bytes + marker
                                const int inc = wave_inclusive_scan(sz, lane);
                                if (s2 < nseg_wg) lds_seg_off[s2] = carry + (uint32_t) (inc - sz);
                                carry += (uint32_t) __builtin_amdgcn_readlane(inc, 63);
                        }
                        if (a.slots != nullptr) { // wave-uniform
                                if (lane == 0) {
                                        a.wg_bytes[(size_t) frame * a.n_wg + wg] = carry;
                                        lds_base = 0;
                                        if ((size_t) carry > a.slot_bytes) a.total_pinned[kMaxBatch + 1] = 1u; // does not fit its slot: the host runs the call again, with the look-back
                                }
                        } else {
                                const uint32_t before = a.flat ? lookback_exclusive_flat<8>(a.status + (long) frame * a.n_status, wg, carry, a.gen, lane, a.total_pinned + kMaxBatch)
                                                               : lookback_exclusive(a.status + (long) frame * a.n_status, wg, carry, a.gen, lane, a.total_pinned + kMaxBatch);
                                if (lane == 0) lds_base = (uint32_t) a.header_len + before;
                        }
                }
                __syncthreads();
        };
        const uint32_t base0 = SRC == 0 && a.base != nullptr ? a.base[frame] - 2u : 0u; // (wave-uniform: a scalar load)
        uint8_t *const out = a.slots != nullptr ? a.slots + ((size_t) frame * a.n_wg + wg) * a.slot_bytes : a.out + (size_t) frame * a.out_stride + base0;
        const size_t capacity = a.slots != nullptr ? a.slot_bytes : (a.capacity > base0 ? a.capacity - base0 : 0);
        // the window words of this pass to their place in the stream, 0x00 after every 0xFF; after the last pass RSTm / EOI
        auto write_pass = [&](int lo_idx, bool last_pass, bool pad) {
#pragma unroll 1
                for (int s2 = wv; s2 < nseg_wg; s2 += WAVES) {
                        if (pad) pad_segment(s2, lo_idx); // (a pass that was emitted again for the write-out)
                        const int sg = seg0 + s2;
                        const int bits = lds_seg_bits[s2];
                        const uint32_t *const sw = win + s2 * S * kWin;
                        const int nbytes = (bits + 7) >> 3;
                        const uint32_t start = lds_base + lds_seg_off[s2], end = start + (uint32_t) nbytes + lds_seg_ff[s2] + 2u;
                        const bool fits = (size_t) end <= capacity; // would not fit: nothing of this segment is written, the host reports the needed size
                        const uint32_t done = lds_seg_done[s2];
                        uint8_t *const d = out + start + done;
                        const int nw = min(cap, ((nbytes + 3) >> 2) - lo_idx);
                        uint32_t written = 0; // wave-uniform
                        for (int i0 = 0; i0 < nw; i0 += 64) {
                                const int i = i0 + lane;
                                const uint32_t word = i < nw ? sw[i] : 0u;
                                const int valid = i < nw ? min(4, nbytes - 4 * (lo_idx + i)) : 0;
                                const int mine = valid + count_ff_valid(word, valid);
                                const int inc = wave_inclusive_scan(mine, lane);
                                if (fits) {
                                        uint8_t *p = d + written + (uint32_t) (inc - mine);
#pragma unroll
                                        for (int t = 0; t < 4; t++) {
                                                if (t < valid) {
                                                        const uint8_t byte = (uint8_t) (word >> (24 - 8 * t));
                                                        *p++ = byte;
                                                        if (byte == 0xFF) *p++ = 0;
                                                }
                                        }
                                }
                                written += (uint32_t) __builtin_amdgcn_readlane(inc, 63);
                        }
                        if (lane == 0) {
                                lds_seg_done[s2] = done + written;
                                if (last_pass) {
                                        if (fits) {
                                                out[end - 2] = 0xFF;
                                                out[end - 1] = sg == a.n_seg - 1 ? 0xD9 : (uint8_t) (0xD0 + (sg & 7));
                                        }
                                        if (sg == a.n_seg - 1 && a.slots == nullptr) a.total_pinned[frame] = base0 + end;
                                }
                        }
                }
        };
        if (wg == 0 && a.slots == nullptr) { // the first workgroup also lays down SOI .. SOS
                for (int i = tid; i < a.header_len; i += W) out[i] = a.header[i];
        }
        UG_PHASE(3) // positions (two barriers: the other waves' walks end here)
        if (__builtin_expect(!general, 1)) {
                if (active) { // the private string, shifted to the block's bit position, into the segment's window
                        const uint32_t sh = (uint32_t) p0 & 31u;
                        uint32_t *const dst = mywin + (p0 >> 5);
                        const int nw = (int) ((nbits + 31u) >> 5), nsw = (int) ((sh + nbits + 31u) >> 5);
                        uint32_t prev = 0;
                        for (int k = 0; __ballot(k < nsw) != 0; k++) {
                                const uint32_t cur = k < nw ? row[k] : 0u;
                                const uint32_t o = __builtin_amdgcn_alignbit(prev, cur, sh); // ({prev, cur} >> sh): the word at k of the shifted string
                                if (k < nsw) atomicOr(&dst[k], o);
                                prev = cur;
                        }
                }
                __syncthreads();
                UG_PHASE(4) // merge
                // ---- byte stuffing and write-out, every lane at once: the S lanes of a segment's stretch of the workgroup take K consecutive
                // words of its window each (K = words / S rounded up; the lanes of the segment's blocks or not: this is byte work).  One pass
                // counts the bytes the lane will write (its bytes + one 0x00 per 0xFF; the segment's last lane + 2 for RSTm / EOI); ONE prefix sum
                // over the workgroup turns the counts into positions in the workgroup's stretch of the stream -- segment sizes, segment
                // offsets and the stretch's length all fall out of it --; a second pass over the same words writes.  (Rounds 2-4 went through
                // the segments one by one, a wave each: a count with a wave scan per segment, a scan over the segments, a write-out with
                // another wave scan per 64 words -- a third of the instructions behind the walk.)  The 1-bits that pad a segment's last byte
                // (T.81 F.1.2.3) are ORed in where the word is read.
                // (a partition of its own, by tid: lane tid = segment tid / S, part tid % S -- whatever block the lane coded)
                const int wsl = div16(tid, a.S_m16), wj = tid - wsl * S;
                const int wbits = wsl < nseg_wg ? lds_seg_bits[wsl] : 0; // (written before the merge's barrier)
                const uint32_t *const wwin = win + min(wsl * S, W - 1) * kWin;
                const int nbytes = (wbits + 7) >> 3, nwords = (nbytes + 3) >> 2; // (0 for lanes behind the workgroup's last segment)
                const int K = SRC == 0 && S == 1 ? nwords : div32(nwords + S - 1, a.S_m32), i_first = wj * K; // (S = 1 -- a one-component scan with restart interval 1 -- has no 32-bit reciprocal)
                const int padw = wbits >> 5, padn = 8 - (wbits & 7);
                const uint32_t padmask = (wbits & 7) ? ((1u << padn) - 1u) << (32 - (wbits & 31) - padn) : 0u;
                const bool seg_last = wsl < nseg_wg && wj == S - 1; // the lane that writes the marker behind the segment
                uint32_t mine = seg_last ? 2u : 0u;
                for (int k = 0; k < K; k++) {
                        const int i = i_first + k;
                        if (i < nwords) {
                                const int valid = min(4, nbytes - 4 * i);
                                mine += (uint32_t) (valid + count_ff_valid(wwin[i] | (i == padw ? padmask : 0u), valid));
                        }
                }
                const uint32_t incl_b = (uint32_t) wave_inclusive_scan((int) mine, lane);
                if (lane == 63) lds_wave_total[wv] = (int) incl_b; // (the bit totals were read two barriers ago)
                __syncthreads();
                uint32_t before_w = 0, stretch = 0;
#pragma unroll
                for (int k = 0; k < WAVES; k++) {
                        const uint32_t t = (uint32_t) lds_wave_total[k];
                        before_w += k < wv ? t : 0u;
                        stretch += t;
                }
                UG_PHASE(5) // byte counts + prefix sum
                uint32_t base = 0;
                if (a.slots != nullptr) { // wave-uniform: the stretch goes to the workgroup's slot, the gather kernel places it
                        if (tid == 0) {
                                a.wg_bytes[(size_t) frame * a.n_wg + wg] = stretch;
                                if ((size_t) stretch > a.slot_bytes) a.total_pinned[kMaxBatch + 1] = 1u; // does not fit its slot: the host runs the call again, with the look-back
                        }
                } else {
                        if (wv == 0) {
                                const uint32_t before = a.flat ? lookback_exclusive_flat<8>(a.status + (long) frame * a.n_status, wg, stretch, a.gen, lane, a.total_pinned + kMaxBatch)
                                                               : lookback_exclusive(a.status + (long) frame * a.n_status, wg, stretch, a.gen, lane, a.total_pinned + kMaxBatch);
                                if (lane == 0) lds_base = (uint32_t) a.header_len + before;
                        }
                        __syncthreads();
                        base = lds_base;
                }
                UG_PHASE(6) // the stretch's place (look-back)
                const uint32_t at = base + before_w + incl_b - mine; // this lane's first byte
                if ((size_t) at + mine <= capacity) { // (what would not fit is not written: the host reports the size the stream needs)
                        uint8_t *p = out + at;
                        for (int k = 0; k < K; k++) {
                                const int i = i_first + k;
                                if (i < nwords) {
                                        const int valid = min(4, nbytes - 4 * i);
                                        const uint32_t word = wwin[i] | (i == padw ? padmask : 0u);
#pragma unroll
                                        for (int t = 0; t < 4; t++) {
                                                if (t < valid) {
                                                        const uint8_t byte = (uint8_t) (word >> (24 - 8 * t));
                                                        *p++ = byte;
                                                        if (byte == 0xFF) *p++ = 0;
                                                }
                                        }
                                }
                        }
                        if (seg_last) {
                                const int sg = seg0 + wsl;
                                p[0] = 0xFF;
                                p[1] = sg == a.n_seg - 1 ? 0xD9 : (uint8_t) (0xD0 + (sg & 7));
                        }
                }
                if (seg_last && seg0 + wsl == a.n_seg - 1 && a.slots == nullptr) a.total_pinned[frame] = base0 + at + mine; // the stream's length
                UG_PHASE(7) // write-out
                if (a.prof != nullptr && threadIdx.x == 0) a.prof[((size_t) blockIdx.y * gridDim.x + blockIdx.x) * (kProfPhases + 1) + kProfPhases] = 1ull;
        } else {
                asm volatile("; general path" ::: "memory");
                __syncthreads(); // lds_flag[1]; nobody reads a private string from here on
                if (tid < kMaxSeg) { // (ordered before their first use by the barrier inside emit_general)
                        lds_seg_ff[tid] = 0;
                        lds_seg_done[tid] = 0;
                }
                const int longest = lds_flag[1]; // 0: every segment fits its window
                const int passes = longest ? (longest + cap - 1) / cap : 1;
#pragma unroll 1
                for (int pass = 0; pass < passes; pass++) {
                        emit_general(pass * cap);
                        count_pass(pass * cap);
                        __syncthreads();
                }
                place();
                if (passes == 1) {
                        write_pass(0, true, false); // the windows still hold the one pass
                } else {
#pragma unroll 1
                        for (int pass = 0; pass < passes; pass++) {
                                emit_general(pass * cap);
                                write_pass(pass * cap, pass == passes - 1, true);
                                __syncthreads();
                        }
                }
        }
}

// The second launch of the two-launch placement: one wave per workgroup of the coder.  Its stretch starts where the stretches before it end (the
// byte counts of a frame's workgroups: 4 KB, summed by the wave itself); the bytes come out of the 16-byte aligned slot and go to an arbitrary
// byte address: whole destination words assembled from two source words (v_alignbyte), the ragged ends byte by byte.
// Why two launches: in one launch every workgroup waits ~5 us (of its ~20) for its position -- a cross-CU hand-off costs ~3 us under load, in the
// consumer's memory queue (MI355X_MICROARCH.md "handoff-1to1") --, with its LDS held all the while; the kernel boundary orders the same data for free.
__global__ __launch_bounds__(256) void jpeg_gather_kernel(const uint8_t *__restrict__ slots, size_t slot_bytes, const uint32_t *__restrict__ wg_bytes, int n_wg,
                                                          uint8_t *__restrict__ out, size_t out_stride, size_t capacity, const uint8_t *__restrict__ header,
                                                          int header_len, uint32_t *__restrict__ total_pinned)
{
        const int frame = blockIdx.y, lane = threadIdx.x & 63;
        const int wg = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
        out += (size_t) frame * out_stride;
        if (blockIdx.x == 0) { // SOI .. SOS
                for (int i = threadIdx.x; i < header_len; i += 256) out[i] = header[i];
        }
        if (wg >= n_wg) return; // wave-uniform
        const uint32_t *const sizes = wg_bytes + (size_t) frame * n_wg;
        const uint8_t *const s = slots + ((size_t) frame * n_wg + wg) * slot_bytes;
        const uint32_t *const sw = (const uint32_t *) s;
        // everything the wave needs from memory is requested before anything is waited for: its own byte count, the counts before it, and the
        // first kAhead x 64 words of its slot with their right-hand neighbours (a slot is larger than that whatever the count turns out to be) --
        // one memory round trip instead of three dependent ones
        constexpr int kAhead = 5; // 1280 bytes: a 4K q75 stretch is ~1.4 KB
        uint32_t lo[kAhead], hi[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; k++) {
                lo[k] = sw[lane + 64 * k];
                hi[k] = sw[lane + 64 * k + 1];
        }
        const uint32_t n = sizes[wg];
        int sum = 0;
#pragma unroll 1
        for (int base = 0; base < wg; base += 512) { // eight loads in flight per lane (one at a time: up to 17 dependent L2 round trips per wave)
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                        const int i = base + lane + 64 * k;
                        v[k] = i < wg ? sizes[i] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) sum += (int) v[k];
        }
        const uint32_t off = (uint32_t) header_len + (uint32_t) __builtin_amdgcn_readlane(wave_inclusive_scan(sum, lane), 63), end = off + n;
        if (wg == n_wg - 1 && lane == 0) total_pinned[frame] = end; // pinned host memory mapped into the device: the length needs no copy back
        if ((size_t) end > capacity || (size_t) n > slot_bytes) return; // would not fit: the host reports the needed size from the total
        uint8_t *const d = out + off;
        const uint32_t head = min(n, (uint32_t) ((4u - ((uintptr_t) d & 3u)) & 3u)); // bytes up to the first aligned destination word
        if ((uint32_t) lane < head) d[lane] = s[lane];
        const uint32_t rem = n - head, nd = rem >> 2, r = head & 3u;
        uint32_t *const dw = (uint32_t *) (d + head);
        // destination word i = slot bytes head + 4 i .. + 3 = ({sw[i + 1], sw[i]} >> 8 head)  (head < 4)
#pragma unroll
        for (int k = 0; k < kAhead; k++) {
                const uint32_t i = lane + 64 * k;
                if (i < nd) dw[i] = __builtin_amdgcn_alignbyte(hi[k], lo[k], r);
        }
        for (uint32_t i = lane + 64 * kAhead; i < nd; i += 64) dw[i] = __builtin_amdgcn_alignbyte(sw[i + 1], sw[i], r);
        const uint32_t tail = rem & 3u;
        if ((uint32_t) lane < tail) d[head + 4 * nd + lane] = s[head + 4 * nd + lane];
}

// one wave per segment: find its position (see kChunk), move its bytes there, inserting 0x00 after every 0xFF (T.81 B.1.1.5), then
// append RSTm (or EOI after the last segment).  The wave of the last segment also reports the stream length.  (Round 4: the companion of
// the wave-per-segment coder only -- long restart intervals; the block-parallel coder places its bytes itself.)
__global__ __launch_bounds__(256) void compact_kernel(const uint8_t *__restrict__ raw, int cap_bytes, const uint32_t *__restrict__ seg_len,
                                                      const uint32_t *__restrict__ seg_ff, const uint32_t *__restrict__ chunk_tot,
                                                      int n_seg, uint8_t *__restrict__ out,
                                                      const uint8_t *__restrict__ header, int header_len, size_t capacity,
                                                      uint32_t *__restrict__ total_pinned, const uint32_t *__restrict__ base /* see CodeArgs::base */, BatchStride bs)
{
        raw += blockIdx.y * bs.raw_words * 4; seg_len += blockIdx.y * bs.seg; seg_ff += blockIdx.y * bs.seg;
        chunk_tot += blockIdx.y * bs.tot_words; out += blockIdx.y * bs.out_bytes; total_pinned += blockIdx.y;
        const uint32_t base0 = base != nullptr ? base[blockIdx.y] - 2u : 0u; // a later scan of a non-interleaved stream goes on over the EOI of the one before
        out += base0;
        capacity = capacity > base0 ? capacity - base0 : 0;
        if (blockIdx.x == 0 && (size_t) header_len <= capacity) { // the first workgroup also lays down SOI .. SOS (a later scan: its SOS)
                for (int i = threadIdx.x; i < header_len; i += 256) out[i] = header[i];
        }
        const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (seg >= n_seg) return;
        uint32_t before = 0; // per lane; reduced below
        const int chunk = seg / kChunk;
        for (int i = lane; i < chunk; i += 64) before += chunk_tot[i * kChunkStride];
        for (int i = chunk * kChunk + lane; i < seg; i += 64) before += seg_ff[i];
        const uint32_t off = (uint32_t) header_len + (uint32_t) __builtin_amdgcn_readlane(wave_inclusive_scan((int) before, lane), 63);
        const uint32_t end = off + seg_ff[seg];
        if (seg == n_seg - 1 && lane == 0) *total_pinned = base0 + end; // pinned host memory mapped into the device: the length needs no copy back
        if ((size_t) end > capacity) return; // would not fit: the host reports the needed size from the total
        const uint8_t *s = raw + (size_t) seg * cap_bytes;
        uint8_t *d = out + off;
        const uint32_t n = seg_len[seg];
        uint32_t extra = 0; // zeros inserted so far (wave-uniform)
        for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < n;
                const uint8_t b = valid ? s[i] : 0;
                const unsigned long long ffm = __ballot(valid && b == 0xFF);
                const uint32_t bef = (uint32_t) __builtin_popcountll(ffm & ((1ull << lane) - 1ull));
                if (valid) {
                        d[i + extra + bef] = b;
                        if (b == 0xFF) d[i + extra + bef + 1] = 0;
                }
                extra += (uint32_t) __builtin_popcountll(ffm);
        }
        if (lane == 0) {
                d[n + extra] = 0xFF;
                d[n + extra + 1] = seg == n_seg - 1 ? 0xD9 : (uint8_t) (0xD0 + (seg & 7));
        }
}

// ---- a scan WITHOUT restart intervals (restart_interval 0), in parallel ----------------------------------------------------------------------------------------
// One segment was one wave walking the frame block after block (entropy_wave_kernel with a grid of one wave): 40 ms for a 1080p frame, 159 ms at 4K.  Nothing in
// Huffman CODING is sequential but where the bits go: a lane per block works out its block's bit count (the DC prediction of a block is the DC value of the block before
// it in the component, which is in the coefficient planes already), a prefix sum turns the counts into bit positions, a lane per block codes its bits to that position
// of a zeroed buffer (atomic OR: blocks share words at their ends), and byte stuffing is one more count / prefix sum / move over 64-byte pieces of the buffer.  The
// stream is the sequential coder's, bit for bit (tests/test_jpeg_colour_options.py::test_gpu_no_restart_intervals).
struct NoriScan {
        const int16_t *c0, *c1, *c2; // component 0 (hs x vs blocks per MCU), the components behind it (nc = 2) -- or one component alone (nc = 0: a scan of a non-interleaved stream)
        int mcu_w, n_mcu, hs, vs, nc, tab0, ctab;
        int ri; // MCUs per restart interval (the predictions start from 0 there); n_mcu: none
};

// block b of the scan -> its coefficients, Huffman table set, and the DC value its prediction starts from
__device__ __forceinline__ const int16_t *nori_block(const NoriScan &s, uint32_t b, int &tab, int &pred)
{
        const int ybl = s.hs * s.vs, per_mcu = ybl + s.nc;
        const int m = (int) (b / (uint32_t) per_mcu), j = (int) (b - (uint32_t) m * per_mcu);
        auto luma = [&](int mm, int jj) {
                const int my = mm / s.mcu_w, mx = mm - my * s.mcu_w;
                const int yrow = s.vs * my + (s.hs == 2 ? jj >> 1 : 0), ycol = s.hs * mx + (s.hs == 2 ? jj & 1 : 0);
                return s.c0 + 64 * ((long) yrow * (s.hs * s.mcu_w) + ycol);
        };
        if (j < ybl) {
                tab = s.tab0;
                pred = j > 0 ? luma(m, j - 1)[0] : (m % s.ri ? luma(m - 1, ybl - 1)[0] : 0);
                return luma(m, j);
        }
        const int16_t *const plane = j == ybl ? s.c1 : s.c2;
        tab = s.ctab;
        pred = m % s.ri ? plane[64L * (m - 1)] : 0;
        return plane + 64L * m;
}

// the symbols of one block, in order: put(bits, n) for every code and every run of value bits
template <class Put>
__device__ __forceinline__ void nori_walk(const int16_t *__restrict__ c, int tab, int pred, Put put)
{
        auto value = [&](int v, int &size) -> uint32_t {
                const int a = v < 0 ? -v : v;
                size = a ? 32 - __builtin_clz((unsigned) a) : 0;
                return (uint32_t) (v < 0 ? v + (1 << size) - 1 : v) & ((1u << size) - 1);
        };
        int size;
        const uint32_t dbits = value((int) c[0] - pred, size);
        const uint32_t de = kDcTab[tab][size];
        put(de & 0xffff, (int) (de >> 16));
        if (size) put(dbits, size);
        int run = 0;
        for (int k = 1; k < 64; k++) {
                const int v = c[k];
                if (v == 0) {
                        run++;
                        continue;
                }
                for (; run > 15; run -= 16) {
                        const uint32_t z = kAcTab[tab][0xF0];
                        put(z & 0xffff, (int) (z >> 16));
                }
                const uint32_t vb = value(v, size);
                const uint32_t e = kAcTab[tab][(run << 4) | size];
                put(e & 0xffff, (int) (e >> 16));
                put(vb, size);
                run = 0;
        }
        if (run) {
                const uint32_t e = kAcTab[tab][0x00];
                put(e & 0xffff, (int) (e >> 16));
        }
}

__global__ __launch_bounds__(256) void nori_len_kernel(NoriScan s, uint32_t n_blocks, uint32_t *__restrict__ bits)
{
        const uint32_t b = blockIdx.x * 256 + threadIdx.x;
        if (b >= n_blocks) return;
        int tab, pred;
        const int16_t *c = nori_block(s, b, tab, pred);
        uint32_t n = 0;
        nori_walk(c, tab, pred, [&](uint32_t, int len) { n += (uint32_t) len; });
        bits[b] = n;
}

// v[0 .. n) -> the sums in front of every entry, v[n] = the total (one workgroup, 8 entries per lane)
__global__ __launch_bounds__(1024) void nori_scan_kernel(uint32_t *__restrict__ v, uint32_t n)
{
        constexpr int kPer = 8;
        __shared__ uint32_t wave_tot[16];
        __shared__ uint32_t carry_s;
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 1024 * kPer) {
                uint32_t x[kPer], sum = 0;
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        const uint32_t i = i0 + (uint32_t) tid * kPer + e;
                        x[e] = i < n ? v[i] : 0;
                        sum += x[e];
                }
                uint32_t incl = sum;
                for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += o;
                }
                if (lane == 63) wave_tot[wv] = incl;
                __syncthreads();
                uint32_t before = carry_s + incl - sum;
                for (int w = 0; w < wv; w++) before += wave_tot[w];
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        const uint32_t i = i0 + (uint32_t) tid * kPer + e;
                        if (i < n) v[i] = before;
                        before += x[e];
                }
                __syncthreads();
                if (tid == 1023) carry_s = before;
                __syncthreads();
        }
        if (tid == 0) v[n] = carry_s;
}

// raw: zeroed; the bits of block b go to bit position pos[b] (stream order: the first bit is the top bit of byte 0) of its segment's region (seg_blocks blocks per
// segment, regions seg_words apart; one segment: seg_blocks = n_blocks); the last block of a segment also pads its last byte with 1-bits (F.1.2.3)
__global__ __launch_bounds__(256) void nori_emit_kernel(NoriScan s, uint32_t n_blocks, const uint32_t *__restrict__ pos, uint32_t *__restrict__ raw, uint32_t seg_blocks, uint32_t seg_words)
{
        const uint32_t b = blockIdx.x * 256 + threadIdx.x;
        if (b >= n_blocks) return;
        int tab, pred;
        const int16_t *c = nori_block(s, b, tab, pred);
        const uint32_t sg = b / seg_blocks;
        raw += (size_t) sg * seg_words;
        uint32_t at = pos[b];              // the next bit
        unsigned long long acc = 0;        // bits waiting, left-aligned behind the `at & 31` bits of the word that belong to whoever came before
        int fill = (int) (at & 31);
        uint32_t word = at >> 5;
        auto flush = [&]() { // the top 32 bits are a word's worth
                atomicOr(raw + word, __builtin_bswap32((uint32_t) (acc >> 32)));
                acc <<= 32;
                fill -= 32;
                word++;
        };
        nori_walk(c, tab, pred, [&](uint32_t v, int len) {
                acc |= (unsigned long long) v << (64 - fill - len); // fill < 32, len <= 16 + 11
                fill += len;
                if (fill >= 32) flush();
        });
        if (b == n_blocks - 1 || b + 1 == (sg + 1) * seg_blocks) {
                const int pad = (8 - (fill & 7)) & 7;
                if (pad) {
                        acc |= (unsigned long long) ((1u << pad) - 1u) << (64 - fill - pad);
                        fill += pad;
                        if (fill >= 32) flush();
                }
        }
        if (fill) atomicOr(raw + word, __builtin_bswap32((uint32_t) (acc >> 32)));
}

// Several segments (restart intervals too long for the block coder): v[0 .. n) -> the sums in front of every entry INSIDE its segment of S entries; seg_bits[k] = the
// sum of segment k.  (one workgroup, 8 entries per lane; the scan of sync_dc_kernel in jpeg_decode.hip)
__global__ __launch_bounds__(1024) void nori_segscan_kernel(uint32_t *__restrict__ v, uint32_t n, uint32_t S, uint32_t *__restrict__ seg_bits)
{
        constexpr int kPer = 8;
        __shared__ uint32_t wave_v[16], wave_f[16];
        __shared__ uint32_t carry_s;
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 1024 * kPer) {
                uint32_t x[kPer], incl[kPer], fl[kPer], f = 0, run = 0;
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        const uint32_t i = i0 + (uint32_t) tid * kPer + e;
                        x[e] = i < n ? v[i] : 0;
                        const uint32_t start = i < n && i % S == 0;
                        run = start ? x[e] : run + x[e];
                        f |= start;
                        incl[e] = run;
                        fl[e] = f;
                }
                uint32_t sv = run, sf = f;
                for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t of = __shfl_up(sf, d, 64), ov = __shfl_up(sv, d, 64);
                        if (lane >= d) {
                                if (!sf) sv += ov;
                                sf |= of;
                        }
                }
                if (lane == 63) { wave_v[wv] = sv; wave_f[wv] = sf; }
                uint32_t pf = __shfl_up(sf, 1, 64), pv = __shfl_up(sv, 1, 64);
                if (lane == 0) { pf = 0; pv = 0; }
                __syncthreads();
                uint32_t cv = carry_s;
                for (int w = 0; w < wv; w++) cv = wave_f[w] ? wave_v[w] : cv + wave_v[w];
                const uint32_t before = pf ? pv : cv + pv;
#pragma unroll
                for (int e = 0; e < kPer; e++) {
                        const uint32_t i = i0 + (uint32_t) tid * kPer + e;
                        if (i < n) {
                                const uint32_t in = fl[e] ? incl[e] : before + incl[e]; // inclusive, inside the segment
                                v[i] = in - x[e];
                                if (i % S == S - 1 || i == n - 1) seg_bits[i / S] = in;
                        }
                }
                __syncthreads();
                if (tid == 1023) carry_s = fl[kPer - 1] ? incl[kPer - 1] : before + incl[kPer - 1];
                __syncthreads();
        }
}

// what compact_kernel wants to know of every segment: its bytes, its bytes in the final stream (a 0x00 per 0xFF, the marker behind it), the chunk totals; one wave per segment
__global__ __launch_bounds__(256) void nori_segstat_kernel(const uint32_t *__restrict__ raw, uint32_t seg_words, const uint32_t *__restrict__ seg_bits, int n_seg,
                                                           uint32_t *__restrict__ seg_len, uint32_t *__restrict__ seg_ff, uint32_t *__restrict__ chunk_tot)
{
        const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (seg >= n_seg) return;
        const uint32_t bytes = (seg_bits[seg] + 7) / 8;
        const uint8_t *const p = (const uint8_t *) (raw + (size_t) seg * seg_words);
        int ff = 0;
        for (uint32_t i = lane; i < bytes; i += 64) ff += p[i] == 0xFF;
        ff = __builtin_amdgcn_readlane(wave_inclusive_scan(ff, lane), 63);
        if (lane == 0) {
                seg_len[seg] = bytes;
                seg_ff[seg] = bytes + (uint32_t) ff + 2;
                atomicAdd(&chunk_tot[(seg / kChunk) * kChunkStride], bytes + (uint32_t) ff + 2);
        }
}

// bytes 0xFF per 64-byte piece of the scan's bytes (total_bits = pos[n_blocks])
__global__ __launch_bounds__(256) void nori_ff_kernel(const uint32_t *__restrict__ raw, const uint32_t *__restrict__ total_bits, uint32_t n_pieces, uint32_t *__restrict__ ff)
{
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_pieces) return;
        const uint32_t bytes = (*total_bits + 7) / 8;
        uint32_t n = 0;
        for (uint32_t k = 0; k < 64 && 64 * i + k < bytes; k++) n += ((const uint8_t *) raw)[64 * i + k] == 0xFF;
        ff[i] = n;
}

// header, the scan's bytes with a 0x00 behind every 0xFF (B.1.1.5), EOI; the length to mapped host memory.  `base`: see CodeArgs::base (compact_kernel does the same)
__global__ __launch_bounds__(256) void nori_place_kernel(const uint32_t *__restrict__ raw, const uint32_t *__restrict__ total_bits, uint32_t n_pieces, const uint32_t *__restrict__ ff_before,
                                                         uint8_t *__restrict__ out, const uint8_t *__restrict__ header, int header_len, size_t capacity, uint32_t *__restrict__ total_pinned,
                                                         const uint32_t *__restrict__ base, int frame)
{
        const uint32_t base0 = base != nullptr ? base[frame] - 2u : 0u;
        out += base0;
        capacity = capacity > base0 ? capacity - base0 : 0;
        const uint32_t bytes = (*total_bits + 7) / 8;
        const uint32_t total = (uint32_t) header_len + bytes + ff_before[n_pieces] + 2u;
        if (blockIdx.x == 0 && (size_t) header_len <= capacity) {
                for (int i = threadIdx.x; i < header_len; i += 256) out[i] = header[i];
        }
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i == 0) total_pinned[frame] = base0 + total;
        if (i >= n_pieces || (size_t) total > capacity) return; // would not fit: nothing of the scan is written, the host reports the needed size from the total
        uint8_t *d = out + header_len + 64 * (size_t) i + ff_before[i];
        const uint8_t *src = (const uint8_t *) raw + 64 * (size_t) i;
        for (uint32_t k = 0; k < 64 && 64 * i + k < bytes; k++) {
                const uint8_t v = src[k];
                *d++ = v;
                if (v == 0xFF) *d++ = 0;
        }
        if (i == 0) { // (any lane knows where the scan ends)
                uint8_t *const e = out + header_len + bytes + ff_before[n_pieces];
                e[0] = 0xFF;
                e[1] = 0xD9;
        }
}

struct Encoder {
        int width, height, quality, ri, sub, hs, vs, ybl, mcu_w, mcu_h, n_mcu, n_seg, cap, device;
        int batch_cap; // frames the workspace below is sized for (1 after create; encode_batch grows it)
        int raw_cap;   // frames the buffers of the wave-per-segment path (scratch, seg_len, seg_ff, chunk totals) are sized for (0 until that path runs)
        bool force_wave_kernel; // UG_JPEG_WAVE_KERNEL=1: A/B switch back to the wave-per-segment coder + compaction (it remains the path for long restart intervals)
        bool allow_fused;       // UG_JPEG_FUSED=0: A/B switch, the front end always writes the coefficients to HBM
        std::vector<uint8_t> header;
        // device workspace
        float *div;
        int16_t *cy, *cb, *cr;
        uint32_t *scratch; // per-segment scan data before byte stuffing, cap bytes each
        uint32_t *seg_len, *seg_ff;
        uint32_t *chunk_tot;    // sums of seg_ff per kChunk segments (cleared before every call's coder)
        unsigned long long *status; // look-back words of the placing coder, n_mcu per frame (never cleared: they carry the call's generation)
        uint32_t gen;
        uint32_t *ticket;       // start-order counter of the placing coder's workgroups (self-resetting)
        uint8_t *slots;         // two-launch placement: a slot per workgroup of the coder (grown on demand), and the workgroups' byte counts
        size_t slots_cap;
        uint32_t *wg_bytes;
        bool two_launch;        // UG_JPEG_LOOKBACK=1 switches to the one-launch placement (decoupled look-back) for A/B
        bool force_two_launch;  // UG_JPEG_LOOKBACK=0: the two-launch placement for one-frame calls too (tests: every path with every input)
        unsigned long long *prof; // UG_JPEG_PROF=1: phase clock sums of the placing coder (device memory, kProfPhases + 1 words)
        bool use_ticket;        // workgroup index = start-order ticket instead of blockIdx (UG_JPEG_TICKET=1, or for good after a wait was given up)
        bool flat_lookback;     // one-frame calls: the flat form of the look-back (the default; UG_JPEG_FLAT=0 switches back to the windowed walk for A/B)
        uint8_t *header_dev;
        uint32_t *nori_bits, *nori_ff; // restart_interval 0, the parallel way: bit counts -> positions per block (+ the total); 0xFF bytes per 64-byte piece (+ the total)
        size_t nori_ff_cap;
        uint32_t *total_host; // pinned, mapped: kTotalWords words for the stream(s) of a call, then as many per scan of a non-interleaved stream
        uint32_t *total_host_dev; // the same word as the device sees it
        // ug_hip_jpeg_encoder_create_ex (gpujpeg.cpp:303-305,396-405)
        bool ycc;      // 4:4:4 coded as Y'CbCr (components 1, 2, 3, chroma tables for Cb / Cr) instead of R, G, B
        bool nonint;   // one scan per component (4:4:4 only)
        int ctab;      // Huffman table pair of the chroma blocks: 0 = the luma pair (R, G, B streams), 1
        int cs_rgb;    // RGB input is converted to this colour space in front of the FDCT (0: coded as it comes)
        int cs_uyvy;   // UYVY input (BT.709 limited range) is converted to this one (0: coded as it comes)
        bool in_uyvy;  // UG_JPEG_INPUT_UYVY: a 4:4:4 encoder fed UYVY (brought to 3 B/px, in colour space cs_444, in front of everything else)
        int cs_444;
        uint8_t *cs_tmp;       // the converted frame(s)
        size_t cs_tmp_bytes;
        std::vector<uint8_t> scan_header[3]; // non-interleaved: what precedes the entropy-coded bytes of scan c (scan 0: the whole header)
        uint8_t *scan_header_dev[3];
};
constexpr int kTotalWords = 32; // per block of total_host: kMaxBatch lengths, [kMaxBatch] "a wait was given up", [kMaxBatch + 1] "a slot overflowed"

void put16(std::vector<uint8_t> &v, int x) { v.push_back((uint8_t) (x >> 8)); v.push_back((uint8_t) x); }

// 4:2:0 / 4:2:2: JFIF (YCbCr, BT.709 limited range samples as they come from the UYVY frame, or what internal_cs asked for).  4:4:4, !ycc: the
// components are R, G, B without colour transform (what the reference's module asks of GPUJPEG for RGB input: color_space_internal =
// GPUJPEG_RGB, gpujpeg.cpp:303-305), signalled the libjpeg way: Adobe APP14 with transform 0 and ids 'R','G','B'; every component uses
// quantiser / Huffman table 0.  4:4:4, ycc: JFIF again, components 1, 2, 3 at 1x1.  scan >= 0: the header of a NON-INTERLEAVED stream up to and
// including the SOS of its first scan (scan = 0; an R, G, B stream then carries table 0 only, the layout of tests/jpeg_bitstream.py
// write_jpeg_noninterleaved), or just the SOS segment of scan 1 / 2.
std::vector<uint8_t> build_header(int w, int h, const uint8_t *ql, const uint8_t *qc, int ri, int sub, bool ycc = false, int scan = -1)
{
        const bool rgb = sub == 444 && !ycc;
        const uint8_t id[3] = { (uint8_t) (rgb ? 'R' : 1), (uint8_t) (rgb ? 'G' : 2), (uint8_t) (rgb ? 'B' : 3) };
        const uint8_t t12 = rgb ? 0 : 1;
        std::vector<uint8_t> v;
        if (scan > 0) {
                v.insert(v.end(), { 0xFF, 0xDA, 0, 8, 1, id[scan], (uint8_t) (t12 * 0x11), 0, 63, 0 });
                return v;
        }
        v = { 0xFF, 0xD8 };
        if (rgb) {
                v.insert(v.end(), { 0xFF, 0xEE, 0, 14, 'A', 'd', 'o', 'b', 'e', 0, 100, 0, 0, 0, 0, 0 });
        } else {
                v.insert(v.end(), { 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 });
        }
        const bool table0_only = scan == 0 && rgb;
        for (int t = 0; t < (table0_only ? 1 : 2); t++) {
                v.insert(v.end(), { 0xFF, 0xDB, 0, 67, (uint8_t) t });
                for (int i = 0; i < 64; i++) v.push_back((t ? qc : ql)[kZigHost[i]]);
        }
        v.insert(v.end(), { 0xFF, 0xC0, 0, 17, 8 });
        put16(v, h); put16(v, w);
        const uint8_t s0 = sub == 420 ? 0x22 : (sub == 422 ? 0x21 : 0x11); // H x V sampling of component 0
        v.insert(v.end(), { 3, id[0], s0, 0, id[1], 0x11, t12, id[2], 0x11, t12 });
        const struct { int tc, th; const uint8_t *bits, *vals; int n; } dht[4] = {
                { 0, 0, kDcL_bits, kDcL_vals, (int) sizeof kDcL_vals }, { 1, 0, kAcL_bits, kAcL_vals, (int) sizeof kAcL_vals },
                { 0, 1, kDcC_bits, kDcC_vals, (int) sizeof kDcC_vals }, { 1, 1, kAcC_bits, kAcC_vals, (int) sizeof kAcC_vals } };
        for (int k = 0; k < (table0_only ? 2 : 4); k++) {
                const auto &d = dht[k];
                v.insert(v.end(), { 0xFF, 0xC4 });
                put16(v, 19 + d.n);
                v.push_back((uint8_t) (d.tc << 4 | d.th));
                v.insert(v.end(), d.bits, d.bits + 16);
                v.insert(v.end(), d.vals, d.vals + d.n);
        }
        if (ri) {
                v.insert(v.end(), { 0xFF, 0xDD, 0, 4 });
                put16(v, ri);
        }
        if (scan == 0) {
                v.insert(v.end(), { 0xFF, 0xDA, 0, 8, 1, id[0], 0x00, 0, 63, 0 });
        } else {
                const uint8_t h12 = (uint8_t) (t12 * 0x11);
                v.insert(v.end(), { 0xFF, 0xDA, 0, 12, 3, id[0], 0x00, id[1], h12, id[2], h12, 0, 63, 0 });
        }
        return v;
}

BatchStride strides_of(const Encoder *e)
{
        BatchStride bs = {};
        bs.coef_y = (long) e->ybl * e->n_mcu * 64;
        bs.coef_c = (long) e->n_mcu * 64;
        bs.raw_words = (long) e->n_seg * (e->cap / 4);
        bs.seg = e->n_seg + 4;
        bs.tot_words = (long) ((e->n_seg + kChunk - 1) / kChunk) * kChunkStride;
        return bs;
}

void free_raw(Encoder *e)
{
        for (void **p : { (void **) &e->scratch, (void **) &e->seg_len, (void **) &e->seg_ff, (void **) &e->chunk_tot, (void **) &e->nori_bits, (void **) &e->nori_ff }) {
                if (*p) (void) hipFree(*p);
                *p = nullptr;
        }
        e->raw_cap = 0;
        e->nori_ff_cap = 0;
}

void free_workspace(Encoder *e)
{
        for (void **p : { (void **) &e->cy, (void **) &e->cb, (void **) &e->cr, (void **) &e->status, (void **) &e->slots, (void **) &e->wg_bytes }) {
                if (*p) (void) hipFree(*p);
                *p = nullptr;
        }
        e->batch_cap = 0;
        e->slots_cap = 0;
        free_raw(e);
}

// the per-frame work buffers, for `frames` frames in flight inside one call
hipError_t alloc_workspace(Encoder *e, int frames)
{
        free_workspace(e);
        const BatchStride bs = strides_of(e);
        hipError_t err = hipSuccess;
        auto alloc = [&](void **p, size_t n) { if (err == hipSuccess) err = hipMalloc(p, n * (size_t) frames); };
        alloc((void **) &e->cy, (size_t) bs.coef_y * 2);
        alloc((void **) &e->cb, (size_t) bs.coef_c * 2);
        alloc((void **) &e->cr, (size_t) bs.coef_c * 2);
        alloc((void **) &e->status, (size_t) e->n_mcu * 8);
        alloc((void **) &e->wg_bytes, (size_t) e->n_mcu * 4);
        e->slots = nullptr; // (grown on first use)
        if (err == hipSuccess) err = hipMemset(e->status, 0, (size_t) e->n_mcu * 8 * frames);
        if (err == hipSuccess) e->batch_cap = frames;
        return err;
}

// the wave-per-segment coder writes unstuffed segments for the compaction pass: its buffers exist only once that path has run
hipError_t alloc_raw(Encoder *e, int frames)
{
        free_raw(e);
        const BatchStride bs = strides_of(e);
        hipError_t err = hipSuccess;
        auto alloc = [&](void **p, size_t n) { if (err == hipSuccess) err = hipMalloc(p, n * (size_t) frames); };
        alloc((void **) &e->scratch, (size_t) bs.raw_words * 4);
        alloc((void **) &e->seg_len, (size_t) bs.seg * 4);
        alloc((void **) &e->seg_ff, (size_t) bs.seg * 4);
        alloc((void **) &e->chunk_tot, (size_t) bs.tot_words * 4);
        if (err == hipSuccess) e->raw_cap = frames;
        return err;
}

void destroy(Encoder *e)
{
        if (!e) return;
        free_workspace(e);
        if (e->prof) { // profiling run: the phase clock of the LAST call, averaged over its workgroups, as the encoder goes away
                const size_t words = (size_t) kMaxBatch * e->n_mcu * (kProfPhases + 1);
                std::vector<unsigned long long> h(words);
                if (hipMemcpy(h.data(), e->prof, words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                        double sum[kProfPhases] = {};
                        unsigned long long n = 0;
                        for (size_t g = 0; g < words / (kProfPhases + 1); g++) {
                                if (!h[g * (kProfPhases + 1) + kProfPhases]) continue;
                                n++;
                                for (int i = 0; i < kProfPhases; i++) sum[i] += (double) h[g * (kProfPhases + 1) + i];
                        }
                        fprintf(stderr, "UG_JPEG_PROF workgroups=%llu clock_ticks_per_workgroup:", n);
                        for (int i = 0; i < kProfPhases && n; i++) fprintf(stderr, " %.0f", sum[i] / (double) n);
                        fprintf(stderr, "\n");
                }
        }
        for (void *p : { (void *) e->div, (void *) e->header_dev, (void *) e->ticket, (void *) e->prof, (void *) e->cs_tmp,
                         (void *) e->scan_header_dev[0], (void *) e->scan_header_dev[1], (void *) e->scan_header_dev[2] }) {
                if (p) (void) hipFree(p);
        }
        if (e->total_host) (void) hipHostFree(e->total_host);
        delete e;
}

} // namespace

extern "C" {

typedef struct ug_hip_jpeg_encoder ug_hip_jpeg_encoder;

int ug_hip_jpeg_encoder_create_ex(int width, int height, int quality, int restart_interval, int subsampling, int internal_cs, int flags, ug_hip_jpeg_encoder **out)
{
        if (!out || width <= 0 || height <= 0 || width > 65535 || height > 65535 || restart_interval < 0 || restart_interval > 65535) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (subsampling != 420 && subsampling != 422 && subsampling != 444) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create: subsampling must be 420, 422 or 444");
                return UG_HIP_EUNSUPP;
        }
        if (internal_cs < UG_JPEG_CS_ASIS || internal_cs > UG_JPEG_CS_YCBCR_BT709 || (flags & ~(UG_JPEG_NONINTERLEAVED | UG_JPEG_INPUT_UYVY))) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create_ex: unknown colour space or flag");
                return UG_HIP_EINVAL;
        }
        if (subsampling != 444 && ((flags & (UG_JPEG_NONINTERLEAVED | UG_JPEG_INPUT_UYVY)) || internal_cs == UG_JPEG_CS_RGB)) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create_ex: a 4:2:x stream is Y'CbCr in one interleaved scan (R, G, B components, one scan per component, "
                                       "UG_JPEG_INPUT_UYVY: 4:4:4)");
                return UG_HIP_EUNSUPP;
        }
        Encoder *e = new Encoder();
        e->nonint = (flags & UG_JPEG_NONINTERLEAVED) != 0;
        e->in_uyvy = (flags & UG_JPEG_INPUT_UYVY) != 0;
        e->cs_444 = internal_cs;
        e->ycc = subsampling == 444 && (internal_cs >= UG_JPEG_CS_YCBCR_BT601 || (e->in_uyvy && internal_cs == UG_JPEG_CS_ASIS));
        e->cs_rgb = e->ycc && !e->in_uyvy ? internal_cs : 0;
        e->cs_uyvy = subsampling != 444 && (internal_cs == UG_JPEG_CS_YCBCR_BT601 || internal_cs == UG_JPEG_CS_YCBCR_BT601_256LVLS) ? internal_cs : 0;
        e->ctab = subsampling == 444 && !e->ycc ? 0 : 1;
        e->width = width; e->height = height; e->quality = quality;
        e->force_wave_kernel = getenv("UG_JPEG_WAVE_KERNEL") != nullptr && getenv("UG_JPEG_WAVE_KERNEL")[0] == '1';
        e->allow_fused = !(getenv("UG_JPEG_FUSED") != nullptr && getenv("UG_JPEG_FUSED")[0] == '0');
        e->use_ticket = getenv("UG_JPEG_TICKET") != nullptr && getenv("UG_JPEG_TICKET")[0] == '1';
        e->flat_lookback = !(getenv("UG_JPEG_FLAT") != nullptr && getenv("UG_JPEG_FLAT")[0] == '0'); // the default since round 5 (profiles/r05_jpeg_one_frame.txt)
        e->two_launch = !(getenv("UG_JPEG_LOOKBACK") != nullptr && getenv("UG_JPEG_LOOKBACK")[0] == '1');
        e->force_two_launch = getenv("UG_JPEG_LOOKBACK") != nullptr && getenv("UG_JPEG_LOOKBACK")[0] == '0';
        e->sub = subsampling;
        e->hs = subsampling == 444 ? 1 : 2; e->vs = subsampling == 420 ? 2 : 1; e->ybl = e->hs * e->vs;
        e->mcu_w = (width + 8 * e->hs - 1) / (8 * e->hs); e->mcu_h = (height + 8 * e->vs - 1) / (8 * e->vs); e->n_mcu = e->mcu_w * e->mcu_h;
        // restart_interval 0 (gpujpeg.cpp:345: the option's value goes to GPUJPEG as it is, and 0 = no restart markers): the scan is ONE segment, no DRI in the
        // header -- coded by a single wave, block after block (entropy_wave_kernel): milliseconds per frame instead of microseconds; for streams a reader without
        // restart marker support must take
        e->ri = restart_interval ? restart_interval : e->n_mcu;
        if ((long) e->ri * (e->ybl + 2) * kRawBytesPerBlock + 8 > (1L << 30)) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create: picture too large for a scan without restart intervals");
                delete e;
                return UG_HIP_EUNSUPP;
        }
        e->n_seg = (e->n_mcu + e->ri - 1) / e->ri;
        e->cap = e->ri * (e->ybl + 2) * kRawBytesPerBlock + 8; // unstuffed scan bytes of one segment (worst case 27 bits per coefficient)
        uint8_t ql[64], qc[64];
        float div[128];
        ug_hip_jpeg_qtable(quality, 0, ql);
        ug_hip_jpeg_qtable(quality, 1, qc);
        ug_hip_jpeg_divisors(ql, div);
        ug_hip_jpeg_divisors(e->ctab == 0 ? ql : qc, div + 64); // R, G, B: every component is quantised with table 0
        e->header = build_header(width, height, ql, qc, restart_interval, e->sub, e->ycc);
        if (e->nonint) {
                for (int c = 0; c < 3; c++) e->scan_header[c] = build_header(width, height, ql, qc, restart_interval, e->sub, e->ycc, c);
                e->header = e->scan_header[0]; // (what max_size and the capacity check count)
        }
        hipError_t err = hipSuccess;
        auto alloc = [&](void **p, size_t n) { if (err == hipSuccess) err = hipMalloc(p, n); };
        alloc((void **) &e->div, sizeof div);
        if (err == hipSuccess) err = alloc_workspace(e, 1);
        alloc((void **) &e->header_dev, e->header.size());
        alloc((void **) &e->ticket, 64);
        if (err == hipSuccess) err = hipMemset(e->ticket, 0, 64);
        if (getenv("UG_JPEG_PROF") != nullptr && getenv("UG_JPEG_PROF")[0] == '1') {
                const size_t bytes = (size_t) kMaxBatch * e->n_mcu * (kProfPhases + 1) * 8; // a slot per workgroup of the largest launch
                alloc((void **) &e->prof, bytes);
                if (err == hipSuccess) err = hipMemset(e->prof, 0, bytes);
        }
        static_assert(kMaxBatch * sizeof(uint32_t) <= 64, "one length word per frame of a batch");
        static_assert(kMaxBatch + 2 <= kTotalWords, "lengths + the two flag words");
        if (err == hipSuccess) err = hipHostMalloc((void **) &e->total_host, 5 * kTotalWords * 4, hipHostMallocMapped); // the call's block, one per scan, one for odd words
        if (err == hipSuccess) memset(e->total_host, 0, 5 * kTotalWords * 4);
        for (int c = 0; c < 3 && e->nonint; c++) {
                alloc((void **) &e->scan_header_dev[c], e->scan_header[c].size());
                if (err == hipSuccess) err = hipMemcpy(e->scan_header_dev[c], e->scan_header[c].data(), e->scan_header[c].size(), hipMemcpyHostToDevice);
        }
        if (err == hipSuccess) err = hipHostGetDevicePointer((void **) &e->total_host_dev, e->total_host, 0);
        if (err == hipSuccess) err = hipMemcpy(e->div, div, sizeof div, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy(e->header_dev, e->header.data(), e->header.size(), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_jpeg_encoder_create");
                destroy(e);
                return UG_HIP_ERUNTIME;
        }
        *out = (ug_hip_jpeg_encoder *) e;
        return UG_HIP_SUCCESS;
}

int ug_hip_jpeg_encoder_create_sub(int width, int height, int quality, int restart_interval, int subsampling, ug_hip_jpeg_encoder **out)
{
        return ug_hip_jpeg_encoder_create_ex(width, height, quality, restart_interval, subsampling, UG_JPEG_CS_ASIS, 0, out);
}

int ug_hip_jpeg_encoder_create(int width, int height, int quality, int restart_interval, ug_hip_jpeg_encoder **out)
{
        return ug_hip_jpeg_encoder_create_sub(width, height, quality, restart_interval, 420, out);
}

void ug_hip_jpeg_encoder_destroy(ug_hip_jpeg_encoder *enc) { destroy((Encoder *) enc); }

size_t ug_hip_jpeg_encoder_max_size(const ug_hip_jpeg_encoder *enc)
{
        const Encoder *e = (const Encoder *) enc;
        return e ? e->header.size() + 20 + (size_t) e->n_seg * (2 * (size_t) e->cap + 2) : 0; // every byte stuffed = worst case (+ the SOS of two more scans)
}

int ug_hip_jpeg_encoder_encode(ug_hip_jpeg_encoder *enc, ug_pixfmt_t in, const void *src_dev, int src_pitch, void *out_dev,
                               size_t out_capacity, size_t *out_len, ug_hip_stream_t stream)
{
        return ug_hip_jpeg_encoder_encode_batch(enc, in, 1, src_dev, src_pitch, 0, out_dev, 0, out_capacity, out_len, stream);
}

int ug_hip_jpeg_encoder_encode_batch(ug_hip_jpeg_encoder *enc, ug_pixfmt_t in, int frames, const void *src_dev, int src_pitch, size_t src_stride,
                                     void *out_dev, size_t out_stride, size_t out_capacity, size_t *out_len, ug_hip_stream_t stream)
{
        Encoder *e = (Encoder *) enc;
        if (!e || !src_dev || !out_dev || !out_len || (15 & (uintptr_t) out_dev) || frames < 1 || frames > kMaxBatch ||
            (frames > 1 && ((out_stride & 15) || out_stride < out_capacity))) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: bad arguments (1..16 frames; out_stride a multiple of 16, >= out_capacity)");
                return UG_HIP_EINVAL;
        }
        if (out_capacity < e->header.size() + 2) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: output buffer smaller than the JPEG header");
                return UG_HIP_EINVAL;
        }
        hipStream_t st = (hipStream_t) stream;
        if (frames > e->batch_cap) { // the work buffers of a frame, once per frame of the batch (grown once; kernels of earlier calls have finished: encode is synchronous)
                const hipError_t err = alloc_workspace(e, frames);
                if (err != hipSuccess) {
                        ug::set_last_error(err, "ug_hip_jpeg_encoder_encode_batch: work buffers");
                        if (alloc_workspace(e, 1) != hipSuccess) free_workspace(e);
                        return UG_HIP_ERUNTIME;
                }
        }
        BatchStride bs = strides_of(e);
        bs.out_bytes = (long) out_stride;
        int rc = UG_HIP_SUCCESS;
        const int w = e->width, h = e->height;
        const int S_frame = e->ri * (e->ybl + 2); // blocks per (full) restart segment
        const bool wave_path = (e->nonint ? e->ri : S_frame) > 256 || e->force_wave_kernel; // (a scan of one component: a segment is ri blocks)
        if (!src_pitch && in == UG_PF_UYVY) src_pitch = ug::linesize(UG_PF_UYVY, w);
        // ---- the colour stage (create_ex's internal_cs): the frame(s) converted into a buffer of the encoder's, which then takes the input's place ----
        const int cs_to = in == UG_PF_RGB ? e->cs_rgb : (in == UG_PF_UYVY ? e->cs_uyvy : 0);
        if (in == UG_PF_I420 && e->cs_uyvy) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: planar input is coded as it comes (no colour conversion)");
                return UG_HIP_EUNSUPP;
        }
        if (cs_to) {
                if (!src_pitch) src_pitch = 3 * w;
                const int ls = in == UG_PF_RGB ? 3 * w : ug::linesize(UG_PF_UYVY, w);
                const size_t fb = ((size_t) ls * h + 15) / 16 * 16;
                if (e->cs_tmp_bytes < fb * frames) {
                        if (e->cs_tmp) (void) hipFree(e->cs_tmp);
                        e->cs_tmp = nullptr;
                        e->cs_tmp_bytes = 0;
                        UG_HIP_TRY(hipMalloc((void **) &e->cs_tmp, fb * kMaxBatch));
                        e->cs_tmp_bytes = fb * kMaxBatch;
                }
                const int crc = ug::jpeg_colour_convert(in, in == UG_PF_RGB ? UG_JPEG_CS_RGB : UG_JPEG_CS_YCBCR_BT709, cs_to, src_dev, src_pitch, e->cs_tmp, ls, w, h, frames,
                                                        src_stride, fb, stream);
                if (crc != UG_HIP_SUCCESS) return crc;
                src_dev = e->cs_tmp;
                src_pitch = ls;
                src_stride = fb;
        }
        // ---- a 4:4:4 encoder fed UYVY (UG_JPEG_INPUT_UYVY): the frame(s) as 3 B/px in the coded colour space, which then take the place of RGB input ----
        if (e->sub == 444 && e->in_uyvy != (in == UG_PF_UYVY)) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: a 4:4:4 encoder takes RGB, or UYVY when created with UG_JPEG_INPUT_UYVY");
                return UG_HIP_EUNSUPP;
        }
        if (e->sub == 444 && e->in_uyvy) {
                const size_t fb = ((size_t) 3 * w * h + 15) / 16 * 16;
                if (e->cs_tmp_bytes < fb * frames) {
                        if (e->cs_tmp) (void) hipFree(e->cs_tmp);
                        e->cs_tmp = nullptr;
                        e->cs_tmp_bytes = 0;
                        UG_HIP_TRY(hipMalloc((void **) &e->cs_tmp, fb * kMaxBatch));
                        e->cs_tmp_bytes = fb * kMaxBatch;
                }
                const int crc = ug::jpeg_uyvy_to_444(e->cs_444, src_dev, src_pitch, e->cs_tmp, 3 * w, w, h, frames, src_stride, fb, stream);
                if (crc != UG_HIP_SUCCESS) return crc;
                in = UG_PF_RGB;
                src_dev = e->cs_tmp;
                src_pitch = 3 * w;
                src_stride = fb;
        }
        // Fused: forward DCT, quantiser, Huffman coding and byte stuffing in ONE kernel, a workgroup per 32 consecutive MCUs -- the quantised
        // coefficients never reach HBM.  Needs whole segments per workgroup (32 % ri == 0) and the aligned geometry of the fast front end.
        if (!src_pitch && in == UG_PF_RGB) src_pitch = 3 * w;
        // a workgroup = 32 (RGB: 64) consecutive MCUs of the scan = whole segments: the restart interval must divide that.  fused_ok: the fused
        // kernels divide MCU numbers by the MCUs of a row with a multiplication (CodeArgs::mcu_w_m32), exact while (n_mcu + 64) * mcu_w < 2^32
        // and the row has more than one MCU -- every picture from 9 (17) pixels wide up to ~16K x 16K; others take the two-kernel path
        const bool fused_ok = e->mcu_w > 1 && ((long) e->n_mcu + 64) * e->mcu_w < (1L << 32);
        const bool fused_yuv = fused_ok && !wave_path && e->allow_fused && in == UG_PF_UYVY && e->sub != 444 && w % 16 == 0 && !(src_pitch & 15) && !(15 & (uintptr_t) src_dev) &&
                               (frames == 1 || !(src_stride & 15)) && 32 % e->ri == 0;
        // packed RGB (4:4:4, R, G, B components): any width and alignment (the edge blocks of the picture are loaded byte by byte)
        const bool fused_rgb = fused_ok && !wave_path && e->allow_fused && in == UG_PF_RGB && e->sub == 444 && 64 % e->ri == 0 && !e->ycc && !e->nonint;
        // planar I420: 8-byte row pieces of the three planes (width % 16 == 0 keeps the chroma rows 8-byte aligned too)
        const bool fused_i420 = fused_ok && !wave_path && e->allow_fused && in == UG_PF_I420 && e->sub == 420 && w % 16 == 0 && (!src_pitch || src_pitch == w) && !(7 & (uintptr_t) src_dev) &&
                                (frames == 1 || !(src_stride & 7)) && 32 % e->ri == 0;
        const bool fused = fused_yuv || fused_rgb || fused_i420;
        if (fused) {
                // nothing to do here: the coder below reads the frame itself
        } else if (in == UG_PF_UYVY && e->sub != 444) { // fused unpack + subsample + FDCT + quantise, grid.z = frame
                rc = ug_hip_uyvy_to_jpeg42x_coeffs_batch(e->sub, src_dev, src_pitch, w, h, e->div, e->cy, e->cb, e->cr, frames, src_stride,
                                                         (size_t) bs.coef_y * 2, (size_t) bs.coef_c * 2, stream);
        } else if (in == UG_PF_RGB && e->sub == 444) { // GPUJPEG_444_U8_P012, components kept as R, G, B (gpujpeg.cpp:303-305,336)
                if (!src_pitch) src_pitch = 3 * w;
                for (int f = 0; f < frames && rc == UG_HIP_SUCCESS; f++) {
                        const uint8_t *const fr = (const uint8_t *) src_dev + f * src_stride;
                        if (e->ycc) { // Y', Cb, Cr (the colour stage's output): the luma quantiser for component 0, the chroma one for the others
                                rc = ug::jpeg_fdct_quant_strided(fr, src_pitch, 3, w, h, e->mcu_w, e->mcu_h, e->div, e->cy + f * bs.coef_y, nullptr, stream);
                                if (rc == UG_HIP_SUCCESS) rc = ug::jpeg_fdct_quant_strided(fr + 1, src_pitch, 3, w, h, e->mcu_w, e->mcu_h, e->div + 64, e->cb + f * bs.coef_c, nullptr, stream);
                                if (rc == UG_HIP_SUCCESS) rc = ug::jpeg_fdct_quant_strided(fr + 2, src_pitch, 3, w, h, e->mcu_w, e->mcu_h, e->div + 64, e->cr + f * bs.coef_c, nullptr, stream);
                        } else {
                                rc = ug::jpeg_fdct_quant_rgb444(fr, src_pitch, w, h, e->mcu_w, e->mcu_h, e->div, e->cy + f * bs.coef_y, e->cb + f * bs.coef_c, e->cr + f * bs.coef_c, stream);
                        }
                }
        } else if (in == UG_PF_I420 && e->sub == 420) { // planar passthrough (GPUJPEG_420_U8_P0P1P2, gpujpeg.cpp:335): Y, U, V planes back to back
                if (src_pitch && src_pitch != w) {
                        ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: I420 input must be tightly packed");
                        return UG_HIP_EINVAL;
                }
                const int cw = (w + 1) / 2, ch = (h + 1) / 2;
                for (int f = 0; f < frames && rc == UG_HIP_SUCCESS; f++) {
                        const uint8_t *y = (const uint8_t *) src_dev + f * src_stride, *u = y + (size_t) w * h, *v = u + (size_t) cw * ch;
                        rc = ug::jpeg_fdct_quant_strided(y, w, 1, w, h, 2 * e->mcu_w, 2 * e->mcu_h, e->div, e->cy + f * bs.coef_y, nullptr, stream);
                        if (rc == UG_HIP_SUCCESS) rc = ug::jpeg_fdct_quant_strided(u, cw, 1, cw, ch, e->mcu_w, e->mcu_h, e->div + 64, e->cb + f * bs.coef_c, nullptr, stream);
                        if (rc == UG_HIP_SUCCESS) rc = ug::jpeg_fdct_quant_strided(v, cw, 1, cw, ch, e->mcu_w, e->mcu_h, e->div + 64, e->cr + f * bs.coef_c, nullptr, stream);
                }
        } else {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: input must be UYVY (4:2:0 / 4:2:2 encoder), I420 (4:2:0) or RGB (4:4:4); "
                                       "convert other formats with ug_hip_pixfmt_convert");
                return UG_HIP_EUNSUPP;
        }
        if (rc != UG_HIP_SUCCESS) return rc;
        e->total_host[kMaxBatch] = 0;
        // the block-parallel coder, fused or behind the front end; two_launch: slots + gather launch, else the one-launch placement (look-back)
        // `scan` = nullptr: the frame's one interleaved scan.  Else one scan of a non-interleaved stream: a single component's blocks (its MCU is one block),
        // its own header bytes, destination and length words; always the one-launch placement
        struct ScanPlan { const int16_t *coef; int tab0; const uint8_t *header; int header_len; const uint32_t *base; uint32_t *total; };
        auto launch_coder = [&](bool two_launch, const ScanPlan *scan = nullptr) -> int {
                const int S = scan ? e->ri : S_frame;
                const bool fused = !scan && (fused_yuv || fused_rgb || fused_i420); // (shadows the call's: a scan of one component reads coefficients)
                if (++e->gen >= (1u << 30)) { // the status words carry the call's generation in 30 bits: start over on clean words
                        UG_HIP_TRY(hipMemsetAsync(e->status, 0, (size_t) e->n_mcu * 8 * e->batch_cap, st));
                        e->gen = 1;
                }
                CodeArgs a = {};
                a.mcu_w = e->mcu_w; a.n_mcu = e->n_mcu; a.hs = e->hs; a.vs = e->vs; a.ctab = e->ctab; a.ri = e->ri; a.n_seg = e->n_seg; a.S = S;
                a.nc = scan ? 0 : 2; a.tab0 = scan ? scan->tab0 : 0;
                a.cy = scan ? scan->coef : e->cy; a.cb = e->cb; a.cr = e->cr; a.coef_y = bs.coef_y; a.coef_c = bs.coef_c;
                a.src = (const uint8_t *) src_dev; a.pitch = src_pitch; a.width = w; a.height = h; a.src_stride = src_stride;
                a.out = (uint8_t *) out_dev; a.out_stride = out_stride; a.capacity = out_capacity; a.header = e->header_dev; a.header_len = (int) e->header.size();
                a.total_pinned = e->total_host_dev;
                if (scan) {
                        a.header = scan->header; a.header_len = scan->header_len; a.base = scan->base;
                        a.total_pinned = scan->total;
                }
                a.status = e->status; a.n_status = e->n_mcu; a.gen = e->gen; a.ticket = e->use_ticket ? e->ticket : nullptr; a.prof = e->prof;
                a.flat = 0; // decided below, once the number of workgroups of a frame is known
                // divisions by S, blocks per MCU and MCUs per row as multiplications, where the ranges allow (CodeArgs)
                {
                        const int per_mcu = e->hs * e->vs + a.nc;
                        auto m16 = [](int d) { return (uint32_t) (65536 / d + 1); };
                        auto m32 = [](long d) { return d > 1 ? (uint32_t) ((1ull << 32) / (unsigned long long) d + 1ull) : 0u; };
                        a.S_m16 = m16(S);             // x < 256 lanes, S <= 256: x * S < 2^16
                        a.per_mcu_m16 = m16(per_mcu); // x < S <= 256, per_mcu <= 6
                        a.S_m32 = m32(S);             // x <= S * 16 window words + S
                        a.mcu_w_m32 = m32(e->mcu_w);  // x < n_mcu + 64 (the lanes of a last, short workgroup): see fused_ok
                }
                int waves, kwin = 16;
                if (fused) {
                        const int mcus = fused_rgb ? 64 : 32; // MCUs per workgroup
                        a.n_wg = (e->n_mcu + mcus - 1) / mcus;
                        a.G = mcus / e->ri;
                        waves = e->sub == 422 ? 2 : 3;
                        if (e->sub != 422) kwin = 12;
                } else {
                        // waves per workgroup: the count that leaves the fewest lanes idle (fewer waves on a tie)
                        waves = 1;
                        for (int k = 1; k <= 4; k++) {
                                if (64 * k >= S && (64 * k / S) * S * (64 * waves) > (64 * waves / S) * S * (64 * k)) waves = k;
                                if (64 * waves < S) waves = k;
                        }
                        // (the kernel's per-segment LDS words are sized for segments of at least 3 blocks, kMaxSeg = lanes / 3 + 1: the shortest ones an
                        // interleaved scan has; a one-component scan with restart interval 1 or 2 has shorter ones and leaves lanes idle instead)
                        a.G = std::min(64 * waves / S, 64 * waves / 3);
                        a.n_wg = (e->n_seg + a.G - 1) / a.G;
                }
                if (two_launch) {
                        // a slot holds whatever the one-pass path can produce: every window full, every byte stuffed, the markers (+ the padding the
                        // gather's word reads may touch); a workgroup on the general path that needs more reports it and the call runs again, one launch
                        a.slot_bytes = ((size_t) 64 * waves * kwin * 4 * 2 + 2 * (size_t) a.G + 16 + 255) / 256 * 256;
                        const size_t need = a.slot_bytes * (size_t) a.n_wg * (size_t) frames;
                        if (need > e->slots_cap) {
                                if (e->slots) (void) hipFree(e->slots);
                                e->slots = nullptr;
                                e->slots_cap = 0;
                                const size_t want = a.slot_bytes * (size_t) a.n_wg * (size_t) e->batch_cap;
                                // + 8: the gather reads the stream in 8-byte words and may touch up to 7 bytes past byte n of a slot; a general-path workgroup
                                // may fill its slot to the last byte, and the last slot of the last frame ends where the allocation would
                                const hipError_t err = hipMalloc((void **) &e->slots, want + 8);
                                if (err != hipSuccess) {
                                        (void) hipGetLastError();
                                        two_launch = false; // no room for the slots: the one-launch placement needs none
                                } else {
                                        e->slots_cap = want;
                                }
                        }
                }
                if (two_launch) {
                        a.slots = e->slots;
                        a.wg_bytes = e->wg_bytes;
                }
                // The flat look-back reads n_wg^2 / 2 eight-byte words per frame: 4 MB at 4K (n_wg ~ 1000) -- all L2 hits, 10 % off the one-frame call --,
                // 64 MB at 8K (-6 %), and it grows with the square: past kFlatMaxWg workgroups a frame the windowed walk (16 round trips per 1024
                // workgroups) is the cheaper form again (nothing bounded it for the sizes create() accepts, up to 65535 x 65535)
                constexpr int kFlatMaxWg = 4096;
                a.flat = frames == 1 && e->flat_lookback && a.n_wg <= kFlatMaxWg ? 1 : 0;
                const dim3 grid((unsigned) a.n_wg, (unsigned) frames);
                if (fused) {
                        if (fused_i420) hipLaunchKernelGGL((jpeg_code_kernel<3, 1420>), grid, dim3(192), 0, st, a, (const float *) e->div);
                        else if (e->sub == 420) hipLaunchKernelGGL((jpeg_code_kernel<3, 420>), grid, dim3(192), 0, st, a, (const float *) e->div);
                        else if (e->sub == 422 && frames >= 2) hipLaunchKernelGGL((jpeg_code_kernel<2, 422, true>), grid, dim3(128), 0, st, a, (const float *) e->div);
                        else if (e->sub == 422) hipLaunchKernelGGL((jpeg_code_kernel<2, 422>), grid, dim3(128), 0, st, a, (const float *) e->div);
                        else hipLaunchKernelGGL((jpeg_code_kernel<3, 444>), grid, dim3(192), 0, st, a, (const float *) e->div);
                } else {
                        switch (waves) {
                        case 1: hipLaunchKernelGGL((jpeg_code_kernel<1, 0>), grid, dim3(64), 0, st, a, (const float *) e->div); break;
                        case 2: hipLaunchKernelGGL((jpeg_code_kernel<2, 0>), grid, dim3(128), 0, st, a, (const float *) e->div); break;
                        case 3: hipLaunchKernelGGL((jpeg_code_kernel<3, 0>), grid, dim3(192), 0, st, a, (const float *) e->div); break;
                        default: hipLaunchKernelGGL((jpeg_code_kernel<4, 0>), grid, dim3(256), 0, st, a, (const float *) e->div); break;
                        }
                }
                if (two_launch) {
                        hipLaunchKernelGGL(jpeg_gather_kernel, dim3((unsigned) ((a.n_wg + 3) / 4), (unsigned) frames), dim3(256), 0, st, (const uint8_t *) e->slots, a.slot_bytes,
                                           (const uint32_t *) e->wg_bytes, a.n_wg, (uint8_t *) out_dev, out_stride, out_capacity, (const uint8_t *) e->header_dev,
                                           (int) e->header.size(), e->total_host_dev);
                }
                return UG_HIP_SUCCESS;
        };
        // A scan that is ONE segment (restart_interval 0), coded in parallel (nori_*_kernel above); frame after frame on the stream, one more host synchronisation per frame
        // (the scan's length in bits, to size the byte-stuffing launches).  scan = nullptr: the interleaved scan.
        static const bool nori_off = getenv("UG_JPEG_NORI") != nullptr && getenv("UG_JPEG_NORI")[0] == '0';
        const bool nori = wave_path && !e->force_wave_kernel && e->n_seg == 1 && !nori_off && (unsigned long long) e->cap * 8ull < (1ull << 32); // (UG_JPEG_WAVE_KERNEL=1: the old kernels throughout)
        auto code_without_restart = [&](const ScanPlan *scan) -> int {
                const uint32_t n_blocks = (uint32_t) e->n_mcu * (uint32_t) (scan ? 1 : e->ybl + 2);
                if (!e->nori_bits) UG_HIP_TRY(hipMalloc((void **) &e->nori_bits, ((size_t) e->n_mcu * (e->ybl + 2) + 1) * 4));
                for (int f = 0; f < frames; f++) {
                        NoriScan s = {};
                        s.c0 = (scan ? scan->coef : e->cy) + f * bs.coef_y; s.c1 = e->cb + f * bs.coef_c; s.c2 = e->cr + f * bs.coef_c;
                        s.mcu_w = e->mcu_w; s.n_mcu = e->n_mcu; s.hs = scan ? 1 : e->hs; s.vs = scan ? 1 : e->vs; s.nc = scan ? 0 : 2; s.tab0 = scan ? scan->tab0 : 0; s.ctab = e->ctab; s.ri = e->n_mcu;
                        hipLaunchKernelGGL(nori_len_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, s, n_blocks, e->nori_bits);
                        hipLaunchKernelGGL(nori_scan_kernel, dim3(1), dim3(1024), 0, st, e->nori_bits, n_blocks);
                        uint32_t *const bits_host = e->total_host + 4 * kTotalWords; // (a word of the pinned block of its own)
                        UG_HIP_TRY(hipMemcpyAsync(bits_host, e->nori_bits + n_blocks, 4, hipMemcpyDeviceToHost, st));
                        UG_HIP_TRY(hipStreamSynchronize(st));
                        const size_t bytes = ((size_t) *bits_host + 7) / 8;
                        const uint32_t n_pieces = (uint32_t) ((bytes + 63) / 64);
                        if (e->nori_ff_cap < (size_t) n_pieces + 1) {
                                if (e->nori_ff) (void) hipFree(e->nori_ff);
                                e->nori_ff = nullptr;
                                e->nori_ff_cap = 0;
                                UG_HIP_TRY(hipMalloc((void **) &e->nori_ff, ((size_t) n_pieces + 1) * 4 * 2));
                                e->nori_ff_cap = ((size_t) n_pieces + 1) * 2;
                        }
                        UG_HIP_TRY(hipMemsetAsync(e->scratch, 0, (bytes + 8 + 3) & ~(size_t) 3, st));
                        hipLaunchKernelGGL(nori_emit_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, s, n_blocks, (const uint32_t *) e->nori_bits, e->scratch, n_blocks, 0u);
                        if (n_pieces) hipLaunchKernelGGL(nori_ff_kernel, dim3((n_pieces + 255) / 256), dim3(256), 0, st, (const uint32_t *) e->scratch, (const uint32_t *) e->nori_bits + n_blocks, n_pieces, e->nori_ff);
                        hipLaunchKernelGGL(nori_scan_kernel, dim3(1), dim3(1024), 0, st, e->nori_ff, n_pieces);
                        hipLaunchKernelGGL(nori_place_kernel, dim3(n_pieces ? (n_pieces + 255) / 256 : 1), dim3(256), 0, st, (const uint32_t *) e->scratch, (const uint32_t *) e->nori_bits + n_blocks,
                                           n_pieces, (const uint32_t *) e->nori_ff, (uint8_t *) out_dev + (size_t) f * out_stride, scan ? scan->header : (const uint8_t *) e->header_dev,
                                           scan ? scan->header_len : (int) e->header.size(), out_capacity, scan ? scan->total : e->total_host_dev, scan ? scan->base : nullptr, f);
                }
                return UG_HIP_SUCCESS;
        };
        // Restart intervals too long for the block coder (more than 256 blocks per segment), SEVERAL segments: the same lane-per-block coding, into the per-segment buffers
        // of the wave-per-segment coder (bit positions inside a segment by a segmented prefix sum), in front of that coder's compaction kernel -- which moves a segment
        // per wave, where entropy_wave_kernel CODED a segment per wave, block after block (restart=2000 at 4K: 4 ms).  UG_JPEG_NORI=0: the old kernel.
        // (which is faster, profiles/r06_encode_no_restart.txt: the old kernel takes ~0.9 us per block of a segment, the segments side by side; this way costs what the picture
        // costs -- clearing the buffers, two passes over the blocks: ~0.58 ms at 4K -- whatever the interval.  4K: from ~650 blocks per segment, 1080p: from ~290)
        const bool seg_parallel = wave_path && !e->force_wave_kernel && e->n_seg > 1 && !nori_off &&
                                  0.9 * (double) e->ri * (e->nonint ? 1 : e->ybl + 2) > 150.0 + 5.2e-5 * (double) w * (double) h;
        auto fill_segments = [&](const ScanPlan *scan) -> int {
                const uint32_t per_mcu = (uint32_t) (scan ? 1 : e->ybl + 2), n_blocks = (uint32_t) e->n_mcu * per_mcu;
                if (!e->nori_bits) UG_HIP_TRY(hipMalloc((void **) &e->nori_bits, ((size_t) e->n_mcu * (e->ybl + 2) + 1) * 4));
                if (e->nori_ff_cap < (size_t) e->n_seg + 1) {
                        if (e->nori_ff) (void) hipFree(e->nori_ff);
                        e->nori_ff = nullptr;
                        e->nori_ff_cap = 0;
                        UG_HIP_TRY(hipMalloc((void **) &e->nori_ff, ((size_t) e->n_seg + 1) * 4));
                        e->nori_ff_cap = (size_t) e->n_seg + 1;
                }
                for (int f = 0; f < frames; f++) {
                        NoriScan s = {};
                        s.c0 = (scan ? scan->coef : e->cy) + f * bs.coef_y; s.c1 = e->cb + f * bs.coef_c; s.c2 = e->cr + f * bs.coef_c;
                        s.mcu_w = e->mcu_w; s.n_mcu = e->n_mcu; s.hs = scan ? 1 : e->hs; s.vs = scan ? 1 : e->vs; s.nc = scan ? 0 : 2; s.tab0 = scan ? scan->tab0 : 0; s.ctab = e->ctab; s.ri = e->ri;
                        uint32_t *const raw = e->scratch + (size_t) f * bs.raw_words;
                        hipLaunchKernelGGL(nori_len_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, s, n_blocks, e->nori_bits);
                        hipLaunchKernelGGL(nori_segscan_kernel, dim3(1), dim3(1024), 0, st, e->nori_bits, n_blocks, (uint32_t) e->ri * per_mcu, e->nori_ff);
                        UG_HIP_TRY(hipMemsetAsync(raw, 0, (size_t) e->n_seg * (size_t) e->cap, st));
                        hipLaunchKernelGGL(nori_emit_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, s, n_blocks, (const uint32_t *) e->nori_bits, raw, (uint32_t) e->ri * per_mcu,
                                           (uint32_t) (e->cap / 4));
                        hipLaunchKernelGGL(nori_segstat_kernel, dim3((e->n_seg + 3) / 4), dim3(256), 0, st, (const uint32_t *) raw, (uint32_t) (e->cap / 4), (const uint32_t *) e->nori_ff, e->n_seg,
                                           e->seg_len + (size_t) f * bs.seg, e->seg_ff + (size_t) f * bs.seg, e->chunk_tot + (size_t) f * bs.tot_words);
                }
                return UG_HIP_SUCCESS;
        };
        if (e->nonint) {
                // One scan per component (T.81 A.2.2; the reference's default for RGB input, gpujpeg.cpp:303).  Each scan is the block coder over ONE
                // component's coefficients -- DC prediction and restart intervals (in blocks) of its own, its markers numbered from RST0 -- preceded by
                // its header bytes (scan 0: SOI .. SOS; scans 1, 2: their SOS).  Where a scan ends is only known once it is coded: the launch of scan c
                // reads the length scan c - 1 left behind (CodeArgs::base) and goes on from there, over the EOI every coded stream ends with.  Three
                // launches on the stream, one synchronisation, no intermediate buffer.
                memset(e->total_host, 0, 4 * kTotalWords * 4);
                if (wave_path && frames > e->raw_cap) {
                        const hipError_t err = alloc_raw(e, e->batch_cap);
                        if (err != hipSuccess) {
                                ug::set_last_error(err, "ug_hip_jpeg_encoder_encode: work buffers of the wave-per-segment coder");
                                free_raw(e);
                                return UG_HIP_ERUNTIME;
                        }
                }
                for (int c = 0; c < 3; c++) {
                        const ScanPlan pl = { c == 0 ? e->cy : (c == 1 ? e->cb : e->cr), e->ycc && c > 0 ? 1 : 0, e->scan_header_dev[c], (int) e->scan_header[c].size(),
                                              c > 0 ? e->total_host_dev + c * kTotalWords : nullptr, e->total_host_dev + (c + 1) * kTotalWords };
                        if (nori) {
                                const int nrc = code_without_restart(&pl);
                                if (nrc != UG_HIP_SUCCESS) return nrc;
                                continue;
                        }
                        if (wave_path) { // restart intervals of more than 256 blocks: a wave per segment, then the compaction -- scan after scan on the stream
                                UG_HIP_TRY(hipMemsetAsync(e->chunk_tot, 0, (size_t) bs.tot_words * 4 * frames, st));
                                if (seg_parallel) {
                                        const int frc = fill_segments(&pl);
                                        if (frc != UG_HIP_SUCCESS) return frc;
                                } else {
                                        hipLaunchKernelGGL(entropy_wave_kernel, dim3((e->n_seg + 3) / 4, frames), dim3(256), 0, st, pl.coef, pl.coef, pl.coef, e->mcu_w, e->n_mcu, 1, 1,
                                                           0, 0, pl.tab0, e->ri, e->n_seg, e->scratch, e->cap / 4, e->seg_len, e->seg_ff, e->chunk_tot, bs);
                                }
                                hipLaunchKernelGGL(compact_kernel, dim3((e->n_seg + 3) / 4, frames), dim3(256), 0, st, (const uint8_t *) e->scratch, e->cap, e->seg_len, e->seg_ff,
                                                   e->chunk_tot, e->n_seg, (uint8_t *) out_dev, pl.header, pl.header_len, out_capacity, pl.total, pl.base, bs);
                                continue;
                        }
                        const int lrc = launch_coder(false, &pl);
                        if (lrc != UG_HIP_SUCCESS) return lrc;
                }
                UG_HIP_LAUNCH_CHECK();
                UG_HIP_TRY(hipStreamSynchronize(st));
                const uint32_t *const t1 = e->total_host + kTotalWords, *const t2 = t1 + kTotalWords, *const t3 = t2 + kTotalWords;
                if (t1[kMaxBatch] | t2[kMaxBatch] | t3[kMaxBatch]) { // a workgroup gave up waiting for an earlier one (see below): start-order tickets, once more
                        (void) hipMemset(e->ticket, 0, 4);
                        if (!e->use_ticket) {
                                e->use_ticket = true;
                                fprintf(stderr, "[ug_mi355x] JPEG encoder %dx%d: workgroups did not start in index order; this call is encoded again and the encoder uses "
                                                "start-order tickets from now on (UG_JPEG_TICKET=1 selects that from the start)\n", e->width, e->height);
                                return ug_hip_jpeg_encoder_encode_batch(enc, in, frames, src_dev, src_pitch, src_stride, out_dev, out_stride, out_capacity, out_len, stream);
                        }
                        ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: the stream placement gave up waiting for an earlier workgroup");
                        return UG_HIP_ERUNTIME;
                }
                bool fits = true;
                for (int f = 0; f < frames; f++) {
                        out_len[f] = t3[f]; // the last scan's end = the stream's length (what it needs, when it does not fit: nothing past the capacity was written)
                        fits = fits && out_len[f] <= out_capacity;
                }
                if (!fits && frames == 1) {
                        ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: stream does not fit the output buffer (out_len = needed size)");
                        return UG_HIP_EINVAL;
                }
                return UG_HIP_SUCCESS;
        }
        e->total_host[kMaxBatch + 1] = 0;
        if (!wave_path) {
                // one frame per call: the look-back's waits are short (every workgroup of the frame is resident at once) and the second launch
                // costs more than it saves (measured 39.4 against 41.3 us per 4K frame); from two frames up the two-launch placement wins
                const int lrc = launch_coder(e->two_launch && (frames > 1 || e->force_two_launch));
                if (lrc != UG_HIP_SUCCESS) return lrc;
        } else {
                if (frames > e->raw_cap) {
                        const hipError_t err = alloc_raw(e, e->batch_cap);
                        if (err != hipSuccess) {
                                ug::set_last_error(err, "ug_hip_jpeg_encoder_encode: work buffers of the wave-per-segment coder");
                                free_raw(e);
                                return UG_HIP_ERUNTIME;
                        }
                }
                if (nori) {
                        const int nrc = code_without_restart(nullptr);
                        if (nrc != UG_HIP_SUCCESS) return nrc;
                } else {
                // the totals this call's coder adds into start from zero (a smaller batch in between must not leave stale slices behind)
                UG_HIP_TRY(hipMemsetAsync(e->chunk_tot, 0, (size_t) bs.tot_words * 4 * frames, st));
                if (seg_parallel) {
                        const int frc = fill_segments(nullptr);
                        if (frc != UG_HIP_SUCCESS) return frc;
                } else {
                hipLaunchKernelGGL(entropy_wave_kernel, dim3((e->n_seg + 3) / 4, frames), dim3(256), 0, st, e->cy, e->cb, e->cr, e->mcu_w, e->n_mcu, e->hs, e->vs,
                                   e->ctab, 2, 0, e->ri, e->n_seg, e->scratch, e->cap / 4, e->seg_len, e->seg_ff, e->chunk_tot, bs);
                }
                hipLaunchKernelGGL(compact_kernel, dim3((e->n_seg + 3) / 4, frames), dim3(256), 0, st, (const uint8_t *) e->scratch, e->cap, e->seg_len, e->seg_ff, e->chunk_tot,
                                   e->n_seg, (uint8_t *) out_dev, e->header_dev, (int) e->header.size(), out_capacity, e->total_host_dev, nullptr, bs);
                }
        }
        UG_HIP_LAUNCH_CHECK();
        UG_HIP_TRY(hipStreamSynchronize(st)); // ONE synchronisation for the batch
        if (!wave_path && e->total_host[kMaxBatch + 1] != 0) { // a workgroup's bytes did not fit its slot (near-lossless quality on noise): once more, in one launch
                e->total_host[kMaxBatch + 1] = 0;
                const int lrc = launch_coder(false);
                if (lrc != UG_HIP_SUCCESS) return lrc;
                UG_HIP_LAUNCH_CHECK();
                UG_HIP_TRY(hipStreamSynchronize(st));
        }
        if (e->total_host[kMaxBatch] != 0) { // a workgroup gave up waiting for an earlier one: reported, never waited out
                (void) hipMemset(e->ticket, 0, 4);
                if (!e->use_ticket) { // the start order was not the index order after all: from now on the index IS the start order; once more
                        e->use_ticket = true;
                        // said once, loudly: the call that met it cost a spin-out (the waiting waves gave up after kSpinLimit polls) and is encoded again, and
                        // every later call of this encoder takes the ticket form (measured ~8 % of the coder kernel); UG_JPEG_TICKET=1 starts that way
                        fprintf(stderr, "[ug_mi355x] JPEG encoder %dx%d: a workgroup of the one-launch stream placement waited for a predecessor that had not been "
                                        "dispatched (workgroups did not start in index order); this call is encoded again and the encoder uses start-order tickets "
                                        "from now on (UG_JPEG_TICKET=1 selects that from the start)\n", e->width, e->height);
                        return ug_hip_jpeg_encoder_encode_batch(enc, in, frames, src_dev, src_pitch, src_stride, out_dev, out_stride, out_capacity, out_len, stream);
                }
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: the stream placement gave up waiting for an earlier workgroup");
                return UG_HIP_ERUNTIME;
        }
        bool fits = true;
        for (int f = 0; f < frames; f++) {
                out_len[f] = e->total_host[f];
                fits = fits && out_len[f] <= out_capacity;
        }
        if (!fits && frames == 1) { // segments past the end were not written; out_len tells the caller what it takes
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: stream does not fit the output buffer (out_len = needed size)");
                return UG_HIP_EINVAL;
        }
        // a batch reports per frame: out_len[f] > out_capacity = that stream did not fit (the others are complete)
        return UG_HIP_SUCCESS;
}

} // extern "C"
