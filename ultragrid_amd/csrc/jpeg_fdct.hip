// jpeg_fdct.hip -- JPEG 8x8 forward DCT + quantisation on gfx950.
//
// UltraGrid gets this stage from the external libgpujpeg (gpujpeg_encoder_encode,
// src/video_compress/gpujpeg.cpp:624); it is not in the reference tree.  The stage is
// specified in oracle/jpeg_oracle.c (level shift, AAN float FDCT rows-then-columns, fp32
// reciprocal quantiser with the AAN scale folded in, rintf, zig-zag) and this file is
// bit-identical to that specification: same operation order, no FMA contraction
// (-ffp-contract=off).
//
// Mapping: one lane owns one 8x8 block (64 fp32 registers).  Consecutive lanes own
// consecutive blocks of a block row, so each of the 8 row loads of a wave is 64 x 8 B of
// contiguous plane memory.  The stage is HBM-bound (about 12 lane-ops per pixel against
// 1 B read + 2 B written per sample): see DESIGN.md.
#include "ug_common.h"
#include "jpeg_fdct_device.h"

#include <cstring>

namespace {

using namespace ug_jpeg;

// A wave holds 64 consecutive blocks (one per lane) = one contiguous 8 KiB stretch of the output.  Stored straight
// from registers every store instruction would touch 64 different 128-byte lines (16 B each); instead the wave
// transposes through LDS (row pitch 144 B: conflict-free 128-bit writes) so that each store instruction writes
// 1 KiB of contiguous memory -- 32 blocks at a time (round 4): the lower half of the lanes hands its blocks over, all 64 lanes
// store them, then the upper half: 4.5 KB of LDS per wave instead of 9, whole lines per store instruction as before, and a
// sixth workgroup of the fused UYVY kernel fits the CU.  `lds` = this wave's private kStoreRows x 144 B region; `n_valid`
// lanes hold real blocks.
#ifndef UG_JPEG_STORE_HALVES
#define UG_JPEG_STORE_HALVES 1
#endif
constexpr int kStoreRows = UG_JPEG_STORE_HALVES ? 32 : 64;
__device__ __forceinline__ void wave_store_blocks(const uint32_t (&w)[32], uint8_t *lds, int16_t *__restrict__ out_wave,
                                                  int lane, int n_valid)
{
#if UG_JPEG_STORE_HALVES
#pragma unroll
        for (int half = 0; half < 2; half++) {
                if ((lane >> 5) == half) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                                *(uint4 *) (lds + (lane & 31) * kLdsPitch + 16 * j) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
                        }
                }
                __builtin_amdgcn_wave_barrier(); // same wave wrote and reads: only ordering inside the wave is needed
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                        const int row = 8 * j + (lane >> 3), piece = lane & 7, blk = 32 * half + row;
                        const uint4 v = *(const uint4 *) (lds + row * kLdsPitch + 16 * piece);
                        if (blk < n_valid) {
                                ((uint4 *) out_wave)[8 * blk + piece] = v;
                        }
                }
                __builtin_amdgcn_wave_barrier(); // the rows are overwritten by the other half
                __builtin_amdgcn_s_waitcnt(0xc07f);
        }
#else
#pragma unroll
        for (int j = 0; j < 8; j++) {
                *(uint4 *) (lds + lane * kLdsPitch + 16 * j) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
        }
        __builtin_amdgcn_wave_barrier(); // same wave wrote and reads: only ordering inside the wave is needed
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
#pragma unroll
        for (int j = 0; j < 8; j++) {
                const int blk = 8 * j + (lane >> 3), piece = lane & 7;
                const uint4 v = *(const uint4 *) (lds + blk * kLdsPitch + 16 * piece);
                if (blk < n_valid) {
                        ((uint4 *) out_wave)[8 * blk + piece] = v;
                }
        }
#endif
}

// the same for a wave whose lower 32 lanes hold blocks of one plane and whose upper 32 lanes hold blocks of another (Cb | Cr of a strip)
__device__ __forceinline__ void wave_store_blocks_two(const uint32_t (&w)[32], uint8_t *lds, int16_t *__restrict__ out_lo, int16_t *__restrict__ out_hi,
                                                      int lane, int n_valid_each)
{
#pragma unroll
        for (int half = 0; half < 2; half++) {
                if ((lane >> 5) == half) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                                *(uint4 *) (lds + (lane & 31) * kLdsPitch + 16 * j) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
                        }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                int16_t *const out = half ? out_hi : out_lo;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                        const int row = 8 * j + (lane >> 3), piece = lane & 7;
                        const uint4 v = *(const uint4 *) (lds + row * kLdsPitch + 16 * piece);
                        if (row < n_valid_each) {
                                ((uint4 *) out)[8 * row + piece] = v;
                        }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
        }
}

__device__ __forceinline__ void quant_store(const float (&b)[64], const float *__restrict__ div, int16_t *__restrict__ out)
{
        uint32_t w[32];
        quant_pack(b, div, w);
        uint4 *o = (uint4 *) out;
#pragma unroll
        for (int i = 0; i < 8; i++) {
                o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
        }
}

// xstride = byte step between horizontally adjacent samples: 1 for a plane, 3 for one component of packed RGB.
__global__ __launch_bounds__(256) void fdct_quant_plane_kernel(const uint8_t *__restrict__ plane, int pitch, int xstride, int width, int height,
                                                               int blocks_w, long total, const float *__restrict__ div,
                                                               int16_t *__restrict__ out, float *__restrict__ coef)
{
        __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * kStoreRows * kLdsPitch];
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const long wave_first = idx - lane;
        uint32_t w[32];
        if (idx < total) {
                const int by = (int) (idx / blocks_w), bx = (int) (idx - (long) by * blocks_w);
                float b[64];
                const bool interior = xstride == 1 && 8 * bx + 8 <= width && 8 * by + 8 <= height && !(pitch & 7) && !(7 & (uintptr_t) plane);
                if (interior) {
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                                const uint2 q = *(const uint2 *) (plane + (long) (8 * by + r) * pitch + 8 * bx);
#pragma unroll
                                for (int c = 0; c < 4; c++) {
                                        b[8 * r + c] = (float) ((q.x >> (8 * c)) & 0xff);
                                        b[8 * r + 4 + c] = (float) ((q.y >> (8 * c)) & 0xff);
                                }
                        }
                } else { // edge replication
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                                const int y = min(8 * by + r, height - 1);
#pragma unroll
                                for (int c = 0; c < 8; c++) {
                                        const int x = min(8 * bx + c, width - 1);
                                        b[8 * r + c] = (float) plane[(long) y * pitch + (long) x * xstride];
                                }
                        }
                }
                fdct8x8(b);
                if (coef) {
#pragma unroll
                        for (int i = 0; i < 64; i++) coef[64 * idx + i] = b[i];
                }
                quant_pack(b, div, w);
        }
        if (wave_first < total) {
                const long left = total - wave_first;
                wave_store_blocks(w, lds_all + wave * kStoreRows * kLdsPitch, out + 64 * wave_first, lane, left < 64 ? (int) left : 64);
        }
}

// Packed RGB -> three component planes' worth of blocks (4:4:4, components stay R, G, B) in one pass: a lane holds one 8x8
// pixel block = 8 rows x 24 B in 48 registers and runs the three DCTs one after another; a wave's 64 blocks are consecutive,
// so its row loads cover one contiguous 1.5 KiB stretch per row and its stores go through wave_store_blocks().
// 3 B/px read + 6 B/px written.
__global__ __launch_bounds__(256) void rgb_jpeg444_kernel(const uint8_t *__restrict__ src, int pitch, int width, int height, int blocks_w,
                                                          long total, const float *__restrict__ div, int16_t *__restrict__ out0,
                                                          int16_t *__restrict__ out1, int16_t *__restrict__ out2)
{
        __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * kStoreRows * kLdsPitch];
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const long wave_first = idx - lane;
        if (wave_first >= total) return; // wave-uniform
        uint8_t *lds = lds_all + wave * kStoreRows * kLdsPitch;
        const long left = total - wave_first;
        const int n_valid = left < 64 ? (int) left : 64;
        uint32_t raw[8][6];
        if (idx < total) {
                const int by = (int) (idx / blocks_w), bx = (int) (idx - (long) by * blocks_w);
                const bool interior = 8 * bx + 8 <= width && 8 * by + 8 <= height && !(pitch & 3) && !(3 & (uintptr_t) src);
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
                                        const uint8_t *px = row + 3L * min(8 * bx + c, width - 1);
#pragma unroll
                                        for (int comp = 0; comp < 3; comp++) {
                                                const int bi = 3 * c + comp;
                                                raw[r][bi >> 2] |= (uint32_t) px[comp] << (8 * (bi & 3));
                                        }
                                }
                        }
                }
        }
#pragma unroll
        for (int comp = 0; comp < 3; comp++) {
                uint32_t w[32];
                if (idx < total) {
                        float b[64];
#pragma unroll
                        for (int r = 0; r < 8; r++) {
#pragma unroll
                                for (int c = 0; c < 8; c++) {
                                        const int bi = 3 * c + comp;
                                        b[8 * r + c] = (float) ((raw[r][bi >> 2] >> (8 * (bi & 3))) & 0xff);
                                }
                        }
                        fdct8x8(b);
                        quant_pack(b, div, w);
                }
                wave_store_blocks(w, lds, (comp == 0 ? out0 : (comp == 1 ? out1 : out2)) + 64 * wave_first, lane, n_valid);
                __builtin_amdgcn_wave_barrier(); // the LDS region is reused by the next component
        }
}

// batches: frame f reads src + f * src bytes and writes its coefficient planes f * luma / chroma bytes further on
struct FrameStrides {
        size_t src, luma, chroma; // bytes
        __device__ __forceinline__ void apply(unsigned f, const uint8_t *__restrict__ &s, int16_t *__restrict__ &y, int16_t *__restrict__ &cb,
                                              int16_t *__restrict__ &cr) const
        {
                s += (size_t) f * src;
                y = (int16_t *) ((uint8_t *) y + (size_t) f * luma);
                cb = (int16_t *) ((uint8_t *) cb + (size_t) f * chroma);
                cr = (int16_t *) ((uint8_t *) cr + (size_t) f * chroma);
        }
};

// Fused UYVY -> 4:2:0 / 4:2:2 planar -> FDCT+quant.  Tasks [0, n_luma) are luma blocks (8 rows x 16 B of UYVY),
// tasks [n_luma, n_luma + 2*n_chroma) are Cb then Cr blocks.  SUB = 420: chroma block = 16 rows x 32 B with the vertical
// (a+b+1)/2 average of uyvy_to_i420 (to_planar.c:343-378), MCU 16x16.  SUB = 422: chroma block = 8 rows x 32 B, samples
// taken as they are (uyvy_to_i422, video_codec.c:949-969), MCU 16x8.  Edges replicate.
template <int SUB>
__global__ __launch_bounds__(256) void uyvy_jpeg_kernel(const uint8_t *__restrict__ src, int pitch, int width, int height,
                                                        int mcu_w, int mcu_h, const float *__restrict__ div,
                                                        int16_t *__restrict__ out_y, int16_t *__restrict__ out_cb,
                                                        int16_t *__restrict__ out_cr, FrameStrides fs)
{
        fs.apply(blockIdx.y, src, out_y, out_cb, out_cr); // blockIdx.y = frame of the batch
        const long n_chroma = (long) mcu_w * mcu_h, n_luma = (SUB == 420 ? 4L : 2L) * n_chroma;
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= n_luma + 2 * n_chroma) return;
        float b[64];
        const int cw = (width + 1) / 2, ch = (height + 1) / 2; // chroma plane size (ch: 4:2:0 only)
        if (idx < n_luma) {
                const int bw = 2 * mcu_w;
                const int by = (int) (idx / bw), bx = (int) (idx - (long) by * bw);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                        const int y = min(8 * by + r, height - 1);
                        const uint8_t *row = src + (long) y * pitch;
                        if (8 * bx + 8 <= width && !(pitch & 15) && !(15 & (uintptr_t) src)) {
                                const uint4 q = *(const uint4 *) (row + 16 * bx);
                                const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                        b[8 * r + 2 * k] = (float) ((w[k] >> 8) & 0xff);
                                        b[8 * r + 2 * k + 1] = (float) (w[k] >> 24);
                                }
                        } else {
#pragma unroll
                                for (int c = 0; c < 8; c++) {
                                        const int x = min(8 * bx + c, width - 1);
                                        b[8 * r + c] = (float) row[2 * x + 1];
                                }
                        }
                }
                fdct8x8(b);
                quant_store(b, div, out_y + 64 * idx);
        } else {
                long t = idx - n_luma;
                const int comp = t >= n_chroma; // 0 = Cb, 1 = Cr
                t -= comp ? n_chroma : 0;
                const int by = (int) (t / mcu_w), bx = (int) (t - (long) by * mcu_w);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                        int y0, y1;
                        if (SUB == 420) {
                                const int cy = min(8 * by + r, ch - 1);
                                y0 = 2 * cy; y1 = min(2 * cy + 1, height - 1); // odd height: last line doubled
                        } else {
                                y0 = y1 = min(8 * by + r, height - 1);
                        }
                        const uint8_t *r0 = src + (long) y0 * pitch, *r1 = src + (long) y1 * pitch;
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                                const int cx = min(8 * bx + c, cw - 1);
                                const int a = r0[4 * cx + 2 * comp], bb = r1[4 * cx + 2 * comp];
                                b[8 * r + c] = (float) (SUB == 420 ? (a + bb + 1) >> 1 : a);
                        }
                }
                fdct8x8(b);
                quant_store(b, div + 64, (comp ? out_cr : out_cb) + 64 * t);
        }
}

// MCU-aligned fast path of the fused kernel (width % 16 == 0, 16-byte aligned lines).
// Workgroup = kLumaWaves + 1 waves over a strip of 32 MCUs (512 px x 16 rows for 4:2:0, x 8 rows for 4:2:2): the luma
// waves take one luma block row of the strip each (64 blocks), the last wave = 32 Cb blocks (lanes 0-31) + 32 Cr blocks
// (lanes 32-63).  Every wave does 64 block DCTs, reads its rows with 128-bit loads that are contiguous across lanes (the
// chroma wave re-reads the strip from L1/L2, so HBM sees each input byte once) and writes through wave_store_blocks().
// Occupancy: capped at 3 waves per SIMD.  Unlike the light pixel-format kernels (which want all 8), this one is slower with more waves
// resident: 7 / 6 / 5 (the allocator's own choice: 81 VGPRs) / 4 / 3 / 2 waves per SIMD measure 73.0 / 64.3 / 62.9 / 62.0 / 57.1 (58.8 for 5 on
// that box) / 62.7 us per 8 4K frames, interleaved A/B (profiles/r05_jpeg_front_end_occupancy.txt): 0.705 -> 0.727 of 8 TB/s.
// The cap applies to launches of two frames or more (uyvy_jpeg_fast_batch_kernel).  One frame per launch keeps the allocator's choice: the 4 050 waves of a 4K 4:2:2 frame are
// 15.8 per CU -- one round of residency at 5 per SIMD, two at 3 (14.0 -> 14.8 us); a 4:2:0 frame (11.9 per CU) does not care.
template <int SUB>
__device__ __forceinline__ void uyvy_jpeg_fast_body(const uint8_t *__restrict__ src, int pitch, int height, int mcu_w, const float *__restrict__ div,
                                                    int16_t *__restrict__ out_y, int16_t *__restrict__ out_cb, int16_t *__restrict__ out_cr, FrameStrides fs)
{
        fs.apply(blockIdx.z, src, out_y, out_cb, out_cr); // blockIdx.z = frame of the batch
        constexpr int kLumaWaves = SUB == 420 ? 2 : 1;
        __shared__ __attribute__((aligned(16))) uint8_t lds_all[(kLumaWaves + 1) * kStoreRows * kLdsPitch];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int mcu0 = blockIdx.x * 32, my = blockIdx.y;
        const int mcus = min(32, mcu_w - mcu0); // MCUs of this strip that exist
        uint8_t *lds = lds_all + wave * kStoreRows * kLdsPitch;
        float b[64];
        uint32_t w[32];
        if (wave < kLumaWaves) {
                const int bx = 2 * mcu0 + lane; // luma block column
                const bool valid = lane < 2 * mcus;
                const int brow = kLumaWaves * my + wave; // luma block row
                if (valid) {
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                                const int y = min(8 * brow + r, height - 1);
                                const uint4 q = *(const uint4 *) (src + (long) y * pitch + 16 * bx);
                                const uint32_t ww[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                        b[8 * r + 2 * k] = (float) ((ww[k] >> 8) & 0xff);
                                        b[8 * r + 2 * k + 1] = (float) (ww[k] >> 24);
                                }
                        }
                        fdct8x8(b);
                        quant_pack(b, div, w);
                }
                const long first = (long) brow * (2 * mcu_w) + 2 * mcu0;
                wave_store_blocks(w, lds, out_y + 64 * first, lane, 2 * mcus);
        } else {
                const int comp = lane >> 5, m = lane & 31; // 0 = Cb, 1 = Cr ; MCU within the strip
                const bool valid = m < mcus;
                if (valid) {
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                                int y0, y1;
                                if (SUB == 420) {
                                        const int cy = min(8 * my + r, (height + 1) / 2 - 1); // edge replication on the chroma plane
                                        y0 = 2 * cy; y1 = min(2 * cy + 1, height - 1);        // odd height: last line doubled
                                } else {
                                        y0 = y1 = min(8 * my + r, height - 1);
                                }
                                // this lane's half of the MCU's 32-byte row piece (Cb lanes the first 16 bytes, Cr lanes the second)
                                const uint4 a0 = *(const uint4 *) (src + (long) y0 * pitch + 32 * (mcu0 + m) + 16 * comp);
                                uint32_t wa[4] = { a0.x, a0.y, a0.z, a0.w };
                                if (SUB == 420) { // (a + b + 1) / 2 of uyvy_to_i420 (to_planar.c:364-367), all four bytes of a word at once
                                        const uint4 c0 = *(const uint4 *) (src + (long) y1 * pitch + 32 * (mcu0 + m) + 16 * comp);
                                        wa[0] = avg_bytes(wa[0], c0.x); wa[1] = avg_bytes(wa[1], c0.y);
                                        wa[2] = avg_bytes(wa[2], c0.z); wa[3] = avg_bytes(wa[3], c0.w);
                                } // else uyvy_to_i422 (video_codec.c:949-969): samples as they are
                                chroma_row_from_uyvy(wa, b + 8 * r);
                        }
                        fdct8x8(b);
                        quant_pack(b, div + 64, w);
                }
                // lanes 0-31 -> Cb blocks, lanes 32-63 -> Cr blocks of this strip: two contiguous 4 KiB stretches
                const long first = (long) my * mcu_w + mcu0;
#if UG_JPEG_STORE_HALVES
                wave_store_blocks_two(w, lds, out_cb + 64 * first, out_cr + 64 * first, lane, mcus);
#else
#pragma unroll
                for (int j = 0; j < 8; j++) {
                        *(uint4 *) (lds + lane * kLdsPitch + 16 * j) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                        const int blk = 8 * j + (lane >> 3), piece = lane & 7; // blk 0-31 Cb, 32-63 Cr
                        const uint4 v = *(const uint4 *) (lds + blk * kLdsPitch + 16 * piece);
                        const int mm = blk & 31;
                        if (mm < mcus) {
                                int16_t *o = (blk < 32 ? out_cb : out_cr) + 64 * (first + mm);
                                ((uint4 *) o)[piece] = v;
                        }
                }
#endif
        }
}

// one frame per launch: the allocator's own choice (81 VGPRs, 5 waves per SIMD)
template <int SUB>
__global__ __launch_bounds__(SUB == 420 ? 192 : 128) void uyvy_jpeg_fast_kernel(const uint8_t *__restrict__ src, int pitch, int height, int mcu_w,
                                                                                const float *__restrict__ div, int16_t *__restrict__ out_y,
                                                                                int16_t *__restrict__ out_cb, int16_t *__restrict__ out_cr, FrameStrides fs)
{
        uyvy_jpeg_fast_body<SUB>(src, pitch, height, mcu_w, div, out_y, out_cb, out_cr, fs);
}
// two frames or more per launch: capped at 3 waves per SIMD
template <int SUB>
__global__ __attribute__((amdgpu_waves_per_eu(3, 3))) __launch_bounds__(SUB == 420 ? 192 : 128) void uyvy_jpeg_fast_batch_kernel(
        const uint8_t *__restrict__ src, int pitch, int height, int mcu_w, const float *__restrict__ div, int16_t *__restrict__ out_y,
        int16_t *__restrict__ out_cb, int16_t *__restrict__ out_cr, FrameStrides fs)
{
        uyvy_jpeg_fast_body<SUB>(src, pitch, height, mcu_w, div, out_y, out_cb, out_cr, fs);
}

// T.81 Annex K tables (natural order) -- same data as oracle/jpeg_oracle.c by construction of the
// standard; kept separately so the product never links the oracle.
const uint8_t kLuma[64] = {
        16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
        18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99,
};
const uint8_t kChroma[64] = {
        17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
};
const double kAan[8] = { 1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379 };

} // namespace

namespace {
template <int SUB>
int launch_uyvy_jpeg(const void *src, int src_pitch, int width, int height, const float *div, int16_t *out_y, int16_t *out_cb,
                     int16_t *out_cr, int frames, const FrameStrides &fs, ug_hip_stream_t stream, const char *who)
{
        if (!ug::dims_ok(width, height) || src_pitch < 0 || !ug::span_ok(src_pitch ? src_pitch : ug::linesize(UG_PF_UYVY, width), height) ||
            !ug::span_ok(2LL * ((width + 15) / 16 * 16), (height + 15) / 16 * 16)) { // (the luma coefficients: an int16 per padded sample)
                return ug::refuse_size(who);
        }
        if (!src || !div || !out_y || !out_cb || !out_cr || width <= 0 || height <= 0 || frames < 0 || frames > 65535 ||
            ((uintptr_t) out_y | (uintptr_t) out_cb | (uintptr_t) out_cr) & 15 || (frames > 1 && ((fs.luma | fs.chroma) & 15))) {
                ug::set_last_error_msg(who);
                return UG_HIP_EINVAL;
        }
        if (frames == 0) return UG_HIP_SUCCESS;
        if (!src_pitch) src_pitch = ug::linesize(UG_PF_UYVY, width);
        const int mcu_w = (width + 15) / 16, mcu_h = SUB == 420 ? (height + 15) / 16 : (height + 7) / 8;
        if (width % 16 == 0 && !(src_pitch & 15) && !(15 & (uintptr_t) src) && (frames == 1 || !(fs.src & 15))) {
                const dim3 grid((unsigned) ((mcu_w + 31) / 32), (unsigned) mcu_h, (unsigned) frames), wg(SUB == 420 ? 192 : 128);
                if (frames >= 2) {
                        hipLaunchKernelGGL((uyvy_jpeg_fast_batch_kernel<SUB>), grid, wg, 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, height, mcu_w, div, out_y, out_cb, out_cr, fs);
                } else {
                        hipLaunchKernelGGL((uyvy_jpeg_fast_kernel<SUB>), grid, wg, 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, height, mcu_w, div, out_y, out_cb, out_cr, fs);
                }
                UG_HIP_LAUNCH_CHECK();
                return UG_HIP_SUCCESS;
        }
        const long total = (SUB == 420 ? 6L : 4L) * mcu_w * mcu_h;
        hipLaunchKernelGGL((uyvy_jpeg_kernel<SUB>), dim3((unsigned) ((total + 255) / 256), (unsigned) frames), dim3(256), 0, (hipStream_t) stream,
                           (const uint8_t *) src, src_pitch, width, height, mcu_w, mcu_h, div, out_y, out_cb, out_cr, fs);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}
} // namespace

// one component of a planar (xstride 1) or packed (xstride = bytes per pixel) 8-bit image -> quantised blocks
int ug::jpeg_fdct_quant_strided(const void *plane, int pitch, int xstride, int width, int height, int blocks_w, int blocks_h,
                                const float *div, int16_t *out, float *coef, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height) || blocks_w <= 0 || blocks_h <= 0 || blocks_w > ug::kMaxDim / 8 || blocks_h > ug::kMaxDim / 8 || !ug::span_ok(pitch, height) ||
            !ug::span_ok(128LL * blocks_w, blocks_h)) { // (a block is 64 int16 coefficients)
                return ug::refuse_size("ug_hip_jpeg_fdct_quant_plane");
        }
        if (!plane || !div || !out || width <= 0 || height <= 0 || xstride < 1 || blocks_w * 8 < width || blocks_h * 8 < height ||
            (15 & (uintptr_t) out) || pitch < width * xstride) {
                ug::set_last_error_msg("ug_hip_jpeg_fdct_quant_plane: bad arguments");
                return UG_HIP_EINVAL;
        }
        const long total = (long) blocks_w * blocks_h;
        hipLaunchKernelGGL(fdct_quant_plane_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                           (const uint8_t *) plane, pitch, xstride, width, height, blocks_w, total, div, out, coef);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

// packed RGB (3 B/px) -> quantised blocks of the R, G and B components, all with the divisors `div` (64 floats)
int ug::jpeg_fdct_quant_rgb444(const void *src, int pitch, int width, int height, int blocks_w, int blocks_h, const float *div,
                               int16_t *out_r, int16_t *out_g, int16_t *out_b, ug_hip_stream_t stream)
{
        if (!src || !div || !out_r || !out_g || !out_b || width <= 0 || height <= 0 || blocks_w * 8 < width || blocks_h * 8 < height ||
            (15 & ((uintptr_t) out_r | (uintptr_t) out_g | (uintptr_t) out_b)) || pitch < 3 * width) {
                ug::set_last_error_msg("jpeg_fdct_quant_rgb444: bad arguments");
                return UG_HIP_EINVAL;
        }
        const long total = (long) blocks_w * blocks_h;
        hipLaunchKernelGGL(rgb_jpeg444_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                           (const uint8_t *) src, pitch, width, height, blocks_w, total, div, out_r, out_g, out_b);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Colour stage of the JPEG encoder (gpujpeg.cpp:303-305,398-405: color_space_internal = Y601 / Y601full / Y709 / RGB; GPUJPEG converts the
// input to that space in its preprocessor, at full resolution, before it subsamples).  One affine map per pixel, out = M * in + offset, the
// twelve numbers derived on the host in double from the PUBLISHED definitions -- luma weights Kr, Kb of BT.601 (0.299, 0.114) and BT.709
// (0.2126, 0.0722), E'Cb = (B' - Y') / (2 (1 - Kb)), E'Cr = (R' - Y') / (2 (1 - Kr)); 8-bit limited range 16 + 219 E'Y, 128 + 224 E'C;
// "256 levels" (JFIF) 255 E'Y, 128 + 255 E'C -- and rounded to float.  Per sample: three multiply-adds in fp32, in the order written, clamp to
// [0, 255], round to nearest even.  Unpinned towards libgpujpeg like the FDCT (whose preprocessor source is not in the reference tree):
// the oracle restates it operation for operation, and an fp64 evaluation of the same definitions bounds both (tests/test_gpu_jpeg_colour.py).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {

struct ColourMap { float m[12]; };

// 3 x 4 affine map: full-range R'G'B' (0..255) -> the 8-bit Y'CbCr code values of `cs`
void rgb_to_ycbcr(int cs, double t[3][4])
{
        const bool bt709 = cs == UG_JPEG_CS_YCBCR_BT709, full = cs == UG_JPEG_CS_YCBCR_BT601_256LVLS;
        const double kr = bt709 ? 0.2126 : 0.299, kb = bt709 ? 0.0722 : 0.114, kg = 1.0 - kr - kb;
        const double ys = (full ? 255.0 : 219.0) / 255.0, cs_ = (full ? 255.0 : 224.0) / 255.0, y0 = full ? 0.0 : 16.0;
        const double row_y[3] = { kr, kg, kb };
        const double row_cb[3] = { -kr / (2 * (1 - kb)), -kg / (2 * (1 - kb)), 0.5 }, row_cr[3] = { 0.5, -kg / (2 * (1 - kr)), -kb / (2 * (1 - kr)) };
        for (int i = 0; i < 3; i++) {
                t[0][i] = ys * row_y[i];
                t[1][i] = cs_ * row_cb[i];
                t[2][i] = cs_ * row_cr[i];
        }
        t[0][3] = y0; t[1][3] = 128.0; t[2][3] = 128.0;
}

// inverse of an affine 3 x 4 map (Cramer; the maps above are far from singular)
void invert_affine(const double a[3][4], double inv[3][4])
{
        const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
        for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) {
                        const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
                        inv[i][j] = (a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]) / det;
                }
        }
        for (int i = 0; i < 3; i++) inv[i][3] = -(inv[i][0] * a[0][3] + inv[i][1] * a[1][3] + inv[i][2] * a[2][3]);
}

bool colour_map(int cs_in, int cs_out, ColourMap &out)
{
        auto known = [](int cs) { return cs >= UG_JPEG_CS_RGB && cs <= UG_JPEG_CS_YCBCR_BT709; };
        if (!known(cs_in) || !known(cs_out)) return false;
        double to_rgb[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } }, from_rgb[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } };
        if (cs_in != UG_JPEG_CS_RGB) {
                double f[3][4];
                rgb_to_ycbcr(cs_in, f);
                invert_affine(f, to_rgb);
        }
        if (cs_out != UG_JPEG_CS_RGB) rgb_to_ycbcr(cs_out, from_rgb);
        for (int i = 0; i < 3; i++) { // from_rgb o to_rgb
                for (int j = 0; j < 4; j++) {
                        double v = j == 3 ? from_rgb[i][3] : 0.0;
                        for (int k = 0; k < 3; k++) v += from_rgb[i][k] * to_rgb[k][j];
                        out.m[4 * i + j] = (float) v;
                }
        }
        return true;
}

__device__ __forceinline__ float affine_row(const float *m, float a, float b, float c)
{
        float t = m[0] * a;
        t = t + m[1] * b;
        t = t + m[2] * c;
        return t + m[3];
}
__device__ __forceinline__ uint32_t to_code(float v) { return (uint32_t) rintf(fminf(255.0f, fmaxf(0.0f, v))); }

// packed 3 B / px (R,G,B or Y,Cb,Cr) -> packed 3 B / px; one lane per pixel (lines of 3 * width bytes: no alignment to speak of)
__global__ __launch_bounds__(256) void colour_packed3_kernel(const uint8_t *__restrict__ src, int src_pitch, uint8_t *__restrict__ dst, int dst_pitch, int width, int height,
                                                             size_t src_stride, size_t dst_stride, ColourMap cm)
{
        const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
        if (x >= width || y >= height) return;
        const uint8_t *p = src + blockIdx.z * src_stride + (size_t) y * src_pitch + 3 * (size_t) x;
        uint8_t *o = dst + blockIdx.z * dst_stride + (size_t) y * dst_pitch + 3 * (size_t) x;
        const float a = (float) p[0], b = (float) p[1], c = (float) p[2];
        o[0] = (uint8_t) to_code(affine_row(cm.m, a, b, c));
        o[1] = (uint8_t) to_code(affine_row(cm.m + 4, a, b, c));
        o[2] = (uint8_t) to_code(affine_row(cm.m + 8, a, b, c));
}

// UYVY -> UYVY: each pixel of a pair with the pair's chroma (4:2:2 -> 4:4:4 by replication), mapped, the two chroma results averaged
// (mix(a, b, 0.5) as a * 0.5 + b * 0.5) back into one pair; one lane per pair
__global__ __launch_bounds__(256) void colour_uyvy_kernel(const uint8_t *__restrict__ src, int src_pitch, uint8_t *__restrict__ dst, int dst_pitch, int pairs, int height,
                                                          size_t src_stride, size_t dst_stride, ColourMap cm)
{
        const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
        if (x >= pairs || y >= height) return;
        const uint32_t q = *(const uint32_t *) (src + blockIdx.z * src_stride + (size_t) y * src_pitch + 4 * (size_t) x);
        const float u = (float) (q & 0xff), y0 = (float) ((q >> 8) & 0xff), v = (float) ((q >> 16) & 0xff), y1 = (float) (q >> 24);
        const float cb0 = affine_row(cm.m + 4, y0, u, v), cb1 = affine_row(cm.m + 4, y1, u, v);
        const float cr0 = affine_row(cm.m + 8, y0, u, v), cr1 = affine_row(cm.m + 8, y1, u, v);
        const float cb = cb0 * 0.5f + cb1 * 0.5f, cr = cr0 * 0.5f + cr1 * 0.5f;
        *(uint32_t *) (dst + blockIdx.z * dst_stride + (size_t) y * dst_pitch + 4 * (size_t) x) =
                to_code(cb) | to_code(affine_row(cm.m, y0, u, v)) << 8 | to_code(cr) << 16 | to_code(affine_row(cm.m, y1, u, v)) << 24;
}

// UYVY -> packed 3 B / px (4:2:2 -> 4:4:4: both pixels of a pair take the pair's chroma), MAP: through the colour map, else the samples as they are;
// one lane per pair; the last pair of an odd width has one pixel
template <bool MAP>
__global__ __launch_bounds__(256) void uyvy_to_444_kernel(const uint8_t *__restrict__ src, int src_pitch, uint8_t *__restrict__ dst, int dst_pitch, int width, int height,
                                                          size_t src_stride, size_t dst_stride, ColourMap cm)
{
        const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
        if (2 * x >= width || y >= height) return;
        const uint32_t q = *(const uint32_t *) (src + blockIdx.z * src_stride + (size_t) y * src_pitch + 4 * (size_t) x);
        uint8_t *o = dst + blockIdx.z * dst_stride + (size_t) y * dst_pitch + 6 * (size_t) x;
        const uint32_t u = q & 0xff, v = (q >> 16) & 0xff;
        for (int i = 0; i < 2 && 2 * x + i < width; i++) {
                const uint32_t l = i ? q >> 24 : (q >> 8) & 0xff;
                if (MAP) {
                        const float a = (float) l, b = (float) u, c = (float) v;
                        o[3 * i] = (uint8_t) to_code(affine_row(cm.m, a, b, c));
                        o[3 * i + 1] = (uint8_t) to_code(affine_row(cm.m + 4, a, b, c));
                        o[3 * i + 2] = (uint8_t) to_code(affine_row(cm.m + 8, a, b, c));
                } else {
                        o[3 * i] = (uint8_t) l; o[3 * i + 1] = (uint8_t) u; o[3 * i + 2] = (uint8_t) v;
                }
        }
}

} // namespace

int ug::jpeg_uyvy_to_444(int cs_out, const void *src, int src_pitch, void *dst, int dst_pitch, int width, int height, int frames, size_t src_stride, size_t dst_stride,
                         ug_hip_stream_t stream)
{
        ColourMap cm = {};
        const bool map = cs_out != UG_JPEG_CS_ASIS && cs_out != UG_JPEG_CS_YCBCR_BT709;
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_jpeg_encoder_encode");
        const int ls = (width + 1) / 2 * 4;
        if (!src_pitch) src_pitch = ls;
        if (!dst_pitch) dst_pitch = 3 * width;
        if (!src || !dst || frames < 1 || frames > 65535 || (map && !colour_map(UG_JPEG_CS_YCBCR_BT709, cs_out, cm)) || src_pitch < ls || dst_pitch < 3 * width ||
            !ug::span_ok(src_pitch, height) || !ug::span_ok(dst_pitch, height) || (src_pitch & 3) || (3 & (uintptr_t) src) || (frames > 1 && (src_stride & 3))) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: UYVY input of a 4:4:4 encoder: 4-byte aligned lines");
                return UG_HIP_EINVAL;
        }
        const dim3 grid((unsigned) (((width + 1) / 2 + 255) / 256), (unsigned) height, (unsigned) frames);
        if (map) {
                hipLaunchKernelGGL(uyvy_to_444_kernel<true>, grid, dim3(256), 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, (uint8_t *) dst, dst_pitch, width, height,
                                   src_stride, dst_stride, cm);
        } else {
                hipLaunchKernelGGL(uyvy_to_444_kernel<false>, grid, dim3(256), 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, (uint8_t *) dst, dst_pitch, width, height,
                                   src_stride, dst_stride, cm);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int ug::jpeg_colour_convert(ug_pixfmt_t fmt, int cs_in, int cs_out, const void *src, int src_pitch, void *dst, int dst_pitch, int width, int height, int frames,
                            size_t src_stride, size_t dst_stride, ug_hip_stream_t stream)
{
        ColourMap cm;
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_jpeg_colour_convert");
        if (!src || !dst || frames < 1 || frames > 65535 || (fmt != UG_PF_RGB && fmt != UG_PF_UYVY) || !colour_map(cs_in, cs_out, cm)) {
                ug::set_last_error_msg("ug_hip_jpeg_colour_convert: RGB (3 B/px) or UYVY, colour spaces UG_JPEG_CS_RGB .. UG_JPEG_CS_YCBCR_BT709");
                return UG_HIP_EINVAL;
        }
        const int ls = fmt == UG_PF_RGB ? 3 * width : (width + 1) / 2 * 4;
        if (!src_pitch) src_pitch = ls;
        if (!dst_pitch) dst_pitch = ls;
        if (src_pitch < ls || dst_pitch < ls || !ug::span_ok(src_pitch, height) || !ug::span_ok(dst_pitch, height) ||
            (fmt == UG_PF_UYVY && ((width & 1) || ((src_pitch | dst_pitch) & 3) || (3 & ((uintptr_t) src | (uintptr_t) dst)) || (frames > 1 && ((src_stride | dst_stride) & 3))))) {
                ug::set_last_error_msg("ug_hip_jpeg_colour_convert: bad pitch / alignment (UYVY: even width, 4-byte aligned lines)");
                return UG_HIP_EINVAL;
        }
        const int units = fmt == UG_PF_RGB ? width : width / 2;
        const dim3 grid((unsigned) ((units + 255) / 256), (unsigned) height, (unsigned) frames);
        if (fmt == UG_PF_RGB) {
                hipLaunchKernelGGL(colour_packed3_kernel, grid, dim3(256), 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, (uint8_t *) dst, dst_pitch, width, height,
                                   src_stride, dst_stride, cm);
        } else {
                hipLaunchKernelGGL(colour_uyvy_kernel, grid, dim3(256), 0, (hipStream_t) stream, (const uint8_t *) src, src_pitch, (uint8_t *) dst, dst_pitch, units, height,
                                   src_stride, dst_stride, cm);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

extern "C" {

int ug_hip_jpeg_colour_matrix(int cs_in, int cs_out, float m[12])
{
        ColourMap cm;
        if (!m || !colour_map(cs_in, cs_out, cm)) {
                ug::set_last_error_msg("ug_hip_jpeg_colour_matrix: colour spaces UG_JPEG_CS_RGB .. UG_JPEG_CS_YCBCR_BT709");
                return UG_HIP_EINVAL;
        }
        memcpy(m, cm.m, sizeof cm.m);
        return UG_HIP_SUCCESS;
}

int ug_hip_jpeg_colour_convert(ug_pixfmt_t fmt, int cs_in, int cs_out, const void *src_dev, int src_pitch, void *dst_dev, int dst_pitch, int width, int height,
                               ug_hip_stream_t stream)
{
        return ug::jpeg_colour_convert(fmt, cs_in, cs_out, src_dev, src_pitch, dst_dev, dst_pitch, width, height, 1, 0, 0, stream);
}

void ug_hip_jpeg_qtable(int quality, int comp, uint8_t table[64])
{
        if (quality < 1) quality = 1;
        if (quality > 100) quality = 100;
        const int s = quality < 50 ? 5000 / quality : 200 - quality * 2; // IJG quality rule
        const uint8_t *base = comp == 0 ? kLuma : kChroma;
        for (int i = 0; i < 64; i++) {
                const int t = (base[i] * s + 50) / 100;
                table[i] = t < 1 ? 1 : (t > 255 ? 255 : t);
        }
}

void ug_hip_jpeg_divisors(const uint8_t q[64], float div[64])
{
        for (int r = 0; r < 8; r++) {
                for (int c = 0; c < 8; c++) {
                        div[8 * r + c] = (float) (1.0 / ((double) q[8 * r + c] * kAan[r] * kAan[c] * 8.0));
                }
        }
}

int ug_hip_jpeg_fdct_quant_plane(const void *plane, int pitch, int width, int height, int blocks_w, int blocks_h,
                                 const float *div, int16_t *out, float *coef, ug_hip_stream_t stream)
{
        return ug::jpeg_fdct_quant_strided(plane, pitch, 1, width, height, blocks_w, blocks_h, div, out, coef, stream);
}

int ug_hip_uyvy_to_jpeg420_coeffs(const void *src, int src_pitch, int width, int height, const float *div,
                                  int16_t *out_y, int16_t *out_cb, int16_t *out_cr, ug_hip_stream_t stream)
{
        return launch_uyvy_jpeg<420>(src, src_pitch, width, height, div, out_y, out_cb, out_cr, 1, FrameStrides{ 0, 0, 0 }, stream,
                                     "ug_hip_uyvy_to_jpeg420_coeffs: bad arguments");
}

int ug_hip_uyvy_to_jpeg422_coeffs(const void *src, int src_pitch, int width, int height, const float *div,
                                  int16_t *out_y, int16_t *out_cb, int16_t *out_cr, ug_hip_stream_t stream)
{
        return launch_uyvy_jpeg<422>(src, src_pitch, width, height, div, out_y, out_cb, out_cr, 1, FrameStrides{ 0, 0, 0 }, stream,
                                     "ug_hip_uyvy_to_jpeg422_coeffs: bad arguments");
}

int ug_hip_uyvy_to_jpeg42x_coeffs_batch(int subsampling, const void *src, int src_pitch, int width, int height, const float *div,
                                        int16_t *out_y, int16_t *out_cb, int16_t *out_cr, int frames, size_t src_frame_stride,
                                        size_t luma_frame_stride, size_t chroma_frame_stride, ug_hip_stream_t stream)
{
        const FrameStrides fs = { src_frame_stride, luma_frame_stride, chroma_frame_stride };
        if (subsampling == 420) {
                return launch_uyvy_jpeg<420>(src, src_pitch, width, height, div, out_y, out_cb, out_cr, frames, fs, stream,
                                             "ug_hip_uyvy_to_jpeg42x_coeffs_batch: bad arguments");
        }
        if (subsampling == 422) {
                return launch_uyvy_jpeg<422>(src, src_pitch, width, height, div, out_y, out_cb, out_cr, frames, fs, stream,
                                             "ug_hip_uyvy_to_jpeg42x_coeffs_batch: bad arguments");
        }
        ug::set_last_error_msg("ug_hip_uyvy_to_jpeg42x_coeffs_batch: subsampling must be 420 or 422");
        return UG_HIP_EINVAL;
}

} // extern "C"
