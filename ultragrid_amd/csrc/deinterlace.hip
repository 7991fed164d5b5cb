// deinterlace.hip -- vc_deinterlace (src/video_codec.c:597-664) on the device: the linear-blend de-interlace RTDXT applies to
// INTERLACED_MERGED input before it encodes (src/video_compress/dxt_glsl.cpp:195-201,291-293).
//
// What the reference's x86-64 build computes is its SSE2 body (:624-720), not the plain C loop beside it: IN PLACE, 16-byte column by 16-byte
// column, down the lines with pavgb ((a + b + 1) >> 1):
//     x0 = line 0, x1 = line 1;  for (j = 0; j < lines - 4; j += 2) {
//         x2 = line j+2;  x0 = avg(avg(x0, x2), x1);  x1 = line j+3;  line j+1 = x0;  x0 = avg(avg(x0, x1), x2);  line j+2 = x0;  }
// A recursive filter down the picture -- every output line feeds the next --, independent from byte to byte along a line: one lane per 4
// bytes of a line (v_lerp_u8 is pavgb on four bytes), and only 15-30 waves' worth of columns in a 1080-line picture.  Walked serially, a
// column costs 538 steps of 4 dependent operations, each waiting for lines from memory (the vmcnt counter of gfx9 counts the stores of the
// steps before with the loads, so a wave cannot keep more than ~30 lines in flight): 104 us for one 1080-line frame however far ahead the
// lines were requested.  The height is therefore cut into segments that run at the same time, which the arithmetic allows exactly:
//     k operations x -> (x + a + 1) >> 1 in a row are x -> (x + C) >> k with C = sum (a_i + 1) << i   (nested floors of halves collapse),
// so after 8 of them a byte that came in as x in [0, 255] leaves as (C >> 8) + ((x + (C & 255)) >> 8): one of two neighbouring values,
// chosen by a threshold on x.  Every further operation maps the two values on; the threshold stays.  A segment's effect on the value that
// enters it is thus three bytes per byte (the two values after the segment and C & 255), computed from the segment's own lines without
// knowing what enters.  16 waves of a workgroup take a segment of <= 34 steps each (their lines in registers, fetched at once), publish
// that summary in LDS, chain the summaries of the waves above them to the value entering their own segment (16 cheap steps instead of 538
// dependent ones) and then run the real recurrence over their registers, storing as they go.  Pictures higher than 16 x 34 steps take
// several rounds.  Bit-exact with the serial walk by construction; tests/test_deinterlace.py holds it to the oracle.
// The last 16-byte column of a line whose length is no multiple of 16 reaches into the beginning of the NEXT line, which the reference has
// filtered already when it gets there (columns are processed one after the other): a second, 16-lane launch after the first, as there.
#include "ug_common.h"

namespace {

constexpr int kWaves = 16;           // segments of a column in flight: waves of one workgroup
constexpr int kSeg = 34;             // steps (pairs of lines) a wave holds in registers; 16 x 34 = 544 steps >= the 538 of 1080 lines
constexpr uint32_t kEven = 0x00FF00FFu; // the even bytes of a word as two 16-bit fields

__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); } // per byte (a + b + 1) >> 1

// The value leaving a segment whose summary is (lo, hi, cm) when x enters it: per byte, hi where x + (C & 255) carries into bit 8, else lo.
__device__ __forceinline__ uint32_t through_segment(uint32_t lo, uint32_t hi, uint32_t cm, uint32_t x)
{
        const uint32_t se = (((x & kEven) + (cm & kEven)) >> 8) & 0x00010001u;
        const uint32_t so = ((((x >> 8) & kEven) + ((cm >> 8) & kEven)) >> 8) & 0x00010001u;
        const uint32_t mask = (se | (so << 8)) * 255u; // 0x00 / 0xFF per byte
        return (hi & mask) | (lo & ~mask);
}

// T = uint32_t (4 bytes per lane, columns [0, cols) in units of 4 bytes) or uint8_t (one byte per lane: the tail column, odd line sizes;
// the upper three bytes of every word are then zero and stay zero through avg4).
template <class T>
__global__ __launch_bounds__(kWaves * 64) void deinterlace_kernel(uint8_t *__restrict__ base, long linesize, int lines, long first_byte, long cols, size_t frame_stride)
{
        __shared__ uint32_t s_lo[2][kWaves * 64], s_hi[2][kWaves * 64], s_cm[2][kWaves * 64]; // by round parity: one barrier per round
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        long c = (long) blockIdx.x * 64 + lane;
        const bool live = c < cols; // lanes past the last column read the last column and store nothing (they must reach the barriers)
        if (!live) c = cols - 1;
        T *const col = (T *) (base + (size_t) blockIdx.y * frame_stride + first_byte + c * (long) sizeof(T));
        const long step = linesize / (long) sizeof(T); // (T = uint32_t only when linesize % 4 == 0)
        auto at = [&](int line) -> T & { return col[(long) line * step]; };
        const int steps = (lines - 3) / 2; // iterations of the reference's loop: j = 0, 2, ... < lines - 4; step s reads lines 2 s + 2, 2 s + 3
        const int last = 2 * steps + 1; // the last line any step reads (nobody writes it); nothing past it is touched
        const int rounds = (steps + kWaves * kSeg - 1) / (kWaves * kSeg);
        uint32_t carry = at(0); // the value entering the round: x0 of the reference
#pragma unroll 1
        for (int r = 0; r < rounds; r++) {
                // this round's steps [r0, r1), dealt to `act` waves of >= 2 steps each (a summary needs 8 operations = 2 steps)
                const int r0 = (int) ((long) steps * r / rounds), r1 = (int) ((long) steps * (r + 1) / rounds), m = r1 - r0;
                const int act = max(1, min(kWaves, m / 2)), share = m / act, extra = m % act;
                const int n = wv < act ? share + (wv < extra) : 0;    // <= kSeg
                const int sa = r0 + wv * share + min(wv, extra);      // first step of the segment
                // the segment's lines, none of them written before this round's barrier: line 2 sa + 1 (x1) and lines 2 sa + 2 ... No
                // conditions: registers past the segment's end hold lines nobody uses
                uint32_t x1 = (uint32_t) at(min(2 * sa + 1, last));
                uint32_t buf[2 * kSeg];
#pragma unroll
                for (int i = 0; i < 2 * kSeg; i++) buf[i] = (uint32_t) at(min(2 * sa + 2 + i, last));
                const int par = r & 1;
                if (act > 1) {
                        uint32_t lo = 0, hi = 0, cm = 0;
                        if (n > 0) { // n >= 2
                                // the first two steps' eight operands in order: x2 x1 x3 x2 | x2' x3 x3' x2'
                                const uint32_t a[8] = { buf[0], x1, buf[1], buf[0], buf[2], buf[1], buf[3], buf[2] };
                                uint32_t ce = 0, co = 0; // C per byte in 16-bit fields: at most 256 * 255, no carry between fields
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                        ce += ((a[j] & kEven) + 0x00010001u) << j;
                                        co += (((a[j] >> 8) & kEven) + 0x00010001u) << j;
                                }
                                lo = ((ce >> 8) & kEven) | (((co >> 8) & kEven) << 8);                       // what 0 becomes
                                hi = (((ce + kEven) >> 8) & kEven) | ((((co + kEven) >> 8) & kEven) << 8);   // what 255 becomes
                                cm = (ce & kEven) | ((co & kEven) << 8);
#pragma unroll
                                for (int i = 2; i < kSeg; i++) {
                                        if (i < n) { // wave-uniform
                                                const uint32_t x2 = buf[2 * i], xp = buf[2 * i - 1], x3 = buf[2 * i + 1];
                                                lo = avg4(avg4(avg4(avg4(lo, x2), xp), x3), x2);
                                                hi = avg4(avg4(avg4(avg4(hi, x2), xp), x3), x2);
                                        }
                                }
                        }
                        s_lo[par][wv * 64 + lane] = lo;
                        s_hi[par][wv * 64 + lane] = hi;
                        s_cm[par][wv * 64 + lane] = cm;
                }
                __syncthreads(); // every line of the round is in registers: stores may begin
                uint32_t x0 = carry;
                if (act > 1) {
                        uint32_t x = carry;
#pragma unroll
                        for (int k = 0; k < kWaves; k++) {
                                if (k < act) { // wave-uniform
                                        if (k == wv) x0 = x;
                                        x = through_segment(s_lo[par][k * 64 + lane], s_hi[par][k * 64 + lane], s_cm[par][k * 64 + lane], x);
                                }
                        }
                        carry = x; // (with one active wave there is one round, and nobody needs it)
                }
#pragma unroll
                for (int i = 0; i < kSeg; i++) {
                        if (i < n) { // wave-uniform
                                const int s = sa + i;
                                const uint32_t x2 = buf[2 * i], x3 = buf[2 * i + 1];
                                x0 = avg4(avg4(x0, x2), x1);
                                x1 = x3;
                                if (live) at(2 * s + 1) = (T) x0;
                                x0 = avg4(avg4(x0, x1), x2);
                                if (live) at(2 * s + 2) = (T) x0;
                        }
                }
        }
}

} // namespace

extern "C" int ug_hip_deinterlace_blend_batch(void *frame_dev, size_t linesize, int lines, int frames, size_t frame_stride, ug_hip_stream_t stream)
{
        if (lines < 0 || lines > ug::kMaxDim || linesize > 8ull * ug::kMaxDim || !ug::span_ok((long long) linesize, lines)) {
                return ug::refuse_size("ug_hip_deinterlace_blend"); // (8 bytes per pixel: the widest line of the library)
        }
        if (!frame_dev || linesize == 0 || lines < 0 || frames < 0 || frames > 65535 || (frames > 1 && frame_stride < linesize * (size_t) lines)) {
                ug::set_last_error_msg("ug_hip_deinterlace_blend: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (lines < 5 || frames == 0) return UG_HIP_SUCCESS; // vc_deinterlace changes nothing below 5 lines (its loop runs for j < lines - 4)
        if (linesize < 16) {
                // A line shorter than one 16-byte column: the reference's vectors then overlap THEMSELVES from line to line (what one iteration
                // stores the next one loads again inside the same column) and, below 6 bytes, are stored past the end of the frame.  Not a
                // picture (8 UYVY pixels are 16 bytes), not a contract: refused.
                ug::set_last_error_msg("ug_hip_deinterlace_blend: lines shorter than 16 bytes are not supported (the reference's 16-byte columns overlap themselves there)");
                return UG_HIP_EINVAL;
        }
        hipStream_t st = (hipStream_t) stream;
        uint8_t *base = (uint8_t *) frame_dev;
        const long full = (long) (linesize / 16) * 16; // the bytes of a line that lie in whole 16-byte columns
        const bool words = linesize % 4 == 0 && ((uintptr_t) frame_dev & 3) == 0 && frame_stride % 4 == 0;
        if (full > 0) {
                if (words) {
                        hipLaunchKernelGGL(deinterlace_kernel<uint32_t>, dim3((unsigned) ((full / 4 + 63) / 64), (unsigned) frames), dim3(kWaves * 64), 0, st, base, (long) linesize, lines, 0L,
                                           full / 4, frame_stride);
                } else {
                        hipLaunchKernelGGL(deinterlace_kernel<uint8_t>, dim3((unsigned) ((full + 63) / 64), (unsigned) frames), dim3(kWaves * 64), 0, st, base, (long) linesize, lines, 0L, full,
                                           frame_stride);
                }
        }
        if (full != (long) linesize) { // the column that reaches into the next line: after the others, as in the reference
                hipLaunchKernelGGL(deinterlace_kernel<uint8_t>, dim3(1, (unsigned) frames), dim3(kWaves * 64), 0, st, base, (long) linesize, lines, full, 16L, frame_stride);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

extern "C" int ug_hip_deinterlace_blend(void *frame_dev, size_t linesize, int lines, ug_hip_stream_t stream)
{
        return ug_hip_deinterlace_blend_batch(frame_dev, linesize, lines, 1, 0, stream);
}
