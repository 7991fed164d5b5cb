// deinterlace.hip -- vc_deinterlace (src/video_codec.c:597-664) on the device: the linear-blend de-interlace RTDXT applies to
// INTERLACED_MERGED input before it encodes (src/video_compress/dxt_glsl.cpp:195-201,291-293).
//
// What the reference's x86-64 build computes is its SSE2 body (:624-720), not the plain C loop beside it: IN PLACE, 16-byte column by 16-byte
// column, down the lines with pavgb ((a + b + 1) >> 1):
//     x0 = line 0, x1 = line 1;  for (j = 0; j < lines - 4; j += 2) {
//         x2 = line j+2;  x0 = avg(avg(x0, x2), x1);  x1 = line j+3;  line j+1 = x0;  x0 = avg(avg(x0, x1), x2);  line j+2 = x0;  }
// A recursive filter down the picture -- every output line feeds the next --, independent from byte to byte along a line.  So: one lane
// per 4 bytes of a line (v_lerp_u8 is pavgb on four bytes), each walking down its column; the lines a step reads have not been written
// yet, so they are fetched two steps ahead of the dependent chain.  The picture's height is the serial part: 1080 lines = 538 steps of
// 4 dependent operations; an interlaced frame is at most 1920 x 1080 in practice (15-30 waves), and frames of a batch go side by side.
// The last 16-byte column of a line whose length is no multiple of 16 reaches into the beginning of the NEXT line, which the reference has
// filtered already when it gets there (columns are processed one after the other): a second, 16-lane launch after the first, as there.
#include "ug_common.h"

namespace {

__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); } // per byte (a + b + 1) >> 1

// T = uint32_t (4 bytes per lane, columns [0, cols) in units of 4 bytes) or uint8_t (one byte per lane: the tail column, odd line sizes)
template <class T>
__global__ __launch_bounds__(64) void deinterlace_kernel(uint8_t *__restrict__ base, long linesize, int lines, long first_byte, long cols, size_t frame_stride)
{
        const long c = (long) blockIdx.x * 64 + threadIdx.x;
        if (c >= cols) return;
        T *const col = (T *) (base + (size_t) blockIdx.y * frame_stride + first_byte + c * (long) sizeof(T));
        const long step = linesize / (long) sizeof(T); // (T = uint32_t only when linesize % 4 == 0)
        auto at = [&](int line) -> T & { return col[(long) line * step]; };
        auto avg = [](uint32_t a, uint32_t b) -> uint32_t { return sizeof(T) == 4 ? avg4(a, b) : (a + b + 1u) >> 1; };
        if (lines < 5) return;
        uint32_t x0 = at(0), x1 = at(1);
        // two steps of look-ahead on the lines to come (they are only written behind the point where they are read)
        uint32_t n2 = at(2), n3 = at(3), m2 = 0, m3 = 0;
        if (lines - 4 > 2) { m2 = at(4); m3 = at(5); }
#pragma unroll 1
        for (int j = 0; j < lines - 4; j += 2) {
                const uint32_t x2 = n2, x3 = n3;
                n2 = m2; n3 = m3;
                if (j + 4 < lines - 4) { m2 = at(j + 6); m3 = at(j + 7); }
                x0 = avg(avg(x0, x2), x1);
                x1 = x3;
                at(j + 1) = (T) x0;
                x0 = avg(avg(x0, x1), x2);
                at(j + 2) = (T) x0;
        }
}

} // namespace

extern "C" int ug_hip_deinterlace_blend_batch(void *frame_dev, size_t linesize, int lines, int frames, size_t frame_stride, ug_hip_stream_t stream)
{
        if (!frame_dev || linesize == 0 || lines < 0 || frames < 0 || frames > 65535 || (frames > 1 && frame_stride < linesize * (size_t) lines)) {
                ug::set_last_error_msg("ug_hip_deinterlace_blend: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (lines < 5 || frames == 0) return UG_HIP_SUCCESS; // vc_deinterlace changes nothing below 5 lines (its loop runs for j < lines - 4)
        hipStream_t st = (hipStream_t) stream;
        uint8_t *base = (uint8_t *) frame_dev;
        const long full = (long) (linesize / 16) * 16; // the bytes of a line that lie in whole 16-byte columns
        const bool words = linesize % 4 == 0 && ((uintptr_t) frame_dev & 3) == 0 && frame_stride % 4 == 0;
        if (full > 0) {
                if (words) {
                        hipLaunchKernelGGL(deinterlace_kernel<uint32_t>, dim3((unsigned) ((full / 4 + 63) / 64), (unsigned) frames), dim3(64), 0, st, base, (long) linesize, lines, 0L,
                                           full / 4, frame_stride);
                } else {
                        hipLaunchKernelGGL(deinterlace_kernel<uint8_t>, dim3((unsigned) ((full + 63) / 64), (unsigned) frames), dim3(64), 0, st, base, (long) linesize, lines, 0L, full,
                                           frame_stride);
                }
        }
        if (full != (long) linesize) { // the column that reaches into the next line: after the others, as in the reference
                hipLaunchKernelGGL(deinterlace_kernel<uint8_t>, dim3(1, (unsigned) frames), dim3(64), 0, st, base, (long) linesize, lines, full, 16L, frame_stride);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

extern "C" int ug_hip_deinterlace_blend(void *frame_dev, size_t linesize, int lines, ug_hip_stream_t stream)
{
        return ug_hip_deinterlace_blend_batch(frame_dev, linesize, lines, 1, 0, stream);
}
