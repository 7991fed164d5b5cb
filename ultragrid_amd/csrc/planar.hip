// planar.hip -- planar <-> packed shuffles of the decode direction: what UltraGrid's software decoders
// hand back (planar 4:2:0 / 4:2:2, 8 or 10 bit) to the packed wire formats, and UYVY -> planar 4:2:2.
//
//   yuv420p_to_uyvy     src/from_planar.c:583-683 (8-bit planar 4:2:0 -> UYVY; both lines of a pair take the same chroma line,
//                       odd height: the last line stands alone, odd width: last word = Cb Y Cr 0);
//                       i420_8_to_uyvy (src/video_codec.c:1073-1094) is the same shuffle for tightly packed planes
//   yuv422p_to_uyvy     src/from_planar.c:391-423   (8-bit planar 4:2:2 -> UYVY, width/2 pairs per line)
//   yuv422p10le_to_v210 src/from_planar.c:296-333   (10-bit planar 4:2:2 in 16-bit words -> v210, width/6 groups per line)
//   uyvy_to_i422        src/video_codec.c:949-969   (UYVY -> planar 4:2:2, chroma (width+1)/2 wide)
//   uyvy_to_nv12        src/to_planar.c:207-302     (UYVY -> Y plane + interleaved CbCr plane, line pairs averaged)
//
// Pure byte movement, HBM-bound: every kernel reads each input byte once and writes each output byte once.  One lane moves
// 8 pixels (16 B of UYVY, 32 B of v210 per 12 px) with vector accesses when the geometry is aligned; ragged widths and odd
// pitches take the per-pair path.
#include "ug_common.h"

namespace {

struct Planes {
        const uint8_t *y, *cb, *cr;
        long y_pitch, cb_pitch, cr_pitch;
};

__device__ __forceinline__ uint32_t uyvy_word(uint32_t cb, uint32_t y0, uint32_t cr, uint32_t y1)
{
        return cb | y0 << 8 | cr << 16 | y1 << 24;
}

// 4 pairs: y8 = 8 luma bytes, cb4 / cr4 = 4 chroma bytes each -> 16 B of UYVY
__device__ __forceinline__ uint4 interleave8(uint2 y8, uint32_t cb4, uint32_t cr4)
{
        uint4 o;
        o.x = uyvy_word(cb4 & 0xff, y8.x & 0xff, cr4 & 0xff, (y8.x >> 8) & 0xff);
        o.y = uyvy_word((cb4 >> 8) & 0xff, (y8.x >> 16) & 0xff, (cr4 >> 8) & 0xff, y8.x >> 24);
        o.z = uyvy_word((cb4 >> 16) & 0xff, y8.y & 0xff, (cr4 >> 16) & 0xff, (y8.y >> 8) & 0xff);
        o.w = uyvy_word(cb4 >> 24, (y8.y >> 16) & 0xff, cr4 >> 24, y8.y >> 24);
        return o;
}

// V = vertical chroma subsampling (2: 4:2:0, 1: 4:2:2).  grid.y walks chroma lines (V == 2) or picture lines (V == 1).
template <int V, bool FAST>
__global__ __launch_bounds__(256) void planar_to_uyvy_kernel(Planes p, uint8_t *__restrict__ dst, long dst_pitch, int width, int height)
{
        const int cy = blockIdx.y * blockDim.y + threadIdx.y;
        const int y0 = V * cy;
        if (y0 >= height) return;
        const int y1 = V == 2 ? (y0 + 1 < height ? y0 + 1 : y0) : y0; // odd height: the last line stands alone
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        const uint8_t *cbl = p.cb + (long) cy * p.cb_pitch, *crl = p.cr + (long) cy * p.cr_pitch;
        if (FAST) { // 8 pixels per lane
                if (8 * i >= width) return;
                const uint32_t cb4 = *(const uint32_t *) (cbl + 4 * i), cr4 = *(const uint32_t *) (crl + 4 * i);
                ug::st_stream((uint4 *) (dst + (long) y0 * dst_pitch + 16 * i), interleave8(*(const uint2 *) (p.y + (long) y0 * p.y_pitch + 8 * i), cb4, cr4));
                if (V == 2 && y1 != y0) {
                        ug::st_stream((uint4 *) (dst + (long) y1 * dst_pitch + 16 * i), interleave8(*(const uint2 *) (p.y + (long) y1 * p.y_pitch + 8 * i), cb4, cr4));
                }
        } else { // one pixel pair per lane
                const int pairs = V == 2 ? (width + 1) / 2 : width / 2; // 4:2:2 source: width / 2 pairs, nothing for an odd tail
                if (i >= pairs) return;
                const uint32_t cb = cbl[i], cr = crl[i];
                const bool tail = 2 * i + 1 >= width; // odd width (4:2:0 only): Cb Y Cr 0
#pragma unroll
                for (int k = 0; k < V; k++) {
                        const int y = k ? y1 : y0;
                        if (k && y1 == y0) break;
                        const uint8_t *yl = p.y + (long) y * p.y_pitch;
                        ug::st_stream((uint32_t *) (dst + (long) y * dst_pitch + 4 * i), uyvy_word(cb, yl[2 * i], cr, tail ? 0 : yl[2 * i + 1]));
                }
        }
}

// one lane = one 6-pixel group: 6 Y + 3 Cb + 3 Cr 16-bit samples -> 4 words (from_planar.c:307-331: samples are OR-ed in as they
// are, no masking -- inputs are 10-bit by contract)
__global__ __launch_bounds__(256) void yuv422p10le_to_v210_kernel(Planes p, uint8_t *__restrict__ dst, long dst_pitch, int groups, int height)
{
        const int y = blockIdx.y * blockDim.y + threadIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
        if (y >= height || g >= groups) return;
        const uint16_t *sy = (const uint16_t *) (p.y + (long) y * p.y_pitch) + 6 * g;
        const uint16_t *scb = (const uint16_t *) (p.cb + (long) y * p.cb_pitch) + 3 * g;
        const uint16_t *scr = (const uint16_t *) (p.cr + (long) y * p.cr_pitch) + 3 * g;
        uint32_t Y[6], CB[3], CR[3];
        if (!(3 & (uintptr_t) sy)) { // 12 B of luma as three words (always true for even pitches: 12 g bytes)
                const uint32_t *w = (const uint32_t *) sy;
#pragma unroll
                for (int k = 0; k < 3; k++) { const uint32_t v = w[k]; Y[2 * k] = v & 0xffff; Y[2 * k + 1] = v >> 16; }
        } else {
#pragma unroll
                for (int k = 0; k < 6; k++) Y[k] = sy[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { CB[k] = scb[k]; CR[k] = scr[k]; }
        uint4 o;
        o.x = CB[0] | Y[0] << 10 | CR[0] << 20;
        o.y = Y[1] | CB[1] << 10 | Y[2] << 20;
        o.z = CR[1] | Y[3] << 10 | CB[2] << 20;
        o.w = Y[4] | CR[2] << 10 | Y[5] << 20;
        uint8_t *d = dst + (long) y * dst_pitch + 16 * g;
        if (!(15 & (uintptr_t) d)) {
                ug::st_stream((uint4 *) d, o);
        } else {
                uint32_t *dw = (uint32_t *) d;
                dw[0] = o.x; dw[1] = o.y; dw[2] = o.z; dw[3] = o.w;
        }
}

template <bool FAST>
__global__ __launch_bounds__(256) void uyvy_to_i422_kernel(const uint8_t *__restrict__ src, long src_pitch, uint8_t *__restrict__ py, long y_pitch,
                                                          uint8_t *__restrict__ pcb, long cb_pitch, uint8_t *__restrict__ pcr, long cr_pitch,
                                                          int width, int height)
{
        const int y = blockIdx.y * blockDim.y + threadIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
        if (y >= height) return;
        const uint8_t *s = src + (long) y * src_pitch;
        if (FAST) { // 8 pixels = 16 B per lane
                if (8 * i >= width) return;
                const uint4 q = *(const uint4 *) (s + 16 * i);
                const uint32_t w[4] = { q.x, q.y, q.z, q.w };
                uint32_t cb = 0, cr = 0;
                uint2 yy = make_uint2(0, 0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        cb |= (w[k] & 0xff) << (8 * k);
                        cr |= ((w[k] >> 16) & 0xff) << (8 * k);
                        const uint32_t two = ((w[k] >> 8) & 0xff) | (w[k] >> 24) << 8;
                        if (k < 2) yy.x |= two << (16 * k); else yy.y |= two << (16 * (k - 2));
                }
                ug::st_stream((uint2 *) (py + (long) y * y_pitch + 8 * i), yy);
                ug::st_stream((uint32_t *) (pcb + (long) y * cb_pitch + 4 * i), cb);
                ug::st_stream((uint32_t *) (pcr + (long) y * cr_pitch + 4 * i), cr);
        } else {
                if (i >= (width + 1) / 2) return;
                const uint32_t w = *(const uint32_t *) (s + 4 * i);
                pcb[(long) y * cb_pitch + i] = (uint8_t) w;
                pcr[(long) y * cr_pitch + i] = (uint8_t) (w >> 16);
                py[(long) y * y_pitch + 2 * i] = (uint8_t) (w >> 8);
                if (2 * i + 1 < width) py[(long) y * y_pitch + 2 * i + 1] = (uint8_t) (w >> 24);
        }
}


// uyvy_to_nv12 (to_planar.c:207-302) with the arithmetic of the reference's default build (-msse4.1, configure.ac:225): the first
// 16 * (width / 16) pixels of a line pair go through _mm_avg_epu8 = (a + b + 1) >> 1, the rest through the scalar tail
// (a + b) / 2 -- the result depends on the width, and this kernel reproduces that dependence (vec_px = 16 * (width / 16);
// vec_px = 0 gives the build without SSE3).  One lane = one pixel pair of a line pair; odd height: the last line pairs with itself.
__global__ __launch_bounds__(256) void uyvy_to_nv12_kernel(const uint8_t *__restrict__ src, long src_pitch, uint8_t *__restrict__ py, long y_pitch,
                                                          uint8_t *__restrict__ pc, long c_pitch, int width, int height, int vec_px)
{
        const int cy = blockIdx.y * blockDim.y + threadIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
        const int y0 = 2 * cy;
        if (y0 >= height || i >= (width + 1) / 2) return;
        const int y1 = y0 + 1 < height ? y0 + 1 : y0;
        const uint32_t a = *(const uint32_t *) (src + (long) y0 * src_pitch + 4 * i), b = *(const uint32_t *) (src + (long) y1 * src_pitch + 4 * i);
        const uint32_t rnd = 2 * i < vec_px ? 1 : 0;
        const uint32_t cb = ((a & 0xff) + (b & 0xff) + rnd) >> 1, cr = (((a >> 16) & 0xff) + ((b >> 16) & 0xff) + rnd) >> 1;
        *(uint16_t *) (pc + (long) cy * c_pitch + 2 * i) = (uint16_t) (cb | cr << 8);
        const bool second = 2 * i + 1 < width; // odd width: the last pair has one luma sample (to_planar.c:294-299)
        uint8_t *d0 = py + (long) y0 * y_pitch + 2 * i, *d1 = py + (long) y1 * y_pitch + 2 * i;
        d0[0] = (uint8_t) (a >> 8);
        if (second) d0[1] = (uint8_t) (a >> 24);
        if (y1 != y0) {
                d1[0] = (uint8_t) (b >> 8);
                if (second) d1[1] = (uint8_t) (b >> 24);
        }
}

// aligned fast path of the same: one lane = 8 pixels of a line pair (two 16 B loads, two 8 B luma stores, one 8 B CbCr store);
// the (a + b + rnd) >> 1 of four byte lanes at once: per-byte sums cannot carry into the neighbour because the even / odd bytes are
// averaged in separate 16-bit fields
__global__ __launch_bounds__(256) void uyvy_to_nv12_fast_kernel(const uint8_t *__restrict__ src, long src_pitch, uint8_t *__restrict__ py, long y_pitch,
                                                               uint8_t *__restrict__ pc, long c_pitch, int width, int height, int vec_px)
{
        const int cy = blockIdx.y * blockDim.y + threadIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
        const int y0 = 2 * cy;
        if (y0 >= height || 8 * i >= width) return;
        const int y1 = y0 + 1 < height ? y0 + 1 : y0;
        const uint4 a = *(const uint4 *) (src + (long) y0 * src_pitch + 16 * i), b = *(const uint4 *) (src + (long) y1 * src_pitch + 16 * i);
        const uint32_t wa[4] = { a.x, a.y, a.z, a.w }, wb[4] = { b.x, b.y, b.z, b.w };
        const uint32_t rnd = 8 * i < vec_px ? 0x00010001u : 0u; // vec_px is a multiple of 16: the 8 pixels are on one side of it
        uint32_t ya[2] = { 0, 0 }, yb[2] = { 0, 0 }, c[2] = { 0, 0 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
                const uint32_t ca = wa[k] & 0x00ff00ffu, cb = wb[k] & 0x00ff00ffu;        // Cb | Cr << 16
                const uint32_t avg = ((ca + cb + rnd) >> 1) & 0x00ff00ffu;
                c[k >> 1] |= ((avg & 0xff) | (avg >> 8)) << (16 * (k & 1));               // Cb, Cr bytes of this pair
                const uint32_t la = ((wa[k] >> 8) & 0xff) | ((wa[k] >> 24) << 8), lb = ((wb[k] >> 8) & 0xff) | ((wb[k] >> 24) << 8);
                ya[k >> 1] |= la << (16 * (k & 1));
                yb[k >> 1] |= lb << (16 * (k & 1));
        }
        ug::st_stream((uint2 *) (pc + (long) cy * c_pitch + 8 * i), make_uint2(c[0], c[1]));
        ug::st_stream((uint2 *) (py + (long) y0 * y_pitch + 8 * i), make_uint2(ya[0], ya[1]));
        if (y1 != y0) ug::st_stream((uint2 *) (py + (long) y1 * y_pitch + 8 * i), make_uint2(yb[0], yb[1]));
}

template <int V>
int launch_planar_to_uyvy(const Planes &p, void *dst, int dst_pitch, int width, int height, hipStream_t st)
{
        const int lines = V == 2 ? (height + 1) / 2 : height;
        const bool fast = width % 8 == 0 && !(dst_pitch & 15) && !(p.y_pitch & 7) && !(p.cb_pitch & 3) && !(p.cr_pitch & 3) &&
                          !(15 & (uintptr_t) dst) && !(7 & (uintptr_t) p.y) && !(3 & ((uintptr_t) p.cb | (uintptr_t) p.cr));
        const dim3 block(64, 4);
        if (fast) {
                const dim3 grid((unsigned) ((width / 8 + 63) / 64), (unsigned) ((lines + 3) / 4));
                hipLaunchKernelGGL((planar_to_uyvy_kernel<V, true>), grid, block, 0, st, p, (uint8_t *) dst, (long) dst_pitch, width, height);
        } else {
                const dim3 grid((unsigned) (((width + 1) / 2 + 63) / 64), (unsigned) ((lines + 3) / 4));
                hipLaunchKernelGGL((planar_to_uyvy_kernel<V, false>), grid, block, 0, st, p, (uint8_t *) dst, (long) dst_pitch, width, height);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

bool bad_planes(const void *y, const void *cb, const void *cr, const void *other, int w, int h)
{
        return !y || !cb || !cr || !other || w <= 0 || h <= 0 || (h + 3) / 4 > 65535;
}

} // namespace

extern "C" {

int ug_hip_yuv420p_to_uyvy(const void *y, int y_pitch, const void *cb, int cb_pitch, const void *cr, int cr_pitch, void *dst, int dst_pitch,
                           int width, int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_yuv420p_to_uyvy");
        if (bad_planes(y, cb, cr, dst, width, height) || (3 & (uintptr_t) dst)) {
                ug::set_last_error_msg("ug_hip_yuv420p_to_uyvy: bad arguments");
                return UG_HIP_EINVAL;
        }
        const int cw = (width + 1) / 2;
        if (!y_pitch) y_pitch = width;
        if (!cb_pitch) cb_pitch = cw;
        if (!cr_pitch) cr_pitch = cw;
        if (!dst_pitch) dst_pitch = ug::linesize(UG_PF_UYVY, width);
        if (!ug::planes_ok(height, { y_pitch, cb_pitch, cr_pitch, dst_pitch })) return ug::refuse_size("ug_hip_yuv420p_to_uyvy");
        if (dst_pitch & 3) {
                ug::set_last_error_msg("ug_hip_yuv420p_to_uyvy: destination pitch must be a multiple of 4");
                return UG_HIP_EINVAL;
        }
        const Planes p = { (const uint8_t *) y, (const uint8_t *) cb, (const uint8_t *) cr, y_pitch, cb_pitch, cr_pitch };
        return launch_planar_to_uyvy<2>(p, dst, dst_pitch, width, height, (hipStream_t) stream);
}

int ug_hip_yuv422p_to_uyvy(const void *y, int y_pitch, const void *cb, int cb_pitch, const void *cr, int cr_pitch, void *dst, int dst_pitch,
                           int width, int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_yuv422p_to_uyvy");
        if (bad_planes(y, cb, cr, dst, width, height) || (3 & (uintptr_t) dst)) {
                ug::set_last_error_msg("ug_hip_yuv422p_to_uyvy: bad arguments");
                return UG_HIP_EINVAL;
        }
        const int cw = (width + 1) / 2;
        if (!y_pitch) y_pitch = width;
        if (!cb_pitch) cb_pitch = cw;
        if (!cr_pitch) cr_pitch = cw;
        if (!dst_pitch) dst_pitch = ug::linesize(UG_PF_UYVY, width);
        if (!ug::planes_ok(height, { y_pitch, cb_pitch, cr_pitch, dst_pitch })) return ug::refuse_size("ug_hip_yuv422p_to_uyvy");
        if (dst_pitch & 3) {
                ug::set_last_error_msg("ug_hip_yuv422p_to_uyvy: destination pitch must be a multiple of 4");
                return UG_HIP_EINVAL;
        }
        const Planes p = { (const uint8_t *) y, (const uint8_t *) cb, (const uint8_t *) cr, y_pitch, cb_pitch, cr_pitch };
        return launch_planar_to_uyvy<1>(p, dst, dst_pitch, width, height, (hipStream_t) stream);
}

int ug_hip_yuv422p10le_to_v210(const void *y, int y_pitch, const void *cb, int cb_pitch, const void *cr, int cr_pitch, void *dst, int dst_pitch,
                               int width, int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_yuv422p10le_to_v210");
        if (bad_planes(y, cb, cr, dst, width, height) || (3 & (uintptr_t) dst) || (1 & ((uintptr_t) y | (uintptr_t) cb | (uintptr_t) cr))) {
                ug::set_last_error_msg("ug_hip_yuv422p10le_to_v210: bad arguments");
                return UG_HIP_EINVAL;
        }
        const int cw = (width + 1) / 2;
        if (!y_pitch) y_pitch = 2 * width;
        if (!cb_pitch) cb_pitch = 2 * cw;
        if (!cr_pitch) cr_pitch = 2 * cw;
        if (!dst_pitch) dst_pitch = ug::linesize(UG_PF_V210, width);
        if (!ug::planes_ok(height, { y_pitch, cb_pitch, cr_pitch, dst_pitch })) return ug::refuse_size("ug_hip_yuv422p10le_to_v210");
        if ((dst_pitch & 3) || ((y_pitch | cb_pitch | cr_pitch) & 1)) {
                ug::set_last_error_msg("ug_hip_yuv422p10le_to_v210: pitches must keep 16-bit samples / 32-bit words aligned");
                return UG_HIP_EINVAL;
        }
        const int groups = width / 6; // from_planar.c:307: the width % 6 tail is not written
        if (groups == 0) return UG_HIP_SUCCESS;
        const Planes p = { (const uint8_t *) y, (const uint8_t *) cb, (const uint8_t *) cr, y_pitch, cb_pitch, cr_pitch };
        const dim3 block(64, 4), grid((unsigned) ((groups + 63) / 64), (unsigned) ((height + 3) / 4));
        hipLaunchKernelGGL(yuv422p10le_to_v210_kernel, grid, block, 0, (hipStream_t) stream, p, (uint8_t *) dst, (long) dst_pitch, groups, height);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int ug_hip_uyvy_to_i422(const void *src, int src_pitch, void *y, int y_pitch, void *cb, int cb_pitch, void *cr, int cr_pitch, int width,
                        int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_uyvy_to_i422");
        if (bad_planes(y, cb, cr, src, width, height) || (3 & (uintptr_t) src)) {
                ug::set_last_error_msg("ug_hip_uyvy_to_i422: bad arguments");
                return UG_HIP_EINVAL;
        }
        const int cw = (width + 1) / 2;
        if (!src_pitch) src_pitch = ug::linesize(UG_PF_UYVY, width);
        if (!y_pitch) y_pitch = width;
        if (!cb_pitch) cb_pitch = cw;
        if (!cr_pitch) cr_pitch = cw;
        if (!ug::planes_ok(height, { src_pitch, y_pitch, cb_pitch, cr_pitch })) return ug::refuse_size("ug_hip_uyvy_to_i422");
        if (src_pitch & 3) {
                ug::set_last_error_msg("ug_hip_uyvy_to_i422: source pitch must be a multiple of 4");
                return UG_HIP_EINVAL;
        }
        const bool fast = width % 8 == 0 && !(src_pitch & 15) && !(y_pitch & 7) && !(cb_pitch & 3) && !(cr_pitch & 3) && !(15 & (uintptr_t) src) &&
                          !(7 & (uintptr_t) y) && !(3 & ((uintptr_t) cb | (uintptr_t) cr));
        const dim3 block(64, 4);
        if (fast) {
                const dim3 grid((unsigned) ((width / 8 + 63) / 64), (unsigned) ((height + 3) / 4));
                hipLaunchKernelGGL((uyvy_to_i422_kernel<true>), grid, block, 0, (hipStream_t) stream, (const uint8_t *) src, (long) src_pitch,
                                   (uint8_t *) y, (long) y_pitch, (uint8_t *) cb, (long) cb_pitch, (uint8_t *) cr, (long) cr_pitch, width, height);
        } else {
                const dim3 grid((unsigned) ((cw + 63) / 64), (unsigned) ((height + 3) / 4));
                hipLaunchKernelGGL((uyvy_to_i422_kernel<false>), grid, block, 0, (hipStream_t) stream, (const uint8_t *) src, (long) src_pitch,
                                   (uint8_t *) y, (long) y_pitch, (uint8_t *) cb, (long) cb_pitch, (uint8_t *) cr, (long) cr_pitch, width, height);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int ug_hip_uyvy_to_nv12(const void *src, int src_pitch, void *y, int y_pitch, void *cbcr, int cbcr_pitch, int width, int height,
                        ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_uyvy_to_nv12");
        if (!src || !y || !cbcr || width <= 0 || height <= 0 || (height + 7) / 8 > 65535 || (3 & (uintptr_t) src) || (1 & (uintptr_t) cbcr)) {
                ug::set_last_error_msg("ug_hip_uyvy_to_nv12: bad arguments");
                return UG_HIP_EINVAL;
        }
        const int cw = (width + 1) / 2;
        if (!src_pitch) src_pitch = ug::linesize(UG_PF_UYVY, width);
        if (!y_pitch) y_pitch = width;
        if (!cbcr_pitch) cbcr_pitch = 2 * cw;
        if (!ug::planes_ok(height, { src_pitch, y_pitch, cbcr_pitch })) return ug::refuse_size("ug_hip_uyvy_to_nv12");
        if ((src_pitch & 3) || (cbcr_pitch & 1)) {
                ug::set_last_error_msg("ug_hip_uyvy_to_nv12: source pitch must be a multiple of 4, CbCr pitch of 2");
                return UG_HIP_EINVAL;
        }
        const dim3 block(64, 4);
        if (width % 8 == 0 && !(src_pitch & 15) && !(y_pitch & 7) && !(cbcr_pitch & 7) && !(15 & (uintptr_t) src) && !(7 & ((uintptr_t) y | (uintptr_t) cbcr))) {
                const dim3 grid((unsigned) ((width / 8 + 63) / 64), (unsigned) (((height + 1) / 2 + 3) / 4));
                hipLaunchKernelGGL(uyvy_to_nv12_fast_kernel, grid, block, 0, (hipStream_t) stream, (const uint8_t *) src, (long) src_pitch, (uint8_t *) y,
                                   (long) y_pitch, (uint8_t *) cbcr, (long) cbcr_pitch, width, height, 16 * (width / 16));
                UG_HIP_LAUNCH_CHECK();
                return UG_HIP_SUCCESS;
        }
        const dim3 grid((unsigned) ((cw + 63) / 64), (unsigned) (((height + 1) / 2 + 3) / 4));
        hipLaunchKernelGGL(uyvy_to_nv12_kernel, grid, block, 0, (hipStream_t) stream, (const uint8_t *) src, (long) src_pitch, (uint8_t *) y,
                           (long) y_pitch, (uint8_t *) cbcr, (long) cbcr_pitch, width, height, 16 * (width / 16));
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

} // extern "C"
