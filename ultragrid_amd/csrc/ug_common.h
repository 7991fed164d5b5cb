// ug_common.h -- internal helpers shared by the HIP translation units of libug_mi355x.so
#pragma once

#include <hip/hip_runtime.h>
#include <initializer_list>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ug_mi355x.h"

namespace ug {

// thread-local copy of the last HIP error text (cuda_wrapper.cu:115-118 keeps the same state)
void set_last_error(hipError_t e, const char *what);
void set_last_error_msg(const char *msg);

#define UG_HIP_TRY(expr)                                                     \
        do {                                                                 \
                hipError_t e_ = (expr);                                      \
                if (e_ != hipSuccess) {                                      \
                        ug::set_last_error(e_, #expr);                       \
                        return UG_HIP_ERUNTIME;                              \
                }                                                            \
        } while (0)

// check the asynchronous launch error right after a <<<>>> launch
#define UG_HIP_LAUNCH_CHECK()                                                \
        do {                                                                 \
                hipError_t e_ = hipGetLastError();                           \
                if (e_ != hipSuccess) {                                      \
                        ug::set_last_error(e_, "kernel launch");             \
                        return UG_HIP_ERUNTIME;                              \
                }                                                            \
        } while (0)

// ---- argument ranges of the C ABI ----
// The header promises "0 or a negative UG_HIP_E* code" from every entry point, so a size that is not a picture, or whose byte counts leave the
// range the kernels index with, is refused before anything is launched (the reference: cuda_dxt.cu:745-746 returns -1 for a bad size; its
// vc_get_linesize is plain int arithmetic and wraps).  Bound: width and |height| up to 65 536 -- UltraGrid's largest mode is 8K -- and every
// frame or plane up to INT_MAX bytes, which is what the kernels' 32-bit byte offsets inside one frame cover.  (Batches multiply by size_t strides.)
constexpr int kMaxDim = 65536;
constexpr long long kMaxFrameBytes = 0x7fffffffLL;
static inline bool dims_ok(int width, int height) { return width > 0 && width <= kMaxDim && height > 0 && height <= kMaxDim; }
/// height < 0 = bottom-up where an entry point allows it (cuda_dxt.h:36-39); INT_MIN has no absolute value
static inline bool dims_ok_signed(int width, int height) { return height != INT32_MIN && dims_ok(width, height < 0 ? -height : height); }
/// rows lines of pitch bytes: one frame / plane the kernels may index with 32 bits
static inline bool span_ok(long long pitch, long long rows) { return pitch >= 0 && rows >= 0 && (rows == 0 || pitch <= kMaxFrameBytes / rows); }
/// every plane of a picture: no negative pitch, rows * pitch within the bound
static inline bool planes_ok(int rows, std::initializer_list<long long> pitches)
{
        for (long long p : pitches) if (!span_ok(p, rows)) return false;
        return true;
}
static inline int refuse_size(const char *who)
{
        char msg[160];
        snprintf(msg, sizeof msg, "%s: size out of range (width, |height| 1..%d, at most %lld bytes per frame or plane)", who, kMaxDim, kMaxFrameBytes);
        set_last_error_msg(msg);
        return UG_HIP_EINVAL;
}

static inline int linesize(ug_pixfmt_t f, int width)
{
        if (width <= 0 || width > kMaxDim) return 0; // (callers refuse 0; ug_hip_linesize turns it into UG_HIP_EINVAL)
        // video_codec.c:120-206 (block bytes / pixels, h_align) and :507-521
        int bb, bp, ha;
        switch (f) {
        case UG_PF_RGBA: bb = 4; bp = 1; ha = 1; break;
        case UG_PF_UYVY:
        case UG_PF_UYVY_RAW:
        case UG_PF_YUYV: bb = 4; bp = 2; ha = 2; break;
        case UG_PF_RGB:
        case UG_PF_BGR:
        case UG_PF_YUV444: bb = 3; bp = 1; ha = 1; break;
        case UG_PF_DVS10:
        case UG_PF_V210: bb = 16; bp = 6; ha = 48; break;
        case UG_PF_R10K: bb = 4; bp = 1; ha = 64; break;
        case UG_PF_R12L: bb = 36; bp = 8; ha = 8; break;
        case UG_PF_Y216: bb = 8; bp = 2; ha = 2; break;
        case UG_PF_Y416: bb = 8; bp = 1; ha = 1; break;
        case UG_PF_VUYA: bb = 4; bp = 1; ha = 1; break;
        case UG_PF_RG48: bb = 6; bp = 1; ha = 1; break;
        default: return 0;
        }
        width = (width + ha - 1) / ha * ha;
        return (width + bp - 1) / bp * bb;
}

// jpeg_fdct.hip: FDCT+quantise of one 8-bit component, planar (xstride 1) or packed (xstride = bytes per pixel)
int jpeg_fdct_quant_strided(const void *plane, int pitch, int xstride, int width, int height, int blocks_w, int blocks_h,
                            const float *div, int16_t *out, float *coef, ug_hip_stream_t stream);

// the JPEG encoder's colour stage (jpeg_fdct.hip): RGB (3 B/px) or UYVY frames from colour space cs_in to cs_out (UG_JPEG_CS_*), grid.z = frame
int jpeg_colour_convert(ug_pixfmt_t fmt, int cs_in, int cs_out, const void *src, int src_pitch, void *dst, int dst_pitch, int width, int height, int frames,
                        size_t src_stride, size_t dst_stride, ug_hip_stream_t stream);
// UYVY frames -> packed 3 B/px for a 4:4:4 encoder (pair's chroma for both pixels), as they are (cs_out UG_JPEG_CS_ASIS / _BT709) or mapped BT.709 -> cs_out
int jpeg_uyvy_to_444(int cs_out, const void *src, int src_pitch, void *dst, int dst_pitch, int width, int height, int frames, size_t src_stride, size_t dst_stride,
                     ug_hip_stream_t stream);
int jpeg_fdct_quant_rgb444(const void *src, int pitch, int width, int height, int blocks_w, int blocks_h, const float *div,
                           int16_t *out_r, int16_t *out_g, int16_t *out_b, ug_hip_stream_t stream);

// get_color_coeffs(cs, depth) (color_space.c:149-184) as constants: [cs - 1][slot], cs 1 = BT.601, 2 = BT.709; slot 0 = full range, then
// 8, 10, 12, 16 bit limited range; 14 ints in the field order of struct color_coeffs.  Defined in lavc_conv.hip, checked against the compiled
// reference through ug_hip_color_coeffs (tests/test_lavc_conv.py).
extern const int kColorCoeffs[2][5][14];
static inline int color_depth_slot(int depth) { return depth == 0 ? 0 : depth == 8 ? 1 : depth == 10 ? 2 : depth == 12 ? 3 : depth == 16 ? 4 : -1; }

// pixfmt_ext.hip: the pairs of decoders[] outside pixfmt.hip's core
int pixfmt_ext_supported(ug_pixfmt_t in, ug_pixfmt_t out);
int pixfmt_ext_convert(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height, int src_pitch, int dst_pitch, int dst_len,
                       int rshift, int gshift, int bshift, hipStream_t st);

} // namespace ug
#ifdef __HIPCC__
namespace ug {
// Streaming ("non-temporal") stores.  Every kernel of this library writes its output once and never reads it back; without the `nt` hint
// a store allocates its line in L2 and is written back when the line is evicted.  Measured (round 3, interleaved A/B on one box,
// profiles/r03_nt_stores_ab.txt): DXT5-YCoCg -> RGBA, one 4K frame per launch 12.8 -> 9.15 us, 8 frames per launch 8.8 -> 8.05 us per frame.
// -DUG_NO_STREAM_STORES builds the plain-store library for A/B runs.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#ifndef UG_NO_STREAM_STORES
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v) { __builtin_nontemporal_store(u32x4_t{ v.x, v.y, v.z, v.w }, (u32x4_t *) p); }
__device__ __forceinline__ void st_stream(uint2 *p, const uint2 &v) { __builtin_nontemporal_store(u32x2_t{ v.x, v.y }, (u32x2_t *) p); }
__device__ __forceinline__ void st_stream(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(uint16_t *p, uint16_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(uint8_t *p, uint8_t v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v) { *p = v; }
__device__ __forceinline__ void st_stream(uint2 *p, const uint2 &v) { *p = v; }
__device__ __forceinline__ void st_stream(uint32_t *p, uint32_t v) { *p = v; }
__device__ __forceinline__ void st_stream(uint16_t *p, uint16_t v) { *p = v; }
__device__ __forceinline__ void st_stream(uint8_t *p, uint8_t v) { *p = v; }
#endif

// Streaming loads were A/B-measured too (profiles/r03_nt_loads_ab.txt): a loss for the one-word-per-lane kernels (UYVY->RGB 9.5 -> 10.6 us,
// DXT5->RGBA 8.7 -> 10.0 us per 4K frame) and a gain of 4-5 % where a wave fetches its 64 multi-word units as one contiguous region
// through LDS (RG48->RGB 19.2 -> 18.4 us, Y416->UYVY 14.7 -> 13.9): used there only.
__device__ __forceinline__ uint4 ld_stream(const uint4 *p) { const u32x4_t v = __builtin_nontemporal_load((const u32x4_t *) p); return make_uint4(v.x, v.y, v.z, v.w); }

// 128-bit unit I/O of the "K iterations per lane" converter kernels (pixfmt.hip, pixfmt_ext.hip).  A lane's unit is BYTES contiguous
// bytes.  When that is one 16-byte word the lanes of a wave access consecutive words and nothing else is needed.  When it is
// several, per-lane accesses would be strided (every load instruction touching 64 different cache lines and using 16 bytes of each --
// measured: RG48->RGB fell from 0.31 to 0.19 of 8 TB/s that way), so the wave moves its 64 units as ONE contiguous region: word c of
// the region is handled by lane c % 64, and the words change hands through LDS (rows of an odd number of 16-byte words:
// conflict-free on both sides).
template <int BYTES>
struct UnitIO {
        static constexpr int V = BYTES / 16;                    // 16-byte words per unit
        static constexpr int ROW = V == 1 ? 1 : (V | 1);        // LDS row stride in words, odd
        static constexpr int LDS_WORDS = V == 1 ? 0 : 64 * ROW; // per wave
        // region = the wave's first unit in global memory; units = how many of the wave's 64 units exist
        static __device__ __forceinline__ void load(const uint4 *region, uint8_t *priv, uint4 *lds, int lane, int units)
        {
                if (V == 1) {
                        if (lane < units) *(uint4 *) priv = region[lane];
                        return;
                }
#pragma unroll
                for (int i = 0; i < V; i++) {
                        const int c = i * 64 + lane;
                        if (c < units * V) lds[(c / V) * ROW + c % V] = ld_stream(region + c);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < V; i++) ((uint4 *) priv)[i] = lds[lane * ROW + i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
        }
        static __device__ __forceinline__ void store(uint4 *region, const uint8_t *priv, uint4 *lds, int lane, int units)
        {
                if (V == 1) {
                        if (lane < units) st_stream(region + lane, *(const uint4 *) priv);
                        return;
                }
#pragma unroll
                for (int i = 0; i < V; i++) lds[lane * ROW + i] = ((const uint4 *) priv)[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < V; i++) {
                        const int c = i * 64 + lane;
                        if (c < units * V) st_stream(region + c, lds[(c / V) * ROW + c % V]);
                }
        }
};

template <int N> struct WordVec;
template <> struct WordVec<1> { typedef uint32_t type; template <int W> static __device__ __forceinline__ uint32_t make(const uint32_t (&w)[W], int i) { return w[i]; } };
template <> struct WordVec<2> { typedef uint2 type; template <int W> static __device__ __forceinline__ uint2 make(const uint32_t (&w)[W], int i) { return make_uint2(w[i], w[i + 1]); } };
template <> struct WordVec<4> { typedef uint4 type; template <int W> static __device__ __forceinline__ uint4 make(const uint32_t (&w)[W], int i) { return make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]); } };
// The same for a unit of W 32-bit words that a lane holds in registers (24-, 36-, 12-byte units: W = 6, 9, 3 ...): stored lane by lane
// the unit's words go out as W (or W / 2) strided instructions, each covering a fraction of every line it touches -- through L2 that
// merges, streamed past it the pieces reach HBM one by one (rocprofv3 WRITE_SIZE 1.2-1.56 x the output for the 3-byte-pixel rows,
// profiles/r03_write_by_row.txt).  Here the wave's 64 units leave as ONE contiguous region: memory word c of the region (16, 8 or 4 bytes,
// whatever divides the unit) is stored by lane c % 64, so every store instruction covers whole lines; the words change hands through
// LDS rows of an odd number of memory words.
template <int W>
struct WaveWords {
        static constexpr int G = W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1); // 32-bit words per memory word
        static constexpr int V = W / G;                                  // memory words per unit
        static constexpr int ROW = V == 1 ? 1 : (V | 1);
        static constexpr int LDS_DWORDS = V == 1 ? 0 : 64 * ROW * G;     // per wave
        // region = where the wave's first unit goes; units = how many of the wave's 64 units exist (lanes >= units hold nothing);
        // live = the lanes that execute this call: 64 when the lanes past the end stay (idle) in the wave, `units` when they have returned
        static __device__ __forceinline__ void store(uint8_t *region, const uint32_t (&w)[W], uint32_t *lds, int lane, int units, int live = 64)
        {
                using T = typename WordVec<G>::type;
                if (V == 1) {
                        if (lane < units) st_stream((T *) region + lane, WordVec<G>::make(w, 0));
                        return;
                }
                T *const l = (T *) lds;
#pragma unroll
                for (int i = 0; i < V; i++) l[lane * ROW + i] = WordVec<G>::make(w, i * G);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < V; i++) {
                        const int c = i * live + lane;
                        if (c < units * V) st_stream((T *) region + c, l[(c / V) * ROW + c % V]);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // the next use of the rows (a loop around this call) must not overtake the reads
                __builtin_amdgcn_wave_barrier();
        }
};

} // namespace ug
#endif

