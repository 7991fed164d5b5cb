// pixfmt.hip -- whole-frame pixel-format conversion on the device, replacing the per-line CPU
// decoder_t loop (pixfmt_conv.h:87-88, driver loops cuda_dxt.cpp:213-218, gpujpeg.cpp:597-604,
// testcard_common.c:121-129) and the packed->planar whole-buffer converters (to_planar.h).
//
// All arithmetic here is integer (Q14 fixed point, color_space.h:96-109) or byte shuffling;
// results are bit-identical to the reference's C (pinned by oracle/_ref/libugref*.so).
//
// Two code paths per conversion:
//   * generic: one lane per minimal unit of the reference loop, byte-granular accesses, exact
//     line-tail semantics (ragged widths, partial v210 groups, odd sizes);
//   * fast (hot formats, aligned geometry): one lane per 16..32 input bytes, 128-bit loads,
//     wave accesses contiguous -- these are pure HBM-bandwidth kernels.
#include "ug_common.h"

namespace {

// Q14 coefficients, BT.709 limited range (the default, color_space.c:149-191).  Values are the
// compile-time table of the reference, reproduced by oracle/pixfmt_oracle.c:oracle_color_coeffs
// and pinned against get_color_coeffs() in tests/test_oracle_pixfmt.py.
struct Cfs {
        int y_r, y_g, y_b, cb_r, cb_g, cb_b, cr_r, cr_g, cr_b, y_scale, r_cr, g_cb, g_cr, b_cb;
};
__device__ constexpr Cfs kCfs8  = { 2992, 10063, 1016, -1649, -5547, 7196, 7195, -6536, -659, 19077, 29371, -3494, -8733, 34610 };
__device__ constexpr Cfs kCfs10 = { 2983, 10034, 1013, -1644, -5531, 7175, 7174, -6517, -657, 19133, 29457, -3504, -8758, 34712 };
constexpr int kBase = 14; // COMP_BASE, color_space.h:70-71

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p)
{
        return (uint32_t) p[0] | (uint32_t) p[1] << 8 | (uint32_t) p[2] << 16 | (uint32_t) p[3] << 24;
}
__device__ __forceinline__ void st32(uint8_t *p, uint32_t v)
{
        p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24;
}
__device__ __forceinline__ uint32_t alpha_mask(int rs, int gs, int bs)
{
        return 0xFFFFFFFFu ^ (0xFFu << rs) ^ (0xFFu << gs) ^ (0xFFu << bs);
}

struct Args {
        const uint8_t *src;
        uint8_t *dst;
        int width, height, spitch, dpitch, dst_len, rs, gs, bs;
        int src_line; // bytes of a source line (vc_get_linesize of the input codec)
};

// ------------------------------ generic per-unit converters ------------------------------
// Each struct: units(dst_len) = iterations of the reference loop for one line; run() = one iteration.

struct V210toUYVY { // pixfmt_conv.c:86-130
        static __device__ __host__ int units(int dl) { return (dl / 4 + 2) / 3; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                const int words = min(3, a.dst_len / 4 - 3 * k);
                s += 16 * k; d += 12 * k;
                const uint32_t w0 = ld32(s), w1 = ld32(s + 4);
#define S(w, f) ((((w) >> (10 * (f))) & 0x3ffu) >> 2)
                st32(d, S(w0, 0) | S(w0, 1) << 8 | S(w0, 2) << 16 | S(w1, 0) << 24);
                if (words >= 2) {
                        const uint32_t w2 = ld32(s + 8);
                        st32(d + 4, S(w1, 1) | S(w1, 2) << 8 | S(w2, 0) << 16 | S(w2, 1) << 24);
                        if (words >= 3) {
                                const uint32_t w3 = ld32(s + 12);
                                st32(d + 8, S(w2, 2) | S(w3, 0) << 8 | S(w3, 1) << 16 | S(w3, 2) << 24);
                        }
                }
#undef S
        }
};
struct SwapYUYV { // pixfmt_conv.c:136-198
        static __device__ __host__ int units(int dl) { return dl / 4; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &)
        {
                s += 4 * k; d += 4 * k;
                const uint8_t a = s[0], b = s[1], c = s[2], e = s[3];
                d[0] = b; d[1] = a; d[2] = e; d[3] = c;
        }
};
// hipcc (ROCm 7.2) fuses "clamp(x >> 14, 0, 255) | clamp(y >> 14, 0, 255) << 8" into gfx950's
// v_ashr_pk_u8_i32 and then ORs further bytes into the result assuming its upper 16 bits are zero;
// on MI355X the instruction leaves the destination's upper half unchanged, so stale bytes leak into
// the packed word (caught by tests/test_gpu_pixfmt.py).  Making the clamped value opaque keeps the
// clamp (v_med3_i32) and the byte packing as separate, correct instructions at zero run-time cost.
__device__ __forceinline__ int opaque(int v)
{
        asm volatile("" : "+v"(v));
        return v;
}
__device__ __forceinline__ void yuv_to_rgb8(int y, int u, int v, uint8_t *o)
{
        // copylineYUVtoRGB, pixfmt_conv.c:1065-1094: clamp [0,255]
        o[0] = opaque(clampi((y + v * kCfs8.r_cr) >> kBase, 0, 255));
        o[1] = opaque(clampi((y + u * kCfs8.g_cb + v * kCfs8.g_cr) >> kBase, 0, 255));
        o[2] = opaque(clampi((y + u * kCfs8.b_cb) >> kBase, 0, 255));
}
struct UYVYtoRGB { // pixfmt_conv.c:1102-1108
        static __device__ __host__ int units(int dl) { return dl / 6; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &)
        {
                s += 4 * k; d += 6 * k;
                const int u = s[0] - 128, v = s[2] - 128;
                yuv_to_rgb8(kCfs8.y_scale * (s[1] - 16), u, v, d);
                yuv_to_rgb8(kCfs8.y_scale * (s[3] - 16), u, v, d + 3);
        }
};
struct UYVYtoRGBA { // pixfmt_conv.c:1137-1163 (fp64, truncation toward zero)
        static __device__ __host__ int units(int dl) { return dl / 8; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                s += 4 * k; d += 8 * k;
                const int u = s[0], y1 = s[1], v = s[2], y2 = s[3];
                const uint32_t am = alpha_mask(a.rs, a.gs, a.bs);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                        const int y = i ? y2 : y1;
                        // no contraction (-ffp-contract=off): each product and sum rounds separately, as on the CPU
                        int r = 1.164 * (y - 16) + 1.793 * (v - 128);
                        int g = 1.164 * (y - 16) - 0.534 * (v - 128) - 0.213 * (u - 128);
                        int b = 1.164 * (y - 16) + 2.115 * (u - 128);
                        r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                        st32(d + 4 * i, am | (uint32_t) r << a.rs | (uint32_t) g << a.gs | (uint32_t) b << a.bs);
                }
        }
};
template <int RO, int GO, int BO, int PS>
struct ToUYVY { // vc_copylineToUYVY, pixfmt_conv.c:1008-1053
        static __device__ __host__ int units(int dl) { return (dl + 3) / 4; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &)
        {
                s += 2 * PS * k; d += 4 * k;
                int r = s[RO], g = s[GO], b = s[BO];
                const int y1 = ((r * kCfs8.y_r + g * kCfs8.y_g + b * kCfs8.y_b) >> kBase) + 16;
                int u = r * kCfs8.cb_r + g * kCfs8.cb_g + b * kCfs8.cb_b;
                int v = r * kCfs8.cr_r + g * kCfs8.cr_g + b * kCfs8.cr_b;
                s += PS;
                r = s[RO]; g = s[GO]; b = s[BO];
                const int y2 = ((r * kCfs8.y_r + g * kCfs8.y_g + b * kCfs8.y_b) >> kBase) + 16;
                u += r * kCfs8.cb_r + g * kCfs8.cb_g + b * kCfs8.cb_b;
                v += r * kCfs8.cr_r + g * kCfs8.cr_g + b * kCfs8.cr_b;
                u = ((u / 2) >> kBase) + 128; // C '/' truncates toward zero, '>>' floors
                v = ((v / 2) >> kBase) + 128;
                st32(d, ((uint32_t) (y2 & 0xFF) << 24) | ((v & 0xFF) << 16) | ((y1 & 0xFF) << 8) | (u & 0xFF));
        }
};
template <bool OUT16>
struct V210toRGB { // pixfmt_conv.c:2884-2940 / :2942-3002
        static constexpr int kObl = OUT16 ? 36 : 18;
        static __device__ __host__ int units(int dl) { return (dl + kObl - 1) / kObl; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                constexpr int idepth = OUT16 ? 10 : 8;
                const Cfs c = OUT16 ? kCfs10 : kCfs8;
                constexpr int y_shift = 1 << (idepth - 4), c_shift = 1 << (idepth - 1);
                constexpr int sh = OUT16 ? kBase - 6 : kBase;
                constexpr int lo = OUT16 ? 1 << 8 : 1, hi = OUT16 ? (255 << 8) - 1 : 254; // CLAMP_FULL
                constexpr int drop = OUT16 ? 0 : 2;
                s += 16 * k;
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) w[i] = ld32(s + 4 * i);
                uint8_t o[kObl];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                        // samples in UYVY order, 3 per word
                        const int ys = 2 * i + 1, us = 4 * (i / 2), vs = us + 2;
                        const int Y = (int) ((w[ys / 3] >> (10 * (ys % 3))) & 0x3ffu) >> drop;
                        const int u = ((int) ((w[us / 3] >> (10 * (us % 3))) & 0x3ffu) >> drop) - c_shift;
                        const int v = ((int) ((w[vs / 3] >> (10 * (vs % 3))) & 0x3ffu) >> drop) - c_shift;
                        const int y = c.y_scale * (Y - y_shift);
                        const int r = clampi((y + v * c.r_cr) >> sh, lo, hi);
                        const int g = clampi((y + u * c.g_cb + v * c.g_cr) >> sh, lo, hi);
                        const int b = clampi((y + u * c.b_cb) >> sh, lo, hi);
                        if (OUT16) {
                                o[6 * i + 0] = r; o[6 * i + 1] = r >> 8;
                                o[6 * i + 2] = g; o[6 * i + 3] = g >> 8;
                                o[6 * i + 4] = b; o[6 * i + 5] = b >> 8;
                        } else {
                                o[3 * i + 0] = r; o[3 * i + 1] = g; o[3 * i + 2] = b;
                        }
                }
                // The reference writes the whole last block even past dst_len (the spill is
                // overwritten by the next line / lands in MAX_PADDING); we clip to the line.
                const int n = min(kObl, a.dst_len - kObl * k);
                d += kObl * k;
                if (n == kObl) { // static indices only: o[] stays in registers and the stores merge
#pragma unroll
                        for (int i = 0; i < kObl; i++) d[i] = o[i];
                } else {
#pragma unroll
                        for (int i = 0; i < kObl; i++) {
                                if (i < n) d[i] = o[i];
                        }
                }
        }
};
struct RGBAtoRGB { // pixfmt_conv.c:866-900, portable path (vc_copylineRGBAtoRGBwithShift)
        static __device__ __host__ int units(int dl) { return dl / 3; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &)
        {
                s += 4 * k; d += 3 * k;
                d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        }
};
struct RGBtoRGBA { // pixfmt_conv.c:944-990
        static __device__ __host__ int units(int dl) { return dl / 4; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                s += 3 * k; d += 4 * k;
                st32(d, alpha_mask(a.rs, a.gs, a.bs) | (uint32_t) s[0] << a.rs | (uint32_t) s[1] << a.gs | (uint32_t) s[2] << a.bs);
        }
};
struct RGBAshift { // vc_copylineRGBA, pixfmt_conv.c:538-589 (non-default shifts: alpha := 0xFF)
        static __device__ __host__ int units(int dl) { return dl / 4; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                s += 4 * k; d += 4 * k;
                st32(d, alpha_mask(a.rs, a.gs, a.bs) | (uint32_t) s[0] << a.rs | (uint32_t) s[1] << a.gs | (uint32_t) s[2] << a.bs);
        }
};
struct RGBshift { // vc_copylineRGB, pixfmt_conv.c:732-753
        static __device__ __host__ int units(int dl) { return dl / 3; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &a)
        {
                s += 3 * k; d += 3 * k;
                const uint32_t o = (uint32_t) s[0] << a.rs | (uint32_t) s[1] << a.gs | (uint32_t) s[2] << a.bs;
                d[0] = o; d[1] = o >> 8; d[2] = o >> 16;
        }
};
struct UYVYtoV210 { // pixfmt_conv.c:2581-2607: consecutive BYTES -> 10-bit fields (<<2), 3 per word
        static __device__ __host__ int units(int dl) { return dl / 4; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &)
        {
                s += 3 * k; d += 4 * k;
                st32(d, ((uint32_t) s[0] << 2) | ((uint32_t) s[1] << 2) << 10 | ((uint32_t) s[2] << 2) << 20);
        }
};
struct Copy { // vc_memcpy, pixfmt_conv.c:2529-2536
        static __device__ __host__ int units(int dl) { return dl; }
        static __device__ void run(uint8_t *d, const uint8_t *s, int k, const Args &) { d[k] = s[k]; }
};

template <class CONV>
__global__ __launch_bounds__(256) void generic_kernel(Args a, int upl, int k0, long total)
{
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= total) return;
        const int line = (int) (idx / upl), k = k0 + (int) (idx - (long) line * upl); // upl = units of a line from k0 on
        CONV::run(a.dst + (long) line * a.dpitch, a.src + (long) line * a.spitch, k, a);
}

// The converters that have no hand-written fast path below take the wrapper pixfmt_ext.hip uses for the rest of decoders[]: K
// iterations per lane, the K * SB source bytes and K * DB output bytes of a lane moved with 128-bit accesses (whole-wave contiguous
// regions, ug::UnitIO), the unchanged run() working on private arrays.  VecTraits<CONV>::K == 0: no such form.
template <class CONV> struct VecTraits { static constexpr int SB = 16, DB = 16, K = 0; };
template <int RO, int GO, int BO> struct VecTraits<ToUYVY<RO, GO, BO, 4>> { static constexpr int SB = 8, DB = 4, K = 4; };   // RGBA pairs
template <int RO, int GO, int BO> struct VecTraits<ToUYVY<RO, GO, BO, 6>> { static constexpr int SB = 12, DB = 4, K = 4; };  // RG48 pairs
template <> struct VecTraits<V210toRGB<true>> { static constexpr int SB = 16, DB = 36, K = 4; };                               // v210 -> RG48
template <> struct VecTraits<RGBshift> { static constexpr int SB = 3, DB = 3, K = 16; };                                        // BGR -> RGB, RGB with shifts
template <> struct VecTraits<RGBAshift> { static constexpr int SB = 4, DB = 4, K = 4; };

template <class CONV, int SB, int DB, int K>
__global__ __launch_bounds__(256) void generic_vec_kernel(Args a, int nvec)
{
        static_assert((K * SB) % 16 == 0 && (K * DB) % 16 == 0, "a vector unit moves whole 16-byte words");
        using In = ug::UnitIO<K * SB>;
        using Out = ug::UnitIO<K * DB>;
        constexpr int kLdsWords = In::LDS_WORDS > Out::LDS_WORDS ? In::LDS_WORDS : Out::LDS_WORDS;
        __shared__ uint4 lds_all[kLdsWords ? 4 * kLdsWords : 1];
        const int lane = threadIdx.x, y = blockIdx.y * 4 + threadIdx.y; // a wave = 64 consecutive units of one line
        const int u0 = blockIdx.x * 64;
        if (y >= a.height || u0 >= nvec) return; // wave-uniform
        const int units = min(64, nvec - u0), u = u0 + lane;
        uint4 *const lds = lds_all + threadIdx.y * kLdsWords;
        __attribute__((aligned(16))) uint8_t ls[K * SB];
        __attribute__((aligned(16))) uint8_t ld[K * DB];
        In::load((const uint4 *) (a.src + (long) y * a.spitch + (long) u0 * (K * SB)), ls, lds, lane, units);
        Args b = a;
        b.dst_len = a.dst_len - u * (K * DB); // the line as this unit sees it
#pragma unroll
        for (int k = 0; k < K; k++) CONV::run(ld, ls, k, b);
        Out::store((uint4 *) (a.dst + (long) y * a.dpitch + (long) u0 * (K * DB)), ld, lds, lane, units);
}

template <class CONV>
int launch_generic(const Args &a, hipStream_t st)
{
        using VT = VecTraits<CONV>;
        int k0 = 0;
        if constexpr (VT::K > 0) if (!((((uintptr_t) a.src | (uintptr_t) a.dst) | (uintptr_t) a.spitch | (uintptr_t) a.dpitch) & 15)) {
                // whole units whose iterations write all their bytes and read inside the source line (the rest of the line: below)
                const int nvec = min(a.dst_len / (VT::K * VT::DB), a.src_line / (VT::K * VT::SB));
                if (nvec > 0) {
                        hipLaunchKernelGGL((generic_vec_kernel<CONV, VT::SB, VT::DB, VT::K>), dim3((unsigned) ((nvec + 63) / 64), (unsigned) ((a.height + 3) / 4)),
                                           dim3(64, 4), 0, st, a, nvec);
                        UG_HIP_LAUNCH_CHECK();
                        k0 = nvec * VT::K;
                }
        }
        const int upl = CONV::units(a.dst_len) - k0;
        const long total = (long) upl * a.height;
        if (total <= 0) return UG_HIP_SUCCESS;
        hipLaunchKernelGGL((generic_kernel<CONV>), dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, a, upl, k0, total);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

// ------------------------------ fast paths (aligned geometry) ------------------------------
// Lane i of the grid handles chunk i of a line; chunks are laid end to end so a wave reads and
// writes contiguous memory.  Requires width % PX == 0 and 16-byte aligned bases / pitches.

// v210 -> UYVY: 16 B (6 px) -> 12 B.  Lane: 2 groups = 32 B in, 24 B out.
struct FastV210toUYVY {
        static constexpr int PX = 12, W = 6;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &)
        {
                const uint4 *sp = (const uint4 *) s + 2 * c;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                        const uint4 q = sp[g];
#define S(w, f) ((((w) >> (10 * (f) + 2)) & 0xffu))
                        o[3 * g + 0] = S(q.x, 0) | S(q.x, 1) << 8 | S(q.x, 2) << 16 | S(q.y, 0) << 24;
                        o[3 * g + 1] = S(q.y, 1) | S(q.y, 2) << 8 | S(q.z, 0) << 16 | S(q.z, 1) << 24;
                        o[3 * g + 2] = S(q.z, 2) | S(q.w, 0) << 8 | S(q.w, 1) << 16 | S(q.w, 2) << 24;
#undef S
                }
        }
};
// UYVY -> RGB: lane 16 B (8 px) in, 24 B out
struct FastUYVYtoRGB {
        static constexpr int PX = 8, W = 6;
        static __device__ void run(const uint8_t *s, uint32_t (&ow)[W], int c, const Args &)
        {
                const uint4 q = ((const uint4 *) s)[c];
                const uint32_t w[4] = { q.x, q.y, q.z, q.w };
                uint8_t o[24];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        const int u = (int) (w[i] & 0xff) - 128, v = (int) ((w[i] >> 16) & 0xff) - 128;
                        yuv_to_rgb8(kCfs8.y_scale * ((int) ((w[i] >> 8) & 0xff) - 16), u, v, o + 6 * i);
                        yuv_to_rgb8(kCfs8.y_scale * ((int) (w[i] >> 24) - 16), u, v, o + 6 * i + 3);
                }
#pragma unroll
                for (int i = 0; i < 6; i++) ow[i] = o[4 * i] | o[4 * i + 1] << 8 | o[4 * i + 2] << 16 | (uint32_t) o[4 * i + 3] << 24;
        }
};
// RGB -> UYVY: lane 24 B (8 px) in, 16 B out
template <int RO, int BO>
struct FastRGBtoUYVY {
        static constexpr int PX = 8, W = 4;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &)
        {
                const uint2 *sp = (const uint2 *) s + 3 * c;
                uint8_t b[24];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        const uint2 q = sp[i];
#pragma unroll
                        for (int j = 0; j < 4; j++) { b[8 * i + j] = q.x >> (8 * j); b[8 * i + 4 + j] = q.y >> (8 * j); }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        const uint8_t *p = b + 6 * i;
                        int r = p[RO], g = p[1], bb = p[BO];
                        const int y1 = ((r * kCfs8.y_r + g * kCfs8.y_g + bb * kCfs8.y_b) >> kBase) + 16;
                        int u = r * kCfs8.cb_r + g * kCfs8.cb_g + bb * kCfs8.cb_b;
                        int v = r * kCfs8.cr_r + g * kCfs8.cr_g + bb * kCfs8.cr_b;
                        r = p[3 + RO]; g = p[4]; bb = p[3 + BO];
                        const int y2 = ((r * kCfs8.y_r + g * kCfs8.y_g + bb * kCfs8.y_b) >> kBase) + 16;
                        u += r * kCfs8.cb_r + g * kCfs8.cb_g + bb * kCfs8.cb_b;
                        v += r * kCfs8.cr_r + g * kCfs8.cr_g + bb * kCfs8.cr_b;
                        u = ((u / 2) >> kBase) + 128;
                        v = ((v / 2) >> kBase) + 128;
                        o[i] = ((uint32_t) (y2 & 0xFF) << 24) | ((v & 0xFF) << 16) | ((y1 & 0xFF) << 8) | (u & 0xFF);
                }
        }
};
// v210 -> RGB (8-bit): lane 32 B (12 px) in, 36 B out
struct FastV210toRGB {
        static constexpr int PX = 12, W = 9;
        static __device__ void run(const uint8_t *s, uint32_t (&ow)[W], int c, const Args &)
        {
                const uint4 *sp = (const uint4 *) s + 2 * c;
                uint8_t o[36];
#pragma unroll
                for (int g = 0; g < 2; g++) {
                        const uint4 q = sp[g];
                        const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                                const int ys = 2 * i + 1, us = 4 * (i / 2), vs = us + 2;
                                const int Y = (int) ((w[ys / 3] >> (10 * (ys % 3) + 2)) & 0xffu);
                                const int u = (int) ((w[us / 3] >> (10 * (us % 3) + 2)) & 0xffu) - 128;
                                const int v = (int) ((w[vs / 3] >> (10 * (vs % 3) + 2)) & 0xffu) - 128;
                                const int y = kCfs8.y_scale * (Y - 16);
                                o[18 * g + 3 * i + 0] = clampi((y + v * kCfs8.r_cr) >> kBase, 1, 254);
                                o[18 * g + 3 * i + 1] = clampi((y + u * kCfs8.g_cb + v * kCfs8.g_cr) >> kBase, 1, 254);
                                o[18 * g + 3 * i + 2] = clampi((y + u * kCfs8.b_cb) >> kBase, 1, 254);
                        }
                }
#pragma unroll
                for (int i = 0; i < 9; i++) ow[i] = o[4 * i] | o[4 * i + 1] << 8 | o[4 * i + 2] << 16 | (uint32_t) o[4 * i + 3] << 24;
        }
};
// RGBA -> RGB: lane 16 B (4 px) in, 12 B out ; RGB -> RGBA the inverse
struct FastRGBAtoRGB {
        static constexpr int PX = 4, W = 3;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &)
        {
                const uint4 q = ((const uint4 *) s)[c];
                o[0] = (q.x & 0xffffff) | (q.y << 24);
                o[1] = ((q.y >> 8) & 0xffff) | (q.z << 16);
                o[2] = ((q.z >> 16) & 0xff) | (q.w << 8);
        }
};

// UYVY <-> YUYV: lane 16 B, swap the bytes of every 16-bit pair
struct FastSwapYUYV {
        static constexpr int PX = 8, W = 4;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &)
        {
                uint4 q = ((const uint4 *) s)[c];
#define SW(w) ((((w) & 0x00ff00ffu) << 8) | (((w) >> 8) & 0x00ff00ffu))
                q.x = SW(q.x); q.y = SW(q.y); q.z = SW(q.z); q.w = SW(q.w);
#undef SW
                o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
        }
};
// RGB -> RGBA with shifts: lane 12 B (4 px) in, 16 B out
struct FastRGBtoRGBA {
        static constexpr int PX = 4, W = 4;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &a)
        {
                const uint32_t *sp = (const uint32_t *) s + 3 * c;
                const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2];
                const uint32_t am = alpha_mask(a.rs, a.gs, a.bs);
                const uint32_t px[4] = { w0 & 0xffffff, (w0 >> 24) | ((w1 & 0xffff) << 8), (w1 >> 16) | ((w2 & 0xff) << 16), w2 >> 8 };
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        o[i] = am | (px[i] & 0xff) << a.rs | ((px[i] >> 8) & 0xff) << a.gs | (px[i] >> 16) << a.bs;
                }
        }
};
// UYVY -> v210: lane 24 B (12 px) in, 32 B out; consecutive bytes -> 10-bit fields (<< 2), three per word
struct FastUYVYtoV210 {
        static constexpr int PX = 12, W = 8;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &)
        {
                const uint2 *sp = (const uint2 *) s + 3 * c;
                const uint2 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                const uint32_t w[6] = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
#pragma unroll
                for (int k = 0; k < 8; k++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int f = 0; f < 3; f++) {
                                const int b = 3 * k + f; // source byte index 0..23
                                v |= (((w[b / 4] >> (8 * (b % 4))) & 0xffu) << 2) << (10 * f);
                        }
                        o[k] = v;
                }
        }
};
// UYVY -> RGBA (fp64 arithmetic of vc_copylineUYVYtoRGBA, pixfmt_conv.c:1137-1163): lane 16 B (8 px) in, 32 B out
struct FastUYVYtoRGBA {
        static constexpr int PX = 8, W = 8;
        static __device__ void run(const uint8_t *s, uint32_t (&o)[W], int c, const Args &a)
        {
                const uint4 q = ((const uint4 *) s)[c];
                const uint32_t w[4] = { q.x, q.y, q.z, q.w };
                const uint32_t am = alpha_mask(a.rs, a.gs, a.bs);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        const int u = w[i] & 0xff, v = (w[i] >> 16) & 0xff;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                                const int y = h ? (int) (w[i] >> 24) : (int) ((w[i] >> 8) & 0xff);
                                int r = 1.164 * (y - 16) + 1.793 * (v - 128);
                                int g = 1.164 * (y - 16) - 0.534 * (v - 128) - 0.213 * (u - 128);
                                int b = 1.164 * (y - 16) + 2.115 * (u - 128);
                                r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                                o[2 * i + h] = am | (uint32_t) r << a.rs | (uint32_t) g << a.gs | (uint32_t) b << a.bs;
                        }
                }
        }
};

// A wave = 64 consecutive chunks of one line; the chunks' output words leave as one contiguous region (ug::WaveWords).
template <class F>
__global__ __launch_bounds__(256) void fast_kernel(Args a, int cpl)
{
        using WS = ug::WaveWords<F::W>;
        __shared__ uint32_t lds_all[WS::LDS_DWORDS ? 4 * WS::LDS_DWORDS : 1];
        const int c = blockIdx.x * blockDim.x + threadIdx.x, line = blockIdx.y;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
        const int c0 = c - lane;
        if (c0 >= cpl) return; // wave-uniform
        const int units = min(64, cpl - c0);
        uint32_t o[F::W];
        if (lane < units) F::run(a.src + (long) line * a.spitch, o, c, a);
        WS::store(a.dst + (long) line * a.dpitch + (long) c0 * (F::W * 4), o, lds_all + wave * WS::LDS_DWORDS, lane, units);
}

template <class F>
bool try_fast(const Args &a, hipStream_t st, int &rc)
{
        if (a.width % F::PX || (a.spitch & 15) || (a.dpitch & 15) || (15 & (uintptr_t) a.src) || (15 & (uintptr_t) a.dst) ||
            a.height > 65535) {
                return false;
        }
        const int cpl = a.width / F::PX;
        int bx = 64; // workgroup width with the fewest idle lanes at the end of a line (ties: wider)
        for (int cand : { 128, 256 }) {
                if ((cpl + cand - 1) / cand * cand <= (cpl + bx - 1) / bx * bx) bx = cand;
        }
        hipLaunchKernelGGL((fast_kernel<F>), dim3((unsigned) ((cpl + bx - 1) / bx), (unsigned) a.height), dim3(bx), 0, st, a, cpl);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ug::set_last_error(e, "kernel launch"); rc = UG_HIP_ERUNTIME; }
        else rc = UG_HIP_SUCCESS;
        return true;
}

int size_of(ug_pixfmt_t f, int width) // vc_get_size, video_codec.c:530-538
{
        switch (f) {
        case UG_PF_RGBA: return width * 4;
        case UG_PF_UYVY:
        case UG_PF_YUYV: return (width + 1) / 2 * 4;
        case UG_PF_RGB:
        case UG_PF_BGR: return width * 3;
        case UG_PF_DVS10:
        case UG_PF_V210: return (width + 5) / 6 * 16;
        case UG_PF_RG48: return width * 6;
        case UG_PF_R10K:
        case UG_PF_VUYA: return width * 4;
        case UG_PF_R12L: return (width + 7) / 8 * 36;
        case UG_PF_Y216: return (width + 1) / 2 * 8;
        case UG_PF_Y416: return width * 8;
        default: return 0;
        }
}

#define PAIR(a, b) ((a) * 32 + (b))

// ------------------------------ packed -> planar ------------------------------
// uyvy_to_i420, to_planar.c:343-378: one lane per (row pair, pixel pair)
__global__ __launch_bounds__(256) void uyvy_to_i420_kernel(const uint8_t *__restrict__ src, int spitch, uint8_t *__restrict__ yp,
                                                           int ypitch, uint8_t *__restrict__ up, int upitch,
                                                           uint8_t *__restrict__ vp, int vpitch, int width, int height)
{
        const int pairs = (width + 1) / 2;
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        const long total = (long) pairs * ((height + 1) / 2);
        if (idx >= total) return;
        const int i = (int) (idx / pairs), j = (int) (idx - (long) i * pairs);
        const uint8_t *in1 = src + (long) (2 * i) * spitch + 4 * j;
        const bool last_odd = 2 * i + 1 == height;
        const uint8_t *in2 = last_odd ? in1 : in1 + spitch;
        uint8_t *y1 = yp + (long) (2 * i) * ypitch + 2 * j;
        uint8_t *y2 = last_odd ? y1 : y1 + ypitch;
        up[(long) i * upitch + j] = (in1[0] + in2[0] + 1) / 2;
        vp[(long) i * vpitch + j] = (in1[2] + in2[2] + 1) / 2;
        y1[0] = in1[1]; y2[0] = in2[1];
        if (2 * j + 1 < width) { // width odd: the trailing pair carries one luma only
                y1[1] = in1[3]; y2[1] = in2[3];
        }
}
// fast: lane = 16 B of two lines (8 px)
__global__ __launch_bounds__(256) void uyvy_to_i420_fast(const uint8_t *__restrict__ src, int spitch, uint8_t *__restrict__ yp,
                                                         int ypitch, uint8_t *__restrict__ up, int upitch,
                                                         uint8_t *__restrict__ vp, int vpitch, int cpl, long total)
{
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= total) return;
        const int i = (int) (idx / cpl), c = (int) (idx - (long) i * cpl);
        const uint4 a = ((const uint4 *) (src + (long) (2 * i) * spitch))[c];
        const uint4 b = ((const uint4 *) (src + (long) (2 * i + 1) * spitch))[c];
        const uint32_t wa[4] = { a.x, a.y, a.z, a.w }, wb[4] = { b.x, b.y, b.z, b.w };
        uint32_t ya[2] = { 0, 0 }, yb[2] = { 0, 0 }, uu = 0, vv = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
                ya[k / 2] |= (((wa[k] >> 8) & 0xff) | ((wa[k] >> 24) << 8)) << (16 * (k & 1));
                yb[k / 2] |= (((wb[k] >> 8) & 0xff) | ((wb[k] >> 24) << 8)) << (16 * (k & 1));
                uu |= (((wa[k] & 0xff) + (wb[k] & 0xff) + 1) >> 1) << (8 * k);
                vv |= ((((wa[k] >> 16) & 0xff) + ((wb[k] >> 16) & 0xff) + 1) >> 1) << (8 * k);
        }
        ug::st_stream(&((uint2 *) (yp + (long) (2 * i) * ypitch))[c], make_uint2(ya[0], ya[1]));
        ug::st_stream(&((uint2 *) (yp + (long) (2 * i + 1) * ypitch))[c], make_uint2(yb[0], yb[1]));
        ug::st_stream(&((uint32_t *) (up + (long) i * upitch))[c], uu);
        ug::st_stream(&((uint32_t *) (vp + (long) i * vpitch))[c], vv);
}
// v210_to_p010le, to_planar.c:64-155, the aligned regular case (width % 6 == 0, even height, 4-byte aligned planes): lane = one 6-px
// group of a row pair, three 32-bit stores per line
// (blockIdx.y = the row pair; a wave = 64 consecutive groups of it, whose 12-byte pieces leave as contiguous runs: ug::WaveWords)
__global__ __launch_bounds__(256) void v210_to_p010le_kernel(const uint8_t *__restrict__ src, int spitch, uint8_t *__restrict__ yp,
                                                             int ypitch, uint8_t *__restrict__ uvp, int uvpitch, int gpl)
{
        using WS = ug::WaveWords<3>;
        __shared__ uint32_t lds_all[4 * WS::LDS_DWORDS];
        const int g = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
        const int lane = threadIdx.x & 63, g0 = g - lane;
        if (g0 >= gpl) return; // wave-uniform
        const int units = min(64, gpl - g0);
        uint32_t *const lds = lds_all + __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6) * WS::LDS_DWORDS;
        const int gl = min(g, gpl - 1); // lanes past the line read the last group again; their words are not stored
        const uint4 a = ((const uint4 *) (src + (long) (2 * i) * spitch))[gl];
        const uint4 b = ((const uint4 *) (src + (long) (2 * i + 1) * spitch))[gl];
        const uint32_t wa[4] = { a.x, a.y, a.z, a.w }, wb[4] = { b.x, b.y, b.z, b.w };
        uint32_t o0[3], o1[3], oc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
                uint32_t v0[2], v1[2], vc[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                        const int px = 2 * k + h;         // luma sample index, chroma sample index (Cb Cr Cb Cr ...)
                        const int ys = 2 * px + 1;        // UYVY order: Y at odd positions
                        const int cs = 2 * px;            // U0 V0 U2 V2 U4 V4 at even positions
                        v0[h] = ((wa[ys / 3] >> (10 * (ys % 3))) & 0x3ffu) << 6;
                        v1[h] = ((wb[ys / 3] >> (10 * (ys % 3))) & 0x3ffu) << 6;
                        const uint32_t ca = (wa[cs / 3] >> (10 * (cs % 3))) & 0x3ffu, cb = (wb[cs / 3] >> (10 * (cs % 3))) & 0x3ffu;
                        vc[h] = ((ca + cb) / 2) << 6;
                }
                o0[k] = v0[0] | v0[1] << 16; o1[k] = v1[0] | v1[1] << 16; oc[k] = vc[0] | vc[1] << 16;
        }
        WS::store(yp + (long) (2 * i) * ypitch + 12 * g0, o0, lds, lane, units);
        WS::store(yp + (long) (2 * i + 1) * ypitch + 12 * g0, o1, lds, lane, units);
        WS::store(uvp + (long) i * uvpitch + 12 * g0, oc, lds, lane, units);
}

// One v210 group (4 words) -> its six luma samples and six chroma samples (Cb Cr Cb Cr Cb Cr), 10 bits each
__device__ __forceinline__ void v210_group_samples(const uint32_t *__restrict__ row, int g, uint32_t luma[6], uint32_t chroma[6])
{
        const uint32_t w[4] = { row[4 * g], row[4 * g + 1], row[4 * g + 2], row[4 * g + 3] };
#pragma unroll
        for (int s = 0; s < 6; s++) {
                luma[s] = (w[(2 * s + 1) / 3] >> (10 * ((2 * s + 1) % 3))) & 0x3ffu;
                chroma[s] = (w[(2 * s) / 3] >> (10 * ((2 * s) % 3))) & 0x3ffu;
        }
}

// v210_to_p010le for every geometry the reference converts (to_planar.c:64-155) with width >= 6: lane = one 6-px group of a row pair,
// ceil(width / 6) groups per line.  What the reference's serial loop leaves in memory, restated per lane:
//  * pairs with more than two lines below them ("interior", :85-89 w = roundup6(width)) write the WHOLE last group, i.e. up to 5
//    samples past `width`.  With a pitch shorter than roundup6(width) samples, the even line's tail lands on the first samples of
//    the odd line of the same pair and is written AFTER them (:113-125 run group by group), so it stays; the odd line's tail and
//    the chroma tail land on lines the next pair rewrites, so they do not.
//  * the last pair (one or two lines, :87-89 w = width) converts width / 6 whole groups; the width % 6 samples behind them are
//    copied (:139-150) from `dst - out_linesize` -- a uint16_t pointer moved by a BYTE count, i.e. from two lines above: luma of
//    both lines from line y - 2, chroma from chroma line y/2 - 2.  Those hold the own samples of interior pairs.
//  * an odd last line is converted alone (:80-83): its chroma is (c + c) / 2.
__global__ __launch_bounds__(256) void v210_to_p010le_any_kernel(const uint8_t *__restrict__ src, int spitch, uint16_t *__restrict__ yp,
                                                                 int lw, uint16_t *__restrict__ uvp, int lc, int width, int height,
                                                                 int gpl, long total)
{
        const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= total) return;
        const int i = (int) (idx / gpl), g = (int) (idx - (long) i * gpl);
        const int y = 2 * i, rem = height - y, whole = width / 6;
        const bool last = rem <= 2, margin = last && g >= whole;
        const int ra = margin ? y - 2 : y, rb = margin ? y - 2 : (rem == 1 ? y : y + 1); // luma source lines
        const int ca = margin ? y - 4 : ra, cb = margin ? y - 3 : rb;                    // chroma source lines
        uint32_t la[6], lb[6], ua[6], ub[6];
        v210_group_samples((const uint32_t *) (src + (long) ra * spitch), g, la, ua);
        v210_group_samples((const uint32_t *) (src + (long) rb * spitch), g, lb, ub);
        if (margin) {
                uint32_t unused[6];
                v210_group_samples((const uint32_t *) (src + (long) ca * spitch), g, unused, ua);
                v210_group_samples((const uint32_t *) (src + (long) cb * spitch), g, unused, ub);
        }
        const int n = margin ? width - 6 * whole : 6;
        const int over = 6 * gpl - lw; // samples of an interior even line that run into the odd line
#pragma unroll
        for (int s = 0; s < 6; s++) {
                const int col = 6 * g + s;
                const uint16_t c = (uint16_t) (((ua[s] + ub[s]) / 2) << 6);
                if (!last) {
                        yp[(long) y * lw + col] = (uint16_t) (la[s] << 6); // col >= lw: the first samples of line y + 1
                        if (col >= over && col < lw) yp[(long) (y + 1) * lw + col] = (uint16_t) (lb[s] << 6);
                        if (col < lc) uvp[(long) i * lc + col] = c;
                } else if (s < n) {
                        yp[(long) y * lw + col] = (uint16_t) (la[s] << 6);
                        if (rem == 2) yp[(long) (y + 1) * lw + col] = (uint16_t) (lb[s] << 6);
                        uvp[(long) i * lc + col] = c;
                }
        }
}

// width < 6 (one group per line, no whole group on the last pair): the tails of BOTH luma lines overlap their neighbours and the
// last pair is nothing but the copy from two lines above, so the outcome depends on the order of the writes -- one lane walks the
// picture in the reference's order (to_planar.c:72-151).  Frames this narrow are a few hundred bytes.  With a pitch below 6 samples
// the reference's last tails run past the end of its planes; those writes are dropped here (yend / uvend = samples in the plane).
__global__ void v210_to_p010le_narrow_kernel(const uint8_t *__restrict__ src, int spitch, uint16_t *yp, int lw, uint16_t *uvp, int lc,
                                             int width, int height)
{
        if (blockIdx.x || threadIdx.x) return;
        const long yend = (long) height * lw, uvend = (long) ((height + 1) / 2) * lc;
        for (int y = 0; y < height; y += 2) {
                const int rem = height - y;
                const long p0 = (long) y * lw, p1 = p0 + lw, pc = (long) (y / 2) * lc;
                if (rem > 2) {
                        uint32_t la[6], lb[6], ua[6], ub[6];
                        v210_group_samples((const uint32_t *) (src + (long) y * spitch), 0, la, ua);
                        v210_group_samples((const uint32_t *) (src + (long) (y + 1) * spitch), 0, lb, ub);
                        for (int s = 0; s < 6; s++) if (p0 + s < yend) yp[p0 + s] = (uint16_t) (la[s] << 6);
                        for (int s = 0; s < 6; s++) if (p1 + s < yend) yp[p1 + s] = (uint16_t) (lb[s] << 6);
                        for (int s = 0; s < 6; s++) if (pc + s < uvend) uvp[pc + s] = (uint16_t) (((ua[s] + ub[s]) / 2) << 6);
                } else {
                        for (int s = 0; s < width; s++) yp[p0 + s] = yp[p0 + s - 2 * (long) lw];
                        if (rem == 2) for (int s = 0; s < width; s++) yp[p1 + s] = yp[p0 + s - 2 * (long) lw];
                        for (int s = 0; s < width; s++) uvp[pc + s] = uvp[pc + s - 2 * (long) lc];
                }
        }
}

// cuda_yuv422_to_yuv444 (cuda_dxt.cu:697-732): lane = 4 px (8 B -> 12 B)
__global__ __launch_bounds__(256) void yuv422_to_yuv444_kernel(const uint2 *__restrict__ src, uint32_t *__restrict__ dst, int quads)
{
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= quads) return;
        const uint2 q = src[i];
        const uint32_t u0 = q.x & 0xff, y0 = (q.x >> 8) & 0xff, v0 = (q.x >> 16) & 0xff, y1 = q.x >> 24;
        const uint32_t u1 = q.y & 0xff, y2 = (q.y >> 8) & 0xff, v1 = (q.y >> 16) & 0xff, y3 = q.y >> 24;
        dst[3 * i + 0] = y0 | u0 << 8 | v0 << 16 | y1 << 24;
        dst[3 * i + 1] = u0 | v0 << 8 | y2 << 16 | u1 << 24;
        dst[3 * i + 2] = v1 | y3 << 8 | u1 << 16 | v1 << 24;
}

} // namespace

extern "C" {

int ug_hip_pixfmt_supported(ug_pixfmt_t in, ug_pixfmt_t out)
{
        if (in == out && in != UG_PF_NONE && size_of(in, 2) != 0) return 1;
        if (ug::pixfmt_ext_supported(in, out)) return 1;
        switch (PAIR(in, out)) {
        case PAIR(UG_PF_V210, UG_PF_UYVY): case PAIR(UG_PF_YUYV, UG_PF_UYVY): case PAIR(UG_PF_UYVY, UG_PF_YUYV):
        case PAIR(UG_PF_UYVY, UG_PF_RGB): case PAIR(UG_PF_UYVY, UG_PF_RGBA): case PAIR(UG_PF_RGB, UG_PF_UYVY):
        case PAIR(UG_PF_BGR, UG_PF_UYVY): case PAIR(UG_PF_RGBA, UG_PF_UYVY): case PAIR(UG_PF_RG48, UG_PF_UYVY):
        case PAIR(UG_PF_V210, UG_PF_RGB): case PAIR(UG_PF_V210, UG_PF_RG48): case PAIR(UG_PF_RGBA, UG_PF_RGB):
        case PAIR(UG_PF_RGB, UG_PF_RGBA): case PAIR(UG_PF_BGR, UG_PF_RGB): case PAIR(UG_PF_UYVY, UG_PF_V210):
                return 1;
        }
        return 0;
}

// the converter behind ug_hip_pixfmt_convert: every argument check but the bound on the number of lines -- a batch of frames laid end to end
// is handed over as ONE picture that may be taller than any frame (the kernels index lines with 64-bit offsets)
static int pixfmt_convert_lines(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height,
                                int src_pitch, int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream);

// `frames` images at constant strides.  Frames that follow each other exactly one picture apart (stride == pitch * height on both
// sides -- tiles and frame rings are laid out like that) are ONE picture of frames * height lines to every line converter, so the
// whole batch is a single launch; any other layout is converted frame by frame.
int ug_hip_pixfmt_convert_batch(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height, int src_pitch,
                                int dst_pitch, int rshift, int gshift, int bshift, int frames, size_t src_frame_stride,
                                size_t dst_frame_stride, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_pixfmt_convert_batch");
        if (frames < 0 || src_pitch < 0 || dst_pitch < 0) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert_batch: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (frames == 0) return UG_HIP_SUCCESS;
        const int sp = src_pitch ? src_pitch : ug::linesize(in, width), dp = dst_pitch ? dst_pitch : ug::linesize(out, width);
        if (sp <= 0 || dp <= 0) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert_batch: unsupported format");
                return UG_HIP_EUNSUPP;
        }
        if (!ug::span_ok(sp, height) || !ug::span_ok(dp, height)) return ug::refuse_size("ug_hip_pixfmt_convert_batch");
        // (one picture of at most 4 * 65535 lines: the line kernels put four lines into a workgroup and the grid's y extent ends at 65535)
        if (frames == 1 || (src_frame_stride == (size_t) sp * height && dst_frame_stride == (size_t) dp * height &&
                            (long long) height * frames <= 4LL * 65535)) {
                return pixfmt_convert_lines(in, out, src, dst, width, height * frames, sp, dp, rshift, gshift, bshift, stream);
        }
        for (int f = 0; f < frames; f++) {
                const int rc = ug_hip_pixfmt_convert(in, out, (const uint8_t *) src + (size_t) f * src_frame_stride,
                                                     (uint8_t *) dst + (size_t) f * dst_frame_stride, width, height, sp, dp, rshift, gshift,
                                                     bshift, stream);
                if (rc != UG_HIP_SUCCESS) return rc;
        }
        return UG_HIP_SUCCESS;
}

int ug_hip_pixfmt_convert(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height,
                          int src_pitch, int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_pixfmt_convert");
        if (src_pitch < 0 || dst_pitch < 0 || !ug::span_ok(src_pitch ? src_pitch : ug::linesize(in, width), height) ||
            !ug::span_ok(dst_pitch ? dst_pitch : ug::linesize(out, width), height)) {
                return ug::refuse_size("ug_hip_pixfmt_convert");
        }
        return pixfmt_convert_lines(in, out, src, dst, width, height, src_pitch, dst_pitch, rshift, gshift, bshift, stream);
}

static int pixfmt_convert_lines(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height,
                                int src_pitch, int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream)
{
        if (!src || !dst || width <= 0 || height <= 0) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (!ug_hip_pixfmt_supported(in, out)) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: unsupported conversion");
                return UG_HIP_EUNSUPP;
        }
        // same argument checks as the decoders[] extension (pixfmt_ext_convert) and the lavc converters: a shift of 32 or more is
        // undefined on host and device alike, and a pitch shorter than the line makes lines overwrite each other
        if ((unsigned) rshift > 24 || (unsigned) gshift > 24 || (unsigned) bshift > 24) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: component shifts must be in [0, 24]");
                return UG_HIP_EINVAL;
        }
        if ((dst_pitch && dst_pitch < size_of(out, width)) || (src_pitch && src_pitch < size_of(in, width))) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: pitch smaller than the line");
                return UG_HIP_EINVAL;
        }
        Args a;
        a.src = (const uint8_t *) src; a.dst = (uint8_t *) dst; a.width = width; a.height = height;
        a.spitch = src_pitch ? src_pitch : ug::linesize(in, width);
        a.dpitch = dst_pitch ? dst_pitch : ug::linesize(out, width);
        a.dst_len = size_of(out, width);
        a.src_line = ug::linesize(in, width);
        a.rs = rshift; a.gs = gshift; a.bs = bshift;
        hipStream_t st = (hipStream_t) stream;
        int rc = UG_HIP_SUCCESS;

        if (in == out && out != UG_PF_RGBA && out != UG_PF_RGB) { // get_decoder_from_to, pixfmt_conv.c:3111-3114
                return launch_generic<Copy>(a, st);
        }
        if (ug::pixfmt_ext_supported(in, out)) {
                return ug::pixfmt_ext_convert(in, out, src, dst, width, height, a.spitch, a.dpitch, a.dst_len, rshift, gshift, bshift, st);
        }
        switch (PAIR(in, out)) {
        case PAIR(UG_PF_V210, UG_PF_UYVY):
                if (try_fast<FastV210toUYVY>(a, st, rc)) return rc;
                return launch_generic<V210toUYVY>(a, st);
        case PAIR(UG_PF_YUYV, UG_PF_UYVY):
        case PAIR(UG_PF_UYVY, UG_PF_YUYV):
                if (try_fast<FastSwapYUYV>(a, st, rc)) return rc;
                return launch_generic<SwapYUYV>(a, st);
        case PAIR(UG_PF_UYVY, UG_PF_RGB):
                if (try_fast<FastUYVYtoRGB>(a, st, rc)) return rc;
                return launch_generic<UYVYtoRGB>(a, st);
        case PAIR(UG_PF_UYVY, UG_PF_RGBA):
                if (try_fast<FastUYVYtoRGBA>(a, st, rc)) return rc;
                return launch_generic<UYVYtoRGBA>(a, st);
        case PAIR(UG_PF_RGB, UG_PF_UYVY):
                if (try_fast<FastRGBtoUYVY<0, 2>>(a, st, rc)) return rc;
                return launch_generic<ToUYVY<0, 1, 2, 3>>(a, st);
        case PAIR(UG_PF_BGR, UG_PF_UYVY):
                if (try_fast<FastRGBtoUYVY<2, 0>>(a, st, rc)) return rc;
                return launch_generic<ToUYVY<2, 1, 0, 3>>(a, st);
        case PAIR(UG_PF_RGBA, UG_PF_UYVY): return launch_generic<ToUYVY<0, 1, 2, 4>>(a, st);
        case PAIR(UG_PF_RG48, UG_PF_UYVY): return launch_generic<ToUYVY<1, 3, 5, 6>>(a, st);
        case PAIR(UG_PF_V210, UG_PF_RGB):
                if (try_fast<FastV210toRGB>(a, st, rc)) return rc;
                return launch_generic<V210toRGB<false>>(a, st);
        case PAIR(UG_PF_V210, UG_PF_RG48): return launch_generic<V210toRGB<true>>(a, st);
        case PAIR(UG_PF_RGBA, UG_PF_RGB):
                if (try_fast<FastRGBAtoRGB>(a, st, rc)) return rc;
                return launch_generic<RGBAtoRGB>(a, st);
        case PAIR(UG_PF_RGB, UG_PF_RGBA):
                if (try_fast<FastRGBtoRGBA>(a, st, rc)) return rc;
                return launch_generic<RGBtoRGBA>(a, st);
        case PAIR(UG_PF_RGBA, UG_PF_RGBA):
                if (rshift == 0 && gshift == 8 && bshift == 16) return launch_generic<Copy>(a, st);
                return launch_generic<RGBAshift>(a, st);
        case PAIR(UG_PF_RGB, UG_PF_RGB):
                if (rshift == 0 && gshift == 8 && bshift == 16) return launch_generic<Copy>(a, st);
                return launch_generic<RGBshift>(a, st);
        case PAIR(UG_PF_BGR, UG_PF_RGB): // vc_copylineBGRtoRGB, pixfmt_conv.c:2520-2527
                a.rs = 16; a.gs = 8; a.bs = 0;
                return launch_generic<RGBshift>(a, st);
        case PAIR(UG_PF_UYVY, UG_PF_V210):
                if (try_fast<FastUYVYtoV210>(a, st, rc)) return rc;
                return launch_generic<UYVYtoV210>(a, st);
        }
        return UG_HIP_EUNSUPP;
}

int ug_hip_uyvy_to_i420(const void *src, int src_pitch, void *y, int y_pitch, void *u, int u_pitch, void *v,
                        int v_pitch, int width, int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_uyvy_to_i420");
        if (!src || !y || !u || !v || width <= 0 || height <= 0) return UG_HIP_EINVAL;
        if (!src_pitch) src_pitch = ug::linesize(UG_PF_UYVY, width);
        if (!ug::planes_ok(height, { src_pitch, y_pitch, u_pitch, v_pitch })) return ug::refuse_size("ug_hip_uyvy_to_i420");
        hipStream_t st = (hipStream_t) stream;
        const bool fast = width % 8 == 0 && height % 2 == 0 && !(src_pitch & 15) && !(y_pitch & 7) && !(u_pitch & 3) &&
                          !(v_pitch & 3) && !(15 & (uintptr_t) src) && !(7 & (uintptr_t) y) && !(3 & (uintptr_t) u) &&
                          !(3 & (uintptr_t) v);
        if (fast) {
                const int cpl = width / 8;
                const long total = (long) cpl * (height / 2);
                hipLaunchKernelGGL(uyvy_to_i420_fast, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st,
                                   (const uint8_t *) src, src_pitch, (uint8_t *) y, y_pitch, (uint8_t *) u, u_pitch,
                                   (uint8_t *) v, v_pitch, cpl, total);
        } else {
                const long total = (long) ((width + 1) / 2) * ((height + 1) / 2);
                hipLaunchKernelGGL(uyvy_to_i420_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st,
                                   (const uint8_t *) src, src_pitch, (uint8_t *) y, y_pitch, (uint8_t *) u, u_pitch,
                                   (uint8_t *) v, v_pitch, width, height);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int ug_hip_v210_to_p010le(const void *src, int src_pitch, void *y, int y_pitch, void *uv, int uv_pitch, int width,
                          int height, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height)) return ug::refuse_size("ug_hip_v210_to_p010le");
        if (!src || !y || !uv || width <= 0 || height <= 0) return UG_HIP_EINVAL;
        if (!src_pitch) src_pitch = ug::linesize(UG_PF_V210, width);
        if (!ug::planes_ok(height, { src_pitch, y_pitch, uv_pitch })) return ug::refuse_size("ug_hip_v210_to_p010le");
        const int gpl = (width + 5) / 6;
        // to_planar.c:68-70 asserts a 4-byte aligned source and even output line sizes; a line must hold its own samples
        if ((src_pitch & 3) || src_pitch < 16 * gpl || (y_pitch & 1) || (uv_pitch & 1) || y_pitch < 2 * width || uv_pitch < 2 * width ||
            (3 & (uintptr_t) src) || (1 & (uintptr_t) y) || (1 & (uintptr_t) uv)) {
                ug::set_last_error_msg("ug_hip_v210_to_p010le: misaligned plane, odd output pitch, or a pitch shorter than the line");
                return UG_HIP_EINVAL;
        }
        if (width % 6 && height < 5) {
                // to_planar.c:142-148 copies the width % 6 margin from two lines (two chroma lines) above: with fewer than 5 lines
                // the reference reads in front of its output planes -- there is no defined result to reproduce
                ug::set_last_error_msg("ug_hip_v210_to_p010le: width % 6 != 0 needs at least 5 lines (the reference reads out of bounds)");
                return UG_HIP_EUNSUPP;
        }
        hipStream_t st = (hipStream_t) stream;
        if (width % 6 == 0 && height % 2 == 0 && height / 2 <= 65535 && !((src_pitch & 15) || (y_pitch & 3) || (uv_pitch & 3) || (15 & (uintptr_t) src) ||
                                                                          (3 & (uintptr_t) y) || (3 & (uintptr_t) uv))) {
                const int bx = gpl > 128 ? 256 : (gpl > 64 ? 128 : 64);
                hipLaunchKernelGGL(v210_to_p010le_kernel, dim3((unsigned) ((gpl + bx - 1) / bx), (unsigned) (height / 2)), dim3(bx), 0, st,
                                   (const uint8_t *) src, src_pitch, (uint8_t *) y, y_pitch, (uint8_t *) uv, uv_pitch, gpl);
        } else if (width < 6) {
                hipLaunchKernelGGL(v210_to_p010le_narrow_kernel, dim3(1), dim3(1), 0, st, (const uint8_t *) src, src_pitch, (uint16_t *) y,
                                   y_pitch / 2, (uint16_t *) uv, uv_pitch / 2, width, height);
        } else {
                const long total = (long) gpl * ((height + 1) / 2);
                hipLaunchKernelGGL(v210_to_p010le_any_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st,
                                   (const uint8_t *) src, src_pitch, (uint16_t *) y, y_pitch / 2, (uint16_t *) uv, uv_pitch / 2, width,
                                   height, gpl, total);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int ug_hip_yuv422_to_yuv444(const void *src, void *out, int pix_count, ug_hip_stream_t stream)
{
        if (pix_count < 0 || pix_count > ug::kMaxFrameBytes / 3) return ug::refuse_size("ug_hip_yuv422_to_yuv444"); // (3 output bytes per pixel)
        if (!src || !out || (pix_count & 3) || (7 & (uintptr_t) src) || (3 & (uintptr_t) out)) {
                return UG_HIP_EINVAL;
        }
        const int quads = pix_count / 4; // cuda_dxt.cu:766
        if (quads == 0) return UG_HIP_SUCCESS;
        hipLaunchKernelGGL(yuv422_to_yuv444_kernel, dim3((quads + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                           (const uint2 *) src, (uint32_t *) out, quads);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

} // extern "C"
