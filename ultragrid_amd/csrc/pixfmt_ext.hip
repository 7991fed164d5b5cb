// pixfmt_ext.hip -- the rest of decoders[] (src/pixfmt_conv.c:3041-3103): the line converters between the codecs
// outside the v210 / UYVY / RGB / RGBA core -- R10k, R12L, RG48, Y216, Y416, VUYA, DVS10 -- so that every pair get_decoder_from_to()
// answers is answered here too.  pixfmt.hip keeps the 17 pairs of P1-P6 with their vectorised fast paths; this file adds the other 44.
//
// Each kernel restates one reference function, one lane per iteration of its loop (a pixel, a pixel pair, a 6-pixel v210 group, an
// 8-pixel R12L group) and per line, with the reference's own iteration count for the `dst_len` of a line (vc_get_size(width, out)),
// so ragged line ends come out as they do there.  The hand-unrolled R12L functions are written here as what they compute -- a
// little-endian stream of 12-bit r, g, b -- and checked byte for byte against the compiled reference (tests/test_gpu_pixfmt_ext.py).
// Integer / byte work, HBM-bound.
#include <stdlib.h>
#include <string.h>

#include "ug_common.h"

namespace {

struct XArgs {
        const uint8_t *src;
        uint8_t *dst;
        long spitch, dpitch;
        int width, height;
        int L; // dst_len of a line
        int x0; // first loop iteration this launch covers (the iterations below it went through the vector kernel)
        int rs, gs, bs;
        uint32_t am;
        int c[14];
};
enum { Y_R, Y_G, Y_B, CB_R, CB_G, CB_B, CR_R, CR_G, CR_B, Y_SCALE, R_CR, G_CB, G_CR, B_CB };
constexpr int kBase = 14;

// A converter is written once, as the body of one iteration of the reference function's loop: XK(name) { XPRO(); if (x >= <its
// iteration count for a.L>) return; ... srow / drow ... }.  It is instantiated twice:
//   name            one lane per iteration, accesses as the body states them (any alignment, ragged line ends);
//   xvec_kernel<>   one lane per K consecutive iterations: the K * SB source bytes are fetched with 128-bit loads into a private
//                   array, the unchanged body runs K times on that array and on a private output array (constant indices after
//                   unrolling: the arrays live in registers, the byte accesses become bit-field operations), and the K * DB output
//                   bytes leave with 128-bit stores.  Same statements, same results; taken for the 16-byte-aligned interior of
//                   every line, the remaining iterations of a line go through `name` (XArgs::x0 = where they start).
// XR: source and destination of a conversion never overlap (the one in-place converter, vc_copylineToRGBA_inplace, is written
// without it); knowing that, the compiler merges the byte accesses of a body into dword / 128-bit ones
#define XR __restrict__
#define XK(name)                                                                                                                        \
        struct name##_body {                                                                                                            \
                static __device__ __forceinline__ void run(const XArgs &a, int x, const uint8_t *XR srow, uint8_t *XR drow);            \
        };                                                                                                                              \
        __global__ void name(const XArgs a)                                                                                             \
        {                                                                                                                               \
                XROW();                                                                                                                 \
                name##_body::run(a, x, srow, drow);                                                                                     \
        }                                                                                                                               \
        __device__ __forceinline__ void name##_body::run(const XArgs &a, const int x, const uint8_t *XR const srow, uint8_t *XR const drow)
#define XROW()                                                      \
        const int x = a.x0 + blockIdx.x * blockDim.x + threadIdx.x; \
        const int y = blockIdx.y * blockDim.y + threadIdx.y;        \
        if (y >= a.height) return;                                  \
        const uint8_t *const srow = a.src + (long) y * a.spitch;    \
        uint8_t *const drow = a.dst + (long) y * a.dpitch
#define XPRO() (void) srow, (void) drow
#define TO_Y(r, g, b) ((r) * a.c[Y_R] + (g) * a.c[Y_G] + (b) * a.c[Y_B])
#define TO_CB(r, g, b) ((r) * a.c[CB_R] + (g) * a.c[CB_G] + (b) * a.c[CB_B])
#define TO_CR(r, g, b) ((r) * a.c[CR_R] + (g) * a.c[CR_G] + (b) * a.c[CR_B])
#define TO_R(ys, u, v) ((ys) + (v) * a.c[R_CR])
#define TO_G(ys, u, v) ((ys) + (u) * a.c[G_CB] + (v) * a.c[G_CR])
#define TO_B(ys, u, v) ((ys) + (u) * a.c[B_CB])

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clamp_full(int v, int depth) { return clampi(v, 1 << (depth - 8), (255 << (depth - 8)) - 1); }

// R10k pixel: 10-bit big-endian R G B in 4 bytes (the bit-field struct of pixfmt_conv.c:214-224)
__device__ __forceinline__ void r10k_get(const uint8_t *s, uint32_t &r, uint32_t &g, uint32_t &b)
{
        r = s[0] << 2 | s[1] >> 6;
        g = (s[1] & 0x3fu) << 4 | s[2] >> 4;
        b = (s[2] & 0x0fu) << 6 | s[3] >> 2;
}

// 36 bytes of R12L <-> 8 x (r, g, b) of 12 bits, little-endian bit stream
__device__ __forceinline__ void r12l_get(const uint8_t *s, uint32_t (&v)[24])
{
        // value i = bits [12 i, 12 i + 12) of the 36-byte group (v[2k] = b0 | (b1 & 0xf) << 8, v[2k+1] = b1 >> 4 | b2 << 4 over bytes 3k..3k+2).
        // The group is fetched as nine words in one go: left as 36 byte loads, the compiler merges them or not depending on
        // how the body was inlined (24 two- and one-byte loads after the refactoring to body functions: 0.68 -> 0.34 of 8 TB/s).
        uint32_t w[9];
        __builtin_memcpy(w, s, 36);
#pragma unroll
        for (int i = 0; i < 24; i++) {
                const int bit = 12 * i, j = bit >> 5, o = bit & 31;
                uint32_t x = w[j] >> o;
                if (o > 20) x |= w[j + 1] << (32 - o);
                v[i] = x & 0xfffu;
        }
}
__device__ __forceinline__ void r12l_put(uint8_t *d, const uint32_t (&v)[24], int nbytes)
{
#pragma unroll
        for (int k = 0; k < 12; k++) {
                const uint32_t e = v[2 * k], o = v[2 * k + 1];
                if (3 * k < nbytes) d[3 * k] = (uint8_t) e;
                if (3 * k + 1 < nbytes) d[3 * k + 1] = (uint8_t) ((o & 0xfu) << 4 | e >> 8);
                if (3 * k + 2 < nbytes) d[3 * k + 2] = (uint8_t) (o >> 4);
        }
}

// ---- R10k sources ------------------------------------------------------------------------------------------------------------------
XK(k_r10k_to_rgba) // vc_copyliner10k :211-276: len / 4 pixels, top 8 bits of each component
{
        XPRO();
        if (x >= a.L / 4) return;
        uint32_t r, g, b;
        r10k_get(srow + 4 * x, r, g, b);
        ((uint32_t *) drow)[x] = a.am | (r >> 2) << a.rs | (g >> 2) << a.gs | (b >> 2) << a.bs;
}
XK(k_r10k_to_rg48) // :279-295: while (dstlen > 0) -> ceil(L / 6) pixels; each 10-bit component moves to the top of a 16-bit one
{
        XPRO();
        if (x >= (a.L + 5) / 6) return;
        uint32_t r, g, b;
        r10k_get(srow + 4 * x, r, g, b);
        uint8_t *d = drow + 6 * x; // byte stores: RG48 lines of an odd pixel count need not be 2-aligned
        const uint32_t c[3] = { r << 6, g << 6, b << 6 };
#pragma unroll
        for (int k = 0; k < 3; k++) d[2 * k] = (uint8_t) c[k], d[2 * k + 1] = (uint8_t) (c[k] >> 8);
}
XK(k_r10k_to_y416) // :297-329, 16-bit coefficients on components scaled to 16 bits
{
        XPRO();
        if (x >= (a.L + 7) / 8) return;
        uint32_t r10, g10, b10;
        r10k_get(srow + 4 * x, r10, g10, b10);
        const int r = r10 << 6, g = g10 << 6, b = b10 << 6;
        uint16_t *d = (uint16_t *) drow + 4 * x;
        d[0] = (uint16_t) ((TO_CB(r, g, b) >> kBase) + (1 << 15));
        d[1] = (uint16_t) ((TO_Y(r, g, b) >> kBase) + (1 << 12));
        d[2] = (uint16_t) ((TO_CR(r, g, b) >> kBase) + (1 << 15));
        d[3] = 0xFFFF;
}
XK(k_r10k_to_rgb) // :331-341: the top 8 bits of each component
{
        XPRO();
        if (x >= (a.L + 2) / 3) return;
        uint32_t r, g, b;
        r10k_get(srow + 4 * x, r, g, b);
        uint8_t *d = drow + 3 * x;
        d[0] = (uint8_t) (r >> 2), d[1] = (uint8_t) (g >> 2), d[2] = (uint8_t) (b >> 2);
}

// vc_copylineToUYVY on two 8-bit RGB pixels (:1008-1053): y = (Y >> 14) + 16, u = ((cb1 + cb2) / 2 >> 14) + 128
__device__ __forceinline__ uint32_t rgb_pair_to_uyvy(const XArgs &a, int r1, int g1, int b1, int r2, int g2, int b2)
{
        const int y1 = (TO_Y(r1, g1, b1) >> kBase) + 16, y2 = (TO_Y(r2, g2, b2) >> kBase) + 16;
        const int u = (((TO_CB(r1, g1, b1) + TO_CB(r2, g2, b2)) / 2) >> kBase) + 128, v = (((TO_CR(r1, g1, b1) + TO_CR(r2, g2, b2)) / 2) >> kBase) + 128;
        return (uint32_t) (u & 0xff) | (uint32_t) (y1 & 0xff) << 8 | (uint32_t) (v & 0xff) << 16 | (uint32_t) (y2 & 0xff) << 24;
}
XK(k_r10k_to_uyvy) // vc_copylineR10ktoUYVY :2320-2340: top 8 bits, then vc_copylineRGBtoUYVY on the pair
{
        XPRO();
        if (x >= (a.L + 3) / 4) return;
        uint32_t r1, g1, b1, r2, g2, b2;
        r10k_get(srow + 8 * x, r1, g1, b1);
        r10k_get(srow + 8 * x + 4, r2, g2, b2);
        ((uint32_t *) drow)[x] = rgb_pair_to_uyvy(a, r1 >> 2, g1 >> 2, b1 >> 2, r2 >> 2, g2 >> 2, b2 >> 2);
}

// ---- R12L sources (one lane per group of 8 pixels = 36 source bytes) ---------------------------------------------------------------
XK(k_r12l_to_rgb) // vc_copylineR12LtoRGB :353-423: whole groups only (x <= dstlen - 24)
{
        XPRO();
        if (x >= a.L / 24) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint8_t *d = drow + 24 * x;
#pragma unroll
        for (int i = 0; i < 24; i++) d[i] = (uint8_t) (v[i] >> 4);
}
XK(k_r12l_to_rgba) // vc_copylineR12L :438-517: every started group, the last one cut at dstlen
{
        XPRO();
        if (x >= (a.L + 31) / 32) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint32_t *d = (uint32_t *) drow + 8 * x;
        const int n = min(8, (a.L - 32 * x) / 4);
        const int rem = a.L - 32 * x - 4 * n; // memcpy(orig_d, tmpbuf, dstlen - x) may end inside a pixel
#pragma unroll
        for (int i = 0; i < 8; i++) { // static indices only: v[] stays in registers
                const uint32_t w = a.am | (v[3 * i] >> 4) << a.rs | (v[3 * i + 1] >> 4) << a.gs | (v[3 * i + 2] >> 4) << a.bs;
                if (i < n) {
                        d[i] = w;
                } else if (i == n) {
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                                if (k < rem) ((uint8_t *) (d + i))[k] = (uint8_t) (w >> (8 * k));
                        }
                }
        }
}
XK(k_r12l_to_rg48) // :1371-1476: whole groups, then the head of one more
{
        XPRO();
        if (x >= (a.L + 47) / 48) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint8_t *d = drow + 48 * x;
        const int nb = min(48, a.L - 48 * x);
#pragma unroll
        for (int i = 0; i < 24; i++) {
                const uint32_t s16 = v[i] << 4;
                if (2 * i < nb) d[2 * i] = (uint8_t) s16;
                if (2 * i + 1 < nb) d[2 * i + 1] = (uint8_t) (s16 >> 8);
        }
}
XK(k_r12l_to_r10k) // :1640-1699: whole groups only
{
        XPRO();
        if (x >= a.L / 32) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint8_t *d = drow + 32 * x;
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const uint32_t r = v[3 * i], g = v[3 * i + 1], b = v[3 * i + 2]; // 12 bit; the low two bits of b fall into the padding
                d[4 * i] = (uint8_t) (r >> 4);
                d[4 * i + 1] = (uint8_t) ((r & 0xC) << 4 | g >> 6);
                d[4 * i + 2] = (uint8_t) ((g & 0x3C) << 2 | b >> 8);
                d[4 * i + 3] = (uint8_t) b;
        }
        d[7] = (uint8_t) ((v[5] & 0xf0) | (v[3] & 0xf)); // pixel 1: `src[8 + 0] << 4 | (src[4 + 0] & 0xF0) >> 4` takes r1's low nibble (:1657)
}
XK(k_r12l_to_y416) // :1478-1542: every started group whole
{
        XPRO();
        if (x >= (a.L + 63) / 64) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint16_t *d = (uint16_t *) drow + 32 * x;
        // the reference writes the last group whole, into the head of the next line, whose own conversion then overwrites it: inside
        // the frame the result is the same as stopping at dst_len (lines are converted concurrently here)
        const int n = min(8, (a.L - 64 * x) / 8);
#pragma unroll
        for (int i = 0; i < 8; i++) {
                if (i >= n) break;
                const int r = v[3 * i] << 4, g = v[3 * i + 1] << 4, b = v[3 * i + 2] << 4;
                d[4 * i] = (uint16_t) ((TO_CB(r, g, b) >> kBase) + (1 << 15));
                d[4 * i + 1] = (uint16_t) ((TO_Y(r, g, b) >> kBase) + (1 << 12));
                d[4 * i + 2] = (uint16_t) ((TO_CR(r, g, b) >> kBase) + (1 << 15));
                d[4 * i + 3] = 0xFFFF;
        }
}
XK(k_r12l_to_uyvy) // :1544-1638: 16-bit-scaled components, 8-bit coefficients, >> (14 + 8)
{
        XPRO();
        if (x >= (a.L + 15) / 16) return;
        uint32_t v[24];
        r12l_get(srow + 36 * x, v);
        uint32_t *d = (uint32_t *) drow + 4 * x;
        const int n = min(4, (a.L - 16 * x) / 4); // see k_r12l_to_y416
#pragma unroll
        for (int i = 0; i < 4; i++) {
                if (i >= n) break;
                const int r1 = v[6 * i] << 4, g1 = v[6 * i + 1] << 4, b1 = v[6 * i + 2] << 4, r2 = v[6 * i + 3] << 4, g2 = v[6 * i + 4] << 4, b2 = v[6 * i + 5] << 4;
                const int u = ((TO_CB(r1, g1, b1) + TO_CB(r2, g2, b2)) >> (kBase + 9)) + 128, vv = ((TO_CR(r1, g1, b1) + TO_CR(r2, g2, b2)) >> (kBase + 9)) + 128;
                const int y1 = (TO_Y(r1, g1, b1) >> (kBase + 8)) + 16, y2 = (TO_Y(r2, g2, b2) >> (kBase + 8)) + 16;
                d[i] = (uint32_t) (u & 0xff) | (uint32_t) (y1 & 0xff) << 8 | (uint32_t) (vv & 0xff) << 16 | (uint32_t) (y2 & 0xff) << 24;
        }
}

// ---- -> R12L -------------------------------------------------------------------------------------------------------------------------
template <int BPP>
struct k_rgb_to_r12l_body {
        static __device__ __forceinline__ void run(const XArgs &a, int x, const uint8_t *XR srow, uint8_t *XR drow);
};
template <int BPP>
__global__ void k_rgb_to_r12l(const XArgs a)
{
        XROW();
        k_rgb_to_r12l_body<BPP>::run(a, x, srow, drow);
}
template <int BPP>
__device__ __forceinline__ void k_rgb_to_r12l_body<BPP>::run(const XArgs &a, const int x, const uint8_t *XR const srow, uint8_t *XR const drow) // vc_copylineRGB_AtoR12L :1263-1322: whole groups only, component << 4
{
        XPRO();
        if (x >= a.L / 36) return;
        uint32_t v[24];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const uint8_t *s = srow + (long) BPP * (8 * x + i);
                v[3 * i] = s[0] << 4, v[3 * i + 1] = s[1] << 4, v[3 * i + 2] = s[2] << 4;
        }
        r12l_put(drow + 36 * x, v, 36);
}
XK(k_rg48_to_r12l) // :1701-1826: whole groups only, component >> 4
{
        XPRO();
        if (x >= a.L / 36) return;
        const uint16_t *s = (const uint16_t *) srow + 24 * x;
        uint32_t v[24];
#pragma unroll
        for (int i = 0; i < 24; i++) v[i] = s[i] >> 4;
        r12l_put(drow + 36 * x, v, 36);
}
XK(k_y416_to_r12l) // :1828-1915: every started group whole
{
        XPRO();
        if (x >= (a.L + 35) / 36) return;
        const uint16_t *s = (const uint16_t *) srow + 32 * x;
        uint32_t v[24];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const int u = s[4 * i] - (1 << 15), ys = a.c[Y_SCALE] * (s[4 * i + 1] - (1 << 12)), vv = s[4 * i + 2] - (1 << 15);
                v[3 * i] = clamp_full(TO_R(ys, u, vv) >> (kBase + 4), 12);
                v[3 * i + 1] = clamp_full(TO_G(ys, u, vv) >> (kBase + 4), 12);
                v[3 * i + 2] = clamp_full(TO_B(ys, u, vv) >> (kBase + 4), 12);
        }
        r12l_put(drow + 36 * x, v, 36);
}

// ---- RGB / RGBA / UYVY -> 16-bit --------------------------------------------------------------------------------------------------
XK(k_rgba_to_rg48) // :1336-1351
{
        XPRO();
        if (x >= a.L / 6) return;
        const uint8_t *s = srow + 4 * x;
        uint16_t *d = (uint16_t *) drow + 3 * x;
        d[0] = s[0] << 8, d[1] = s[1] << 8, d[2] = s[2] << 8;
}
XK(k_rgb_to_rg48) // :1353-1363: one lane per component
{
        XPRO();
        if (x >= a.L / 2) return;
        ((uint16_t *) drow)[x] = srow[x] << 8;
}
template <bool YUYV, bool RGB16>
struct k_yuv422_to_rgb_body {
        static __device__ __forceinline__ void run(const XArgs &a, int x, const uint8_t *XR srow, uint8_t *XR drow);
};
template <bool YUYV, bool RGB16>
__global__ void k_yuv422_to_rgb(const XArgs a)
{
        XROW();
        k_yuv422_to_rgb_body<YUYV, RGB16>::run(a, x, srow, drow);
}
template <bool YUYV, bool RGB16>
__device__ __forceinline__ void k_yuv422_to_rgb_body<YUYV, RGB16>::run(const XArgs &a, const int x, const uint8_t *XR const srow, uint8_t *XR const drow) // copylineYUVtoRGB :1065-1094: vc_copylineUYVYtoRG48 (rgb16), vc_copylineYUYVtoRGB; clamp 0..255
{
        XPRO();
        constexpr int kOut = RGB16 ? 12 : 6;
        if (x >= a.L / kOut) return;
        const uint8_t *s = srow + 4 * x;
        const int y1 = a.c[Y_SCALE] * (s[YUYV ? 0 : 1] - 16), y2 = a.c[Y_SCALE] * (s[YUYV ? 2 : 3] - 16), u = s[YUYV ? 1 : 0] - 128, v = s[YUYV ? 3 : 2] - 128;
        const int o[6] = { clampi(TO_R(y1, u, v) >> kBase, 0, 255), clampi(TO_G(y1, u, v) >> kBase, 0, 255), clampi(TO_B(y1, u, v) >> kBase, 0, 255),
                           clampi(TO_R(y2, u, v) >> kBase, 0, 255), clampi(TO_G(y2, u, v) >> kBase, 0, 255), clampi(TO_B(y2, u, v) >> kBase, 0, 255) };
        if (RGB16) {
                uint16_t *d = (uint16_t *) drow + 6 * x;
#pragma unroll
                for (int i = 0; i < 6; i++) d[i] = (uint16_t) (o[i] << 8);
        } else {
                uint8_t *d = drow + 6 * x;
#pragma unroll
                for (int i = 0; i < 6; i++) d[i] = (uint8_t) o[i];
        }
}

// ---- RG48 sources ---------------------------------------------------------------------------------------------------------------------
XK(k_rg48_to_r10k) // :2008-2029: top 10 bits of each component, R10k byte order, padding bits 11
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint16_t *s = (const uint16_t *) srow + 3 * x;
        const uint32_t r = s[0] >> 6, g = s[1] >> 6, b = s[2] >> 6;
        const uint32_t b0 = r >> 2, b1 = (r & 3u) << 6 | g >> 4, b2 = (g & 0xfu) << 4 | b >> 6, b3 = (b & 0x3fu) << 2 | 3u;
        ((uint32_t *) drow)[x] = b0 | b1 << 8 | b2 << 16 | b3 << 24;
}
XK(k_rg48_to_rgb) // :2031-2043
{
        XPRO();
        if (x >= a.L / 3) return;
        const uint8_t *s = srow + 6 * x;
        uint8_t *d = drow + 3 * x;
        d[0] = s[1], d[1] = s[3], d[2] = s[5];
}
XK(k_rg48_to_rgba) // :2045-2059
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 6 * x;
        ((uint32_t *) drow)[x] = a.am | (uint32_t) s[1] << a.rs | (uint32_t) s[3] << a.gs | (uint32_t) s[5] << a.bs;
}
XK(k_rg48_to_v210) // :2354-2408: whole 16-byte groups of 6 pixels
{
        XPRO();
        if (x >= a.L / 16) return;
        const uint16_t *s = (const uint16_t *) srow + 18 * x;
        constexpr int off = kBase + 6;
        int Y[6], U[3], V[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
                const int r1 = s[6 * i], g1 = s[6 * i + 1], b1 = s[6 * i + 2], r2 = s[6 * i + 3], g2 = s[6 * i + 4], b2 = s[6 * i + 5];
                Y[2 * i] = (TO_Y(r1, g1, b1) >> off) + (1 << 6);
                Y[2 * i + 1] = (TO_Y(r2, g2, b2) >> off) + (1 << 6);
                U[i] = ((TO_CB(r1, g1, b1) >> off) + (TO_CB(r2, g2, b2) >> off)) / 2 + (1 << 9);
                V[i] = ((TO_CR(r1, g1, b1) >> off) + (TO_CR(r2, g2, b2) >> off)) / 2 + (1 << 9);
        }
        uint32_t *d = (uint32_t *) drow + 4 * x; // unmasked ORs, as in the reference
        d[0] = (uint32_t) (U[0] | Y[0] << 10 | V[0] << 20);
        d[1] = (uint32_t) (Y[1] | U[1] << 10 | Y[2] << 20);
        d[2] = (uint32_t) (V[1] | Y[3] << 10 | U[2] << 20);
        d[3] = (uint32_t) (Y[4] | V[2] << 10 | Y[5] << 20);
}
XK(k_rg48_to_y216) // :2410-2449
{
        XPRO();
        if (x >= (a.L + 7) / 8) return;
        const uint16_t *s = (const uint16_t *) srow + 6 * x;
        uint16_t *d = (uint16_t *) drow + 4 * x;
        const int r1 = s[0], g1 = s[1], b1 = s[2], r2 = s[3], g2 = s[4], b2 = s[5];
        d[0] = (uint16_t) ((TO_Y(r1, g1, b1) >> kBase) + (1 << 12));
        d[1] = (uint16_t) ((((TO_CB(r1, g1, b1) >> kBase) + (TO_CB(r2, g2, b2) >> kBase)) / 2) + (1 << 15));
        d[2] = (uint16_t) ((TO_Y(r2, g2, b2) >> kBase) + (1 << 12));
        d[3] = (uint16_t) ((((TO_CR(r1, g1, b1) >> kBase) + (TO_CR(r2, g2, b2) >> kBase)) / 2) + (1 << 15));
}
XK(k_rg48_to_y416) // :2451-2483
{
        XPRO();
        if (x >= (a.L + 7) / 8) return;
        const uint16_t *s = (const uint16_t *) srow + 3 * x;
        uint16_t *d = (uint16_t *) drow + 4 * x;
        const int r = s[0], g = s[1], b = s[2];
        d[0] = (uint16_t) ((TO_CB(r, g, b) >> kBase) + (1 << 15));
        d[1] = (uint16_t) ((TO_Y(r, g, b) >> kBase) + (1 << 12));
        d[2] = (uint16_t) ((TO_CR(r, g, b) >> kBase) + (1 << 15));
        d[3] = 0xFFFF;
}

// ---- Y416 sources (U Y V A, 16 bit) ------------------------------------------------------------------------------------------------
template <int OUT> // 0 RG48 (:2485-2518), 1 R10k (:1917-1946), 2 RGB (:1948-1976), 3 RGBA (:1978-2006)
struct k_y416_to_rgb_body {
        static __device__ __forceinline__ void run(const XArgs &a, int x, const uint8_t *XR srow, uint8_t *XR drow);
};
template <int OUT> // 0 RG48 (:2485-2518), 1 R10k (:1917-1946), 2 RGB (:1948-1976), 3 RGBA (:1978-2006)
__global__ void k_y416_to_rgb(const XArgs a)
{
        XROW();
        k_y416_to_rgb_body<OUT>::run(a, x, srow, drow);
}
template <int OUT> // 0 RG48 (:2485-2518), 1 R10k (:1917-1946), 2 RGB (:1948-1976), 3 RGBA (:1978-2006)
__device__ __forceinline__ void k_y416_to_rgb_body<OUT>::run(const XArgs &a, const int x, const uint8_t *XR const srow, uint8_t *XR const drow)
{
        XPRO();
        constexpr int kBytes = OUT == 0 ? 6 : (OUT == 2 ? 3 : 4);
        if (x >= (a.L + kBytes - 1) / kBytes) return;
        const uint16_t *s = (const uint16_t *) srow + 4 * x;
        const int u = s[0] - (1 << 15), ys = a.c[Y_SCALE] * (s[1] - (1 << 12)), v = s[2] - (1 << 15);
        constexpr int sh = kBase + (OUT == 0 ? 0 : (OUT == 1 ? 6 : 8)), depth = OUT == 0 ? 16 : (OUT == 1 ? 10 : 8);
        const uint32_t r = clamp_full(TO_R(ys, u, v) >> sh, depth), g = clamp_full(TO_G(ys, u, v) >> sh, depth), b = clamp_full(TO_B(ys, u, v) >> sh, depth);
        if (OUT == 0) {
                uint16_t *d = (uint16_t *) drow + 3 * x;
                d[0] = (uint16_t) r, d[1] = (uint16_t) g, d[2] = (uint16_t) b;
        } else if (OUT == 1) {
                uint8_t *d = drow + 4 * x;
                d[0] = (uint8_t) (r >> 2), d[1] = (uint8_t) ((r & 0x3U) << 6U | g >> 4U), d[2] = (uint8_t) ((g & 0xFU) << 4U | b >> 6U), d[3] = (uint8_t) ((b & 0x3FU) << 2U);
        } else if (OUT == 2) {
                uint8_t *d = drow + 3 * x;
                d[0] = (uint8_t) r, d[1] = (uint8_t) g, d[2] = (uint8_t) b;
        } else {
                ((uint32_t *) drow)[x] = a.am | r << a.rs | g << a.gs | b << a.bs;
        }
}
XK(k_y416_to_uyvy) // :2745-2759: high bytes, (a + b) / 2 chroma
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 16 * x;
        ((uint32_t *) drow)[x] = (uint32_t) ((s[1] + s[9]) / 2) | (uint32_t) s[3] << 8 | (uint32_t) ((s[5] + s[13]) / 2) << 16 | (uint32_t) s[11] << 24;
}
XK(k_y416_to_v210) // :3004-3029
{
        XPRO();
        if (x >= a.L / 16) return;
        const uint16_t *s = (const uint16_t *) srow + 24 * x;
        uint32_t *d = (uint32_t *) drow + 4 * x;
        uint32_t u[3], v[3], Y[6];
#pragma unroll
        for (int i = 0; i < 3; i++) {
                u[i] = (uint16_t) ((s[8 * i] + s[8 * i + 4]) / 2), v[i] = (uint16_t) ((s[8 * i + 2] + s[8 * i + 6]) / 2);
                Y[2 * i] = s[8 * i + 1], Y[2 * i + 1] = s[8 * i + 5];
        }
        d[0] = u[0] >> 6U | Y[0] >> 6U << 10U | v[0] >> 6U << 20U;
        d[1] = Y[1] >> 6U | u[1] >> 6U << 10U | Y[2] >> 6U << 20U;
        d[2] = v[1] >> 6U | Y[3] >> 6U << 10U | u[2] >> 6U << 20U;
        d[3] = Y[4] >> 6U | v[2] >> 6U << 10U | Y[5] >> 6U << 20U;
}

// ---- 8-bit packed YUV <-> 16-bit packed YUV, VUYA -------------------------------------------------------------------------------------
XK(k_rgba_to_vuya) // :2281-2309
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 4 * x;
        const int r = s[0], g = s[1], b = s[2];
        ((uint32_t *) drow)[x] = (uint32_t) (((TO_CR(r, g, b) >> kBase) + 128) & 0xff) | (uint32_t) (((TO_CB(r, g, b) >> kBase) + 128) & 0xff) << 8 |
                                 (uint32_t) (((TO_Y(r, g, b) >> kBase) + 16) & 0xff) << 16 | (uint32_t) s[3] << 24;
}
XK(k_rgba_to_r10k) // :2538-2579 (bit-field struct: p3 = 3, the other padding bits 0)
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 4 * x;
        const uint32_t r = s[0], g = s[1], b = s[2];
        ((uint32_t *) drow)[x] = r | (g >> 2) << 8 | (b >> 4) << 16 | (g & 3u) << 22 | 3u << 24 | (b & 0xfu) << 28;
}
XK(k_uyvy_to_y216) // :2609-2627
{
        XPRO();
        if (x >= a.L / 8) return;
        const uint8_t *s = srow + 4 * x;
        uint16_t *d = (uint16_t *) drow + 4 * x;
        d[0] = s[1] << 8, d[1] = s[0] << 8, d[2] = s[3] << 8, d[3] = s[2] << 8;
}
XK(k_uyvy_to_y416) // :2629-2664: pairs while dst_len >= 12 (the second pixel is written whole even when only 12 bytes remain), then one more pixel
{
        XPRO();
        int pairs = 0, len = a.L;
        while (len >= 12) pairs++, len -= 16;
        const bool tail = len >= 8;
        if (x >= pairs + (tail ? 1 : 0)) return;
        const uint8_t *s = srow + 4 * x;
        uint16_t *d = (uint16_t *) drow + 8 * x;
        d[0] = s[0] << 8, d[1] = s[1] << 8, d[2] = s[2] << 8, d[3] = 0xFFFF;
        if (x < pairs) d[4] = s[0] << 8, d[5] = s[3] << 8, d[6] = s[2] << 8, d[7] = 0xFFFF;
}
XK(k_vuya_to_y416) // :2668-2687
{
        XPRO();
        if (x >= a.L / 8) return;
        const uint8_t *s = srow + 4 * x;
        uint16_t *d = (uint16_t *) drow + 4 * x;
        d[0] = s[1] << 8, d[1] = s[2] << 8, d[2] = s[0] << 8, d[3] = s[3] << 8;
}
XK(k_vuya_to_uyvy) // :2689-2704 (Y1 is taken from the second pixel's alpha byte, src[7], as written there)
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 8 * x;
        ((uint32_t *) drow)[x] = (uint32_t) ((s[1] + s[5]) / 2) | (uint32_t) s[2] << 8 | (uint32_t) ((s[0] + s[4]) / 2) << 16 | (uint32_t) s[7] << 24;
}
XK(k_vuya_to_rgb) // :2706-2727
{
        XPRO();
        if (x >= (a.L + 2) / 3) return;
        const uint8_t *s = srow + 4 * x;
        const int v = s[0] - 128, u = s[1] - 128, ys = a.c[Y_SCALE] * (s[2] - 16);
        uint8_t *d = drow + 3 * x;
        d[0] = (uint8_t) clamp_full(TO_R(ys, u, v) >> kBase, 8), d[1] = (uint8_t) clamp_full(TO_G(ys, u, v) >> kBase, 8), d[2] = (uint8_t) clamp_full(TO_B(ys, u, v) >> kBase, 8);
}
XK(k_y216_to_uyvy) // :2729-2743
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint8_t *s = srow + 8 * x;
        ((uint32_t *) drow)[x] = (uint32_t) s[3] | (uint32_t) s[1] << 8 | (uint32_t) s[7] << 16 | (uint32_t) s[5] << 24;
}
XK(k_y216_to_v210) // :2761-2790: (dst_len + 15) / 16 groups; Y216 = Y0 Cb Y1 Cr per pixel pair
{
        XPRO();
        if (x >= (a.L + 15) / 16) return;
        const uint16_t *s = (const uint16_t *) srow + 12 * x;
        uint32_t Y[6], U[3], V[3];
#pragma unroll
        for (int k = 0; k < 3; k++) Y[2 * k] = s[4 * k] >> 6, U[k] = s[4 * k + 1] >> 6, Y[2 * k + 1] = s[4 * k + 2] >> 6, V[k] = s[4 * k + 3] >> 6;
        uint32_t *d = (uint32_t *) drow + 4 * x;
        d[0] = U[0] | Y[0] << 10 | V[0] << 20, d[1] = Y[1] | U[1] << 10 | Y[2] << 20, d[2] = V[1] | Y[3] << 10 | U[2] << 20, d[3] = Y[4] | V[2] << 10 | Y[5] << 20;
}
template <bool Y416>
struct k_v210_to_y2xx_body {
        static __device__ __forceinline__ void run(const XArgs &a, int x, const uint8_t *XR srow, uint8_t *XR drow);
};
template <bool Y416>
__global__ void k_v210_to_y2xx(const XArgs a)
{
        XROW();
        k_v210_to_y2xx_body<Y416>::run(a, x, srow, drow);
}
template <bool Y416>
__device__ __forceinline__ void k_v210_to_y2xx_body<Y416>::run(const XArgs &a, const int x, const uint8_t *XR const srow, uint8_t *XR const drow) // vc_copylineV210toY216 :2792-2832 (dst_len / 24 groups), vc_copylineV210toY416 :2834-2882 (dst_len / 48)
{
        XPRO();
        if (x >= a.L / (Y416 ? 48 : 24)) return;
        const uint32_t *s = (const uint32_t *) srow + 4 * x;
        const uint32_t w0 = s[0], w1 = s[1], w2 = s[2], w3 = s[3];
        const uint32_t Y[6] = { (w0 >> 10) & 0x3ff, w1 & 0x3ff, (w1 >> 20) & 0x3ff, (w2 >> 10) & 0x3ff, w3 & 0x3ff, (w3 >> 20) & 0x3ff };
        const uint32_t U[3] = { w0 & 0x3ff, (w1 >> 10) & 0x3ff, (w2 >> 20) & 0x3ff }, V[3] = { (w0 >> 20) & 0x3ff, w2 & 0x3ff, (w3 >> 10) & 0x3ff };
        if (Y416) {
                uint16_t *d = (uint16_t *) drow + 24 * x;
#pragma unroll
                for (int i = 0; i < 6; i++) d[4 * i] = U[i / 2] << 6, d[4 * i + 1] = Y[i] << 6, d[4 * i + 2] = V[i / 2] << 6, d[4 * i + 3] = 0xFFFF;
        } else {
                uint16_t *d = (uint16_t *) drow + 12 * x;
#pragma unroll
                for (int i = 0; i < 3; i++) d[4 * i] = Y[2 * i] << 6, d[4 * i + 1] = U[i] << 6, d[4 * i + 2] = Y[2 * i + 1] << 6, d[4 * i + 3] = V[i] << 6;
        }
}

// ---- DVS10 ------------------------------------------------------------------------------------------------------------------------------
XK(k_dvs10_to_v210) // :595-617: a DVS10 word carries the top 8 bits of three components in bytes 0-2 and their low 2 bits in byte 3
{
        XPRO();
        if (x >= a.L / 4) return;
        const uint32_t in = ((const uint32_t *) srow)[x], low = in >> 24;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) out |= ((((in >> (8 * k)) & 0xffu) << 2) | ((low >> (2 * k)) & 3u)) << (10 * k);
        ((uint32_t *) drow)[x] = out;
}
XK(k_dvs10_to_uyvy) // vc_copylineDVS10 :690-721: src_len = dst_len / 1.5, one iteration per 16 of it, each moving 32 source bytes to 24
{                   // (three of every four bytes)
        XPRO();
        if (x >= (int) (a.L / 1.5) / 16 * 8) return;
        const uint8_t *s = srow + 4 * x;
        uint8_t *d = drow + 3 * x;
        d[0] = s[0], d[1] = s[1], d[2] = s[2];
}

// ---- exported line converters that are not in decoders[] (pixfmt_conv.h:93-101) ----------------------------------------------------------
XK(k_uyvy_to_grayscale) // vc_copylineUYVYtoGrayscale :927-938: the two luma bytes of every UYVY word
{
        XPRO();
        if (x >= a.L / 2) return;
        const uint32_t s = ((const uint32_t *) srow)[x];
        drow[2 * x] = (uint8_t) (s >> 8), drow[2 * x + 1] = (uint8_t) (s >> 24);
}
XK(k_rgba_to_rgb_shift) // vc_copylineRGBAtoRGBwithShift :769-807 (a.rs/gs/bs = SOURCE shifts): vc_copylineABGRtoRGB (24,16,8), vc_copylineBGRAtoRGB (16,8,0)
{
        XPRO();
        if (x >= a.L / 3) return;
        const uint32_t in = ((const uint32_t *) srow)[x];
        uint8_t *d = drow + 3 * x;
        d[0] = (uint8_t) (in >> a.rs), d[1] = (uint8_t) (in >> a.gs), d[2] = (uint8_t) (in >> a.bs);
}
__global__ void k_to_rgba_inplace(const XArgs a) // vc_copylineToRGBA_inplace :907-921 (source shifts; alpha byte 0); dst may BE src: no XR here
{
        XROW();
        if (x >= a.L / 4) return;
        const uint32_t in = ((const uint32_t *) srow)[x];
        ((uint32_t *) drow)[x] = ((in >> a.rs) & 0xff) | ((in >> a.gs) & 0xff) << 8 | ((in >> a.bs) & 0xff) << 16;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// K iterations of a converter per lane, through private arrays (see XK above).  nvec = vector units per line.
// A lane's unit is K * SB contiguous source bytes and K * DB contiguous output bytes.  When that is one 16-byte word the lanes of a
// wave access consecutive words and nothing else is needed.  When it is several, per-lane accesses would be strided (every load
// instruction touching 64 different cache lines and using 16 bytes of each -- measured: RG48->RGB fell from 0.31 to 0.19 of 8 TB/s that
// way), so the wave moves its 64 units as ONE contiguous region: word c of the region is handled by lane c % 64, and the words change
// hands through LDS (rows of an odd number of 16-byte words: conflict-free on both sides).
template <class Body, int SB, int DB, int K>
__global__ __launch_bounds__(256) void xvec_kernel(const XArgs a, int nvec)
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
        XArgs b = a;
        b.L = a.L - u * (K * DB); // the line as this unit sees it: its own iterations are 0 .. K - 1 of what is left
#pragma unroll
        for (int k = 0; k < K; k++) Body::run(b, k, ls, ld);
        Out::store((uint4 *) (a.dst + (long) y * a.dpitch + (long) u0 * (K * DB)), ld, lds, lane, units);
}
template <class Body, int SB, int DB, int K>
void launch_xvec(const XArgs &a, int nvec, hipStream_t st)
{
        const dim3 block(64, 4, 1), grid((unsigned) ((nvec + 63) / 64), (unsigned) ((a.height + 3) / 4), 1);
        hipLaunchKernelGGL((xvec_kernel<Body, SB, DB, K>), grid, block, 0, st, a, nvec);
}

// ---- R12L by the pixel -------------------------------------------------------------------------------------------------------------------
// R12L is a little-endian stream of 12-bit samples, 36 bits per pixel, 8 pixels = 9 words.  The group-per-lane bodies above hold 24 samples
// (and up to 16 output words) per lane; for the whole groups of a line these kernels go one lane per pixel instead: pixel p sits at bit
// 36 p, i.e. in the two words from word 9p/8 on, shifted by 4 (p % 8) bits -- two aligned loads, one 64-bit shift, one coalesced store.
// The arithmetic per pixel is that of the bodies above (k_r12l_to_rgba / _rg48 / _y416, k_rgb_to_r12l / k_rg48_to_r12l / k_y416_to_r12l).
__device__ __forceinline__ void r12l_pixel(const uint8_t *XR srow, int p, uint32_t &r, uint32_t &g, uint32_t &b)
{
        const uint32_t *w = (const uint32_t *) srow + ((9 * p) >> 3);
        const unsigned long long bits = ((unsigned long long) w[1] << 32 | w[0]) >> ((4 * p) & 31);
        r = (uint32_t) bits & 0xfffu, g = (uint32_t) (bits >> 12) & 0xfffu, b = (uint32_t) (bits >> 24) & 0xfffu;
}
constexpr int kPxPerLane = 4; // a wave covers 4 x 64 consecutive pixels: the loads of all four are in flight together
template <int OUT>
__global__ __launch_bounds__(256) void r12l_px_kernel(const XArgs a, int npx) // npx = pixels of every line that belong to whole groups
{
        const int p0 = blockIdx.x * (64 * kPxPerLane) + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
        if (y >= a.height || p0 - (int) threadIdx.x >= npx) return;
        const uint8_t *XR srow = a.src + (long) y * a.spitch;
        uint8_t *XR drow = a.dst + (long) y * a.dpitch;
        uint32_t r[kPxPerLane], g[kPxPerLane], b[kPxPerLane];
#pragma unroll
        for (int u = 0; u < kPxPerLane; u++) {
                const int p = p0 + 64 * u;
                r12l_pixel(srow, p < npx ? p : 0, r[u], g[u], b[u]);
        }
#pragma unroll
        for (int u = 0; u < kPxPerLane; u++) {
                const int p = p0 + 64 * u;
                if (p >= npx) break;
                if (OUT == UG_PF_RGBA) {
                        ((uint32_t *) drow)[p] = a.am | (r[u] >> 4) << a.rs | (g[u] >> 4) << a.gs | (b[u] >> 4) << a.bs;
                } else if (OUT == UG_PF_Y416) {
                        const int R = r[u] << 4, G = g[u] << 4, B = b[u] << 4;
                        const uint32_t cb = (uint16_t) ((TO_CB(R, G, B) >> kBase) + (1 << 15)), yy = (uint16_t) ((TO_Y(R, G, B) >> kBase) + (1 << 12)),
                                       cr = (uint16_t) ((TO_CR(R, G, B) >> kBase) + (1 << 15));
                        ((uint2 *) drow)[p] = make_uint2(cb | yy << 16, cr | 0xFFFF0000u);
                } else { // RG48: three 16-bit samples
                        uint16_t *d = (uint16_t *) drow + 3 * p;
                        d[0] = (uint16_t) (r[u] << 4), d[1] = (uint16_t) (g[u] << 4), d[2] = (uint16_t) (b[u] << 4);
                }
        }
}
template <int OUT, int DB>
void launch_r12l_px(const XArgs &a, int ngroups, hipStream_t st)
{
        const int npx = 8 * ngroups, per_wave = 64 * kPxPerLane;
        const dim3 block(64, 4, 1), grid((unsigned) ((npx + per_wave - 1) / per_wave), (unsigned) ((a.height + 3) / 4), 1);
        hipLaunchKernelGGL((r12l_px_kernel<OUT>), grid, block, 0, st, a, npx);
}

// R12L -> RG48 / RGB / UYVY / R10k with 4 adjacent pixels per lane: 144 bits in (five words from word 9q/2 on, shifted by 16 bits for odd
// q), 24 / 12 / 8 / 16 contiguous bytes out
template <int OUT>
__global__ __launch_bounds__(256) void r12l_quad_kernel(const XArgs a, int nquads)
{
        const int q = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
        if (y >= a.height || q >= nquads) return;
        const uint32_t *XR s = (const uint32_t *) (a.src + (long) y * a.spitch) + ((9 * q) >> 1);
        const uint32_t sh = (q & 1) * 16;
        const uint32_t w[6] = { s[0], s[1], s[2], s[3], s[4], 0 };
        uint32_t n[5];
#pragma unroll
        for (int k = 0; k < 5; k++) n[k] = __builtin_amdgcn_alignbit(w[k + 1], w[k], sh);
        uint32_t v[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
                const int bit = 12 * i, j = bit >> 5, o = bit & 31;
                uint32_t x = n[j] >> o;
                if (o > 20) x |= n[j + 1] << (32 - o);
                v[i] = x & 0xfffu;
        }
        uint8_t *XR drow = a.dst + (long) y * a.dpitch;
        // 24- and 12-byte units: the wave's quads leave as one contiguous run of whole lines (ug::WaveWords; the lanes past the line have returned)
        constexpr int kW = OUT == UG_PF_RG48 ? 6 : 3;
        __shared__ uint32_t stage[(OUT == UG_PF_RG48 || OUT == UG_PF_RGB) ? 4 * ug::WaveWords<kW>::LDS_DWORDS : 1];
        const int lane = threadIdx.x, q0 = q - lane, units = min(64, nquads - q0);
        if (OUT == UG_PF_RG48) {
                uint32_t o[6];
#pragma unroll
                for (int k = 0; k < 6; k++) o[k] = v[2 * k] << 4 | v[2 * k + 1] << 20;
                ug::WaveWords<6>::store(drow + 24 * (long) q0, o, stage + threadIdx.y * ug::WaveWords<6>::LDS_DWORDS, lane, units, units);
        } else if (OUT == UG_PF_RGB) {
                uint32_t o[3];
#pragma unroll
                for (int k = 0; k < 3; k++) o[k] = v[4 * k] >> 4 | (v[4 * k + 1] >> 4) << 8 | (v[4 * k + 2] >> 4) << 16 | (v[4 * k + 3] >> 4) << 24;
                ug::WaveWords<3>::store(drow + 12 * (long) q0, o, stage + threadIdx.y * ug::WaveWords<3>::LDS_DWORDS, lane, units, units);
        } else if (OUT == UG_PF_UYVY) { // k_r12l_to_uyvy's arithmetic
                uint32_t out[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                        const int r1 = v[6 * i] << 4, g1 = v[6 * i + 1] << 4, b1 = v[6 * i + 2] << 4, r2 = v[6 * i + 3] << 4, g2 = v[6 * i + 4] << 4, b2 = v[6 * i + 5] << 4;
                        const int u = ((TO_CB(r1, g1, b1) + TO_CB(r2, g2, b2)) >> (kBase + 9)) + 128, vv = ((TO_CR(r1, g1, b1) + TO_CR(r2, g2, b2)) >> (kBase + 9)) + 128;
                        const int y1 = (TO_Y(r1, g1, b1) >> (kBase + 8)) + 16, y2 = (TO_Y(r2, g2, b2) >> (kBase + 8)) + 16;
                        out[i] = (uint32_t) (u & 0xff) | (uint32_t) (y1 & 0xff) << 8 | (uint32_t) (vv & 0xff) << 16 | (uint32_t) (y2 & 0xff) << 24;
                }
                ug::st_stream((uint2 *) (drow + 8 * (long) q), make_uint2(out[0], out[1]));
        } else { // R10k: k_r12l_to_r10k's bytes, the slip in the second pixel of every group of 8 included
                uint32_t out[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        const uint32_t r = v[3 * i], g = v[3 * i + 1], b = v[3 * i + 2];
                        uint32_t last = b & 0xff;
                        if (i == 1 && !(q & 1)) last = (b & 0xf0) | (r & 0xf);
                        out[i] = r >> 4 | ((r & 0xC) << 4 | g >> 6) << 8 | ((g & 0x3C) << 2 | b >> 8) << 16 | last << 24;
                }
                ug::st_stream((uint4 *) (drow + 16 * (long) q), make_uint4(out[0], out[1], out[2], out[3]));
        }
}
template <int OUT>
void launch_r12l_quad(const XArgs &a, int ngroups, hipStream_t st)
{
        const int nquads = 2 * ngroups;
        const dim3 block(64, 4, 1), grid((unsigned) ((nquads + 63) / 64), (unsigned) ((a.height + 3) / 4), 1);
        hipLaunchKernelGGL((r12l_quad_kernel<OUT>), grid, block, 0, st, a, nquads);
}

// -> R12L: a wave converts 4 x 64 pixels (32 groups), leaves their 36-bit values in LDS and writes the 288 words they make: word j takes
// its bits from pixel 8j/9 and the next one.
template <int IN>
__global__ __launch_bounds__(256) void to_r12l_px_kernel(const XArgs a, int npx)
{
        constexpr int kPx = 64 * kPxPerLane;
        __shared__ unsigned long long vals[4][kPx + 2];
        const int lane = threadIdx.x, p0 = blockIdx.x * kPx, y = blockIdx.y * 4 + threadIdx.y;
        if (y >= a.height || p0 >= npx) return; // wave-uniform
        const uint8_t *XR srow = a.src + (long) y * a.spitch;
        uint8_t *XR drow = a.dst + (long) y * a.dpitch;
        unsigned long long *v = vals[threadIdx.y];
        // RGBA / Y416: lane + 64 u (4 / 8 bytes per lane and load); RGB / RG48: the lane takes 4 adjacent pixels = 12 / 24 contiguous bytes
        constexpr bool kAdjacent = IN == UG_PF_RGB || IN == UG_PF_RG48;
        uint32_t r[kPxPerLane], g[kPxPerLane], b[kPxPerLane];
        if (kAdjacent) {
                const int first = p0 + 4 * lane;
                const bool in = first < npx; // npx is a multiple of 8: a lane's four pixels are inside or outside together
                if (IN == UG_PF_RGB) {
                        const uint32_t *s = (const uint32_t *) (srow + 3 * (long) (in ? first : 0));
                        const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
                        const uint32_t px[4] = { w0 & 0xffffff, w0 >> 24 | (w1 & 0xffff) << 8, w1 >> 16 | (w2 & 0xff) << 16, w2 >> 8 };
#pragma unroll
                        for (int u = 0; u < 4; u++) r[u] = (px[u] & 0xff) << 4, g[u] = (px[u] >> 8 & 0xff) << 4, b[u] = (px[u] >> 16) << 4;
                } else {
                        const uint2 *s = (const uint2 *) (srow + 6 * (long) (in ? first : 0));
                        const uint2 q0 = s[0], q1 = s[1], q2 = s[2];
                        const uint32_t w[6] = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                                const int i = 3 * u; // 16-bit sample index
                                r[u] = ((w[i / 2] >> (16 * (i % 2))) & 0xffff) >> 4;
                                g[u] = ((w[(i + 1) / 2] >> (16 * ((i + 1) % 2))) & 0xffff) >> 4;
                                b[u] = ((w[(i + 2) / 2] >> (16 * ((i + 2) % 2))) & 0xffff) >> 4;
                        }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) v[4 * lane + u] = (unsigned long long) b[u] << 24 | (unsigned long long) g[u] << 12 | r[u];
        } else {
                uint32_t in0[kPxPerLane], in1[kPxPerLane];
#pragma unroll
                for (int u = 0; u < kPxPerLane; u++) { // all loads first
                        const int p = p0 + 64 * u + lane, q = p < npx ? p : 0;
                        in1[u] = 0;
                        if (IN == UG_PF_RGBA) {
                                in0[u] = ((const uint32_t *) srow)[q];
                        } else { // Y416
                                const uint2 w = ((const uint2 *) srow)[q];
                                in0[u] = w.x, in1[u] = w.y;
                        }
                }
#pragma unroll
                for (int u = 0; u < kPxPerLane; u++) {
                        if (IN == UG_PF_RGBA) {
                                r[u] = (in0[u] & 0xff) << 4, g[u] = (in0[u] >> 8 & 0xff) << 4, b[u] = (in0[u] >> 16 & 0xff) << 4;
                        } else {
                                const int uu = (int) (in0[u] & 0xffff) - (1 << 15), ys = a.c[Y_SCALE] * ((int) (in0[u] >> 16) - (1 << 12)), vv = (int) (in1[u] & 0xffff) - (1 << 15);
                                r[u] = clamp_full(TO_R(ys, uu, vv) >> (kBase + 4), 12);
                                g[u] = clamp_full(TO_G(ys, uu, vv) >> (kBase + 4), 12);
                                b[u] = clamp_full(TO_B(ys, uu, vv) >> (kBase + 4), 12);
                        }
                        v[64 * u + lane] = (unsigned long long) b[u] << 24 | (unsigned long long) g[u] << 12 | r[u];
                }
        }
        if (lane < 2) v[kPx + lane] = 0;
        __syncthreads(); // waves that left above are not waited for
        const int nwords = (min(kPx, npx - p0) / 8) * 9;
        uint32_t *d = (uint32_t *) drow + (p0 / 8) * 9;
#pragma unroll
        for (int j0 = 0; j0 < kPx / 8 * 9; j0 += 64) {
                const int j = j0 + lane;
                if (j < nwords) {
                        const int q = (8 * j) / 9, o = 32 * j - 36 * q; // 0 <= o <= 32
                        const unsigned long long lo = v[q], hi = v[q + 1];
                        d[j] = (uint32_t) (lo >> o) | (uint32_t) (hi << (36 - o) & 0xFFFFFFFFull);
                }
        }
}
template <int IN, int SB>
void launch_to_r12l_px(const XArgs &a, int ngroups, hipStream_t st)
{
        const int npx = 8 * ngroups, per_wave = 64 * kPxPerLane;
        const dim3 block(64, 4, 1), grid((unsigned) ((npx + per_wave - 1) / per_wave), (unsigned) ((a.height + 3) / 4), 1);
        hipLaunchKernelGGL((to_r12l_px_kernel<IN>), grid, block, 0, st, a, npx);
}

using uyvy_to_rg48_body = k_yuv422_to_rgb_body<false, true>;
using yuyv_to_rgb_body = k_yuv422_to_rgb_body<true, false>;
enum Iter { I_PX, I_PAIR, I_G6, I_G8, I_COMP, I_DVS };
struct Vec { // the vector form of an entry: bytes in / out per iteration, iterations per lane
        void (*launch)(const XArgs &, int, hipStream_t);
        int sb, db, k;
};
#define VEC(body, SB, DB, K) { launch_xvec<body, SB, DB, K>, SB, DB, K }
// NOVEC: the one-iteration-per-lane kernel measured faster at 8K (fraction of 8 TB/s: it vs the vector form) -- 9-word R12L units
// cost more in LDS and registers than the compiler-merged accesses of the plain kernel
#define NOVEC { nullptr, 1, 1, 1 }
#define PXVEC(launcher, SB, DB) { launcher, SB, DB, 1 } // the R12L pairs: one lane per pixel over the whole groups of a line
struct Entry {
        int in, out;
        void (*kernel)(const XArgs);
        Iter iter;
        int coeff_depth;
        Vec vec;
};
const Entry kTable[] = {
        { UG_PF_DVS10, UG_PF_UYVY, k_dvs10_to_uyvy, I_DVS, 0, VEC(k_dvs10_to_uyvy_body, 4, 3, 16) },
        { UG_PF_DVS10, UG_PF_V210, k_dvs10_to_v210, I_COMP, 0, VEC(k_dvs10_to_v210_body, 4, 4, 4) },
        { UG_PF_R10K, UG_PF_RGBA, k_r10k_to_rgba, I_PX, 0, VEC(k_r10k_to_rgba_body, 4, 4, 4) },
        { UG_PF_R10K, UG_PF_RG48, k_r10k_to_rg48, I_PX, 0, VEC(k_r10k_to_rg48_body, 4, 6, 8) },
        { UG_PF_R10K, UG_PF_Y416, k_r10k_to_y416, I_PX, 16, VEC(k_r10k_to_y416_body, 4, 8, 4) },
        { UG_PF_R10K, UG_PF_RGB, k_r10k_to_rgb, I_PX, 0, VEC(k_r10k_to_rgb_body, 4, 3, 16) },
        { UG_PF_R10K, UG_PF_UYVY, k_r10k_to_uyvy, I_PAIR, 8, VEC(k_r10k_to_uyvy_body, 8, 4, 4) },
        { UG_PF_R12L, UG_PF_RGBA, k_r12l_to_rgba, I_G8, 0, PXVEC((launch_r12l_px<UG_PF_RGBA, 32>), 36, 32) },
        { UG_PF_R12L, UG_PF_RGB, k_r12l_to_rgb, I_G8, 0, PXVEC(launch_r12l_quad<UG_PF_RGB>, 36, 24) },
        { UG_PF_R12L, UG_PF_RG48, k_r12l_to_rg48, I_G8, 0, PXVEC(launch_r12l_quad<UG_PF_RG48>, 36, 48) },
        { UG_PF_R12L, UG_PF_R10K, k_r12l_to_r10k, I_G8, 0, PXVEC(launch_r12l_quad<UG_PF_R10K>, 36, 32) },
        { UG_PF_R12L, UG_PF_Y416, k_r12l_to_y416, I_G8, 16, PXVEC((launch_r12l_px<UG_PF_Y416, 64>), 36, 64) },
        { UG_PF_R12L, UG_PF_UYVY, k_r12l_to_uyvy, I_G8, 8, PXVEC(launch_r12l_quad<UG_PF_UYVY>, 36, 16) },
        { UG_PF_RGBA, UG_PF_R12L, k_rgb_to_r12l<4>, I_G8, 0, PXVEC((launch_to_r12l_px<UG_PF_RGBA, 32>), 32, 36) },
        { UG_PF_RGB, UG_PF_R12L, k_rgb_to_r12l<3>, I_G8, 0, PXVEC((launch_to_r12l_px<UG_PF_RGB, 24>), 24, 36) },
        { UG_PF_RGBA, UG_PF_RG48, k_rgba_to_rg48, I_PX, 0, VEC(k_rgba_to_rg48_body, 4, 6, 8) },
        { UG_PF_RGB, UG_PF_RG48, k_rgb_to_rg48, I_COMP, 0, VEC(k_rgb_to_rg48_body, 1, 2, 16) },
        { UG_PF_UYVY, UG_PF_RG48, k_yuv422_to_rgb<false, true>, I_PAIR, 8, VEC(uyvy_to_rg48_body, 4, 12, 4) },
        { UG_PF_RG48, UG_PF_R12L, k_rg48_to_r12l, I_G8, 0, PXVEC((launch_to_r12l_px<UG_PF_RG48, 48>), 48, 36) },
        { UG_PF_RG48, UG_PF_R10K, k_rg48_to_r10k, I_PX, 0, VEC(k_rg48_to_r10k_body, 6, 4, 8) },
        { UG_PF_RG48, UG_PF_RGB, k_rg48_to_rgb, I_PX, 0, VEC(k_rg48_to_rgb_body, 6, 3, 16) },
        { UG_PF_RG48, UG_PF_RGBA, k_rg48_to_rgba, I_PX, 0, VEC(k_rg48_to_rgba_body, 6, 4, 8) },
        { UG_PF_RG48, UG_PF_V210, k_rg48_to_v210, I_G6, 10, NOVEC /* 0.73 vs 0.53 */ },
        { UG_PF_RG48, UG_PF_Y216, k_rg48_to_y216, I_PAIR, 16, VEC(k_rg48_to_y216_body, 12, 8, 4) },
        { UG_PF_RG48, UG_PF_Y416, k_rg48_to_y416, I_PX, 16, VEC(k_rg48_to_y416_body, 6, 8, 8) },
        { UG_PF_Y416, UG_PF_RG48, k_y416_to_rgb<0>, I_PX, 16, VEC(k_y416_to_rgb_body<0>, 8, 6, 8) },
        { UG_PF_RGBA, UG_PF_VUYA, k_rgba_to_vuya, I_PX, 8, VEC(k_rgba_to_vuya_body, 4, 4, 4) },
        { UG_PF_YUYV, UG_PF_RGB, k_yuv422_to_rgb<true, false>, I_PAIR, 8, VEC(yuyv_to_rgb_body, 4, 6, 8) },
        { UG_PF_RGBA, UG_PF_R10K, k_rgba_to_r10k, I_PX, 0, VEC(k_rgba_to_r10k_body, 4, 4, 4) },
        { UG_PF_UYVY, UG_PF_Y216, k_uyvy_to_y216, I_PAIR, 0, VEC(k_uyvy_to_y216_body, 4, 8, 4) },
        { UG_PF_UYVY, UG_PF_Y416, k_uyvy_to_y416, I_PAIR, 0, VEC(k_uyvy_to_y416_body, 4, 16, 4) },
        { UG_PF_VUYA, UG_PF_Y416, k_vuya_to_y416, I_PX, 0, VEC(k_vuya_to_y416_body, 4, 8, 4) },
        { UG_PF_VUYA, UG_PF_UYVY, k_vuya_to_uyvy, I_PAIR, 0, VEC(k_vuya_to_uyvy_body, 8, 4, 4) },
        { UG_PF_VUYA, UG_PF_RGB, k_vuya_to_rgb, I_PX, 8, VEC(k_vuya_to_rgb_body, 4, 3, 16) },
        { UG_PF_Y216, UG_PF_UYVY, k_y216_to_uyvy, I_PAIR, 0, VEC(k_y216_to_uyvy_body, 8, 4, 4) },
        { UG_PF_Y216, UG_PF_V210, k_y216_to_v210, I_G6, 0, VEC(k_y216_to_v210_body, 24, 16, 2) },
        { UG_PF_Y416, UG_PF_UYVY, k_y416_to_uyvy, I_PAIR, 0, VEC(k_y416_to_uyvy_body, 16, 4, 4) },
        { UG_PF_Y416, UG_PF_V210, k_y416_to_v210, I_G6, 0, VEC(k_y416_to_v210_body, 48, 16, 1) },
        { UG_PF_Y416, UG_PF_R12L, k_y416_to_r12l, I_G8, 16, PXVEC((launch_to_r12l_px<UG_PF_Y416, 64>), 64, 36) },
        { UG_PF_Y416, UG_PF_R10K, k_y416_to_rgb<1>, I_PX, 16, VEC(k_y416_to_rgb_body<1>, 8, 4, 4) },
        { UG_PF_Y416, UG_PF_RGB, k_y416_to_rgb<2>, I_PX, 16, VEC(k_y416_to_rgb_body<2>, 8, 3, 16) },
        { UG_PF_Y416, UG_PF_RGBA, k_y416_to_rgb<3>, I_PX, 16, VEC(k_y416_to_rgb_body<3>, 8, 4, 4) },
        { UG_PF_V210, UG_PF_Y216, k_v210_to_y2xx<false>, I_G6, 0, VEC(k_v210_to_y2xx_body<false>, 16, 24, 2) },
        { UG_PF_V210, UG_PF_Y416, k_v210_to_y2xx<true>, I_G6, 0, VEC(k_v210_to_y2xx_body<true>, 16, 48, 1) },
};

const Entry *find(int in, int out)
{
        for (const Entry &e : kTable) {
                if (e.in == in && e.out == out) return &e;
        }
        return nullptr;
}

} // namespace

namespace ug {

int pixfmt_ext_supported(ug_pixfmt_t in, ug_pixfmt_t out) { return find(in, out) != nullptr; }

int pixfmt_ext_convert(ug_pixfmt_t in, ug_pixfmt_t out, const void *src, void *dst, int width, int height, int src_pitch, int dst_pitch, int dst_len,
                       int rshift, int gshift, int bshift, hipStream_t st)
{
        const Entry *e = find(in, out);
        if (!e) return UG_HIP_EUNSUPP;
        if ((unsigned) rshift > 24 || (unsigned) gshift > 24 || (unsigned) bshift > 24) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: rgb shift out of range");
                return UG_HIP_EINVAL;
        }
        // the 16- and 32-bit formats are addressed as such (the reference asserts the same alignments)
        const bool wide_src = in == UG_PF_RG48 || in == UG_PF_Y216 || in == UG_PF_Y416 || in == UG_PF_V210 || in == UG_PF_DVS10;
        const bool wide_dst = out != UG_PF_RGB && out != UG_PF_R12L && out != UG_PF_R10K;
        if ((wide_src && (((uintptr_t) src | (uintptr_t) src_pitch) & (in == UG_PF_V210 || in == UG_PF_DVS10 ? 3 : 1))) ||
            (wide_dst && (((uintptr_t) dst | (uintptr_t) dst_pitch) & (out == UG_PF_RG48 || out == UG_PF_Y216 || out == UG_PF_Y416 ? 1 : 3))) ||
            (out == UG_PF_R10K && in != UG_PF_R12L && in != UG_PF_Y416 && (((uintptr_t) dst | (uintptr_t) dst_pitch) & 3))) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: buffer or pitch not aligned for this pair");
                return UG_HIP_EINVAL;
        }
        if (dst_pitch < dst_len) {
                ug::set_last_error_msg("ug_hip_pixfmt_convert: dst_pitch is smaller than a line of the output format");
                return UG_HIP_EINVAL;
        }
        XArgs a = {};
        a.src = (const uint8_t *) src, a.dst = (uint8_t *) dst;
        a.spitch = src_pitch, a.dpitch = dst_pitch;
        a.width = width, a.height = height, a.L = dst_len;
        a.rs = rshift, a.gs = gshift, a.bs = bshift;
        a.am = 0xFFFFFFFFu ^ (0xFFu << rshift) ^ (0xFFu << gshift) ^ (0xFFu << bshift);
        if (e->coeff_depth) memcpy(a.c, ug::kColorCoeffs[1][ug::color_depth_slot(e->coeff_depth)], sizeof a.c); // get_color_coeffs(CS_DFL, depth): BT.709
        int nx = 0; // an upper bound of the lanes a line needs; every kernel re-derives its exact count from dst_len
        switch (e->iter) {
        case I_PX: nx = width + 1; break;
        case I_PAIR: nx = (width + 1) / 2 + 1; break;
        case I_G6: nx = (width + 5) / 6 + 1; break;
        case I_G8: nx = (width + 7) / 8 + 1; break;
        case I_COMP: nx = dst_len; break;
        case I_DVS: nx = (int) (dst_len / 1.5) / 16 * 8; break;
        }
        // The 16-byte-aligned interior of every line goes K iterations per lane with 128-bit accesses (xvec_kernel): as many whole
        // vector units as have all their iterations inside the line on both sides -- output bytes within dst_len, source bytes within
        // the source line.  Whatever is left of a line (ragged ends, clipped last groups; the whole line when a pointer or pitch is
        // not 16-byte aligned) takes the one-iteration-per-lane kernel from iteration x0 on.
        int nvec = 0;
        const Vec &v = e->vec;
        static const bool no_vec = getenv("UG_PIXFMT_NO_VEC") != nullptr; // A/B switch
        if (v.launch && !no_vec && !((((uintptr_t) src | (uintptr_t) dst) | (uintptr_t) src_pitch | (uintptr_t) dst_pitch) & 15)) {
                const int iters = e->iter == I_DVS ? (int) (dst_len / 1.5) / 16 * 8 : dst_len / v.db; // iterations that write all their bytes
                const int src_line = ug::linesize(in, width);
                nvec = min(iters / v.k, src_line / (v.sb * v.k));
                if (nvec > 0) {
                        v.launch(a, nvec, st);
                        UG_HIP_LAUNCH_CHECK();
                }
        }
        a.x0 = nvec * v.k;
        if (v.launch) nx = min(nx, (dst_len + v.db - 1) / v.db + 1); // no converter runs more iterations than its output bytes allow (+ the odd one)
        if (nx > a.x0) {
                const dim3 block(64, 4, 1), grid((unsigned) ((nx - a.x0 + 63) / 64), (unsigned) ((height + 3) / 4), 1);
                hipLaunchKernelGGL(e->kernel, grid, block, 0, st, a);
                UG_HIP_LAUNCH_CHECK();
        }
        return UG_HIP_SUCCESS;
}

} // namespace ug

// get_best_decoder_from (pixfmt_conv.c:3126-3172): of the candidates reachable from `in`, the one compare_pixdesc (video_codec.c:1148-1192,
// default preference "dsc": depth, subsampling, colour space) ranks first; ties go to the lower codec_t (types.h:62-112).
namespace {
struct PixDesc {
        int depth, subsampling, rgb, codec_t_value;
};
bool pix_desc(ug_pixfmt_t f, PixDesc &d) // get_pixfmt_desc (video_codec.c) for the codecs of this library
{
        switch (f) {
        case UG_PF_RGBA: d = { 8, 4444, 1, 1 }; return true;
        case UG_PF_UYVY: d = { 8, 4220, 0, 2 }; return true;
        case UG_PF_YUYV: d = { 8, 4220, 0, 3 }; return true;
        case UG_PF_VUYA: d = { 8, 4444, 0, 4 }; return true;
        case UG_PF_R10K: d = { 10, 4440, 1, 5 }; return true;
        case UG_PF_R12L: d = { 12, 4440, 1, 6 }; return true;
        case UG_PF_V210: d = { 10, 4220, 0, 7 }; return true;
        case UG_PF_DVS10: d = { 10, 4220, 0, 8 }; return true;
        case UG_PF_RGB: d = { 8, 4440, 1, 12 }; return true;
        case UG_PF_BGR: d = { 8, 4440, 1, 20 }; return true;
        case UG_PF_RG48: d = { 16, 4440, 1, 27 }; return true;
        case UG_PF_Y216: d = { 16, 4220, 0, 30 }; return true;
        case UG_PF_Y416: d = { 16, 4444, 0, 31 }; return true;
        default: return false;
        }
}
int compare_pixdesc(const PixDesc &a, const PixDesc &b, const PixDesc &src)
{
        if (a.depth != b.depth && (a.depth < src.depth || b.depth < src.depth)) return b.depth - a.depth;
        if (a.subsampling != b.subsampling && (a.subsampling < src.subsampling || b.subsampling < src.subsampling)) return b.subsampling - a.subsampling;
        if (a.rgb != b.rgb) return a.rgb == src.rgb ? -1 : 1;
        if (a.depth != b.depth) return a.depth - b.depth;
        if (a.subsampling != b.subsampling) return a.subsampling - b.subsampling;
        return 0;
}
} // namespace

extern "C" int ug_hip_pixfmt_best(ug_pixfmt_t in, const ug_pixfmt_t *candidates, ug_pixfmt_t *out)
{
        PixDesc src, best = {}, d;
        if (!candidates || !out || !pix_desc(in, src)) return UG_HIP_EINVAL;
        for (const ug_pixfmt_t *it = candidates; *it != UG_PF_NONE; ++it) { // `in` itself, unless RGB / RGBA (their copy may change the shifts)
                if (*it == in && in != UG_PF_RGBA && in != UG_PF_RGB) {
                        *out = in;
                        return UG_HIP_SUCCESS;
                }
        }
        bool have = false;
        for (const ug_pixfmt_t *it = candidates; *it != UG_PF_NONE; ++it) {
                if (!ug_hip_pixfmt_supported(in, *it) || !pix_desc(*it, d)) continue;
                int c = have ? compare_pixdesc(d, best, src) : -1;
                if (c == 0) c = d.codec_t_value - best.codec_t_value;
                if (c < 0) best = d, *out = *it, have = true;
        }
        return have ? UG_HIP_SUCCESS : UG_HIP_EUNSUPP;
}

extern "C" int ug_hip_pixfmt_line_func(const char *func, const void *src, void *dst, int width, int height, int src_pitch, int dst_pitch, int dst_len,
                                       int rshift, int gshift, int bshift, ug_hip_stream_t stream)
{
        if (!ug::dims_ok(width, height) || src_pitch < 0 || dst_pitch < 0 || dst_len > 8 * ug::kMaxDim || !ug::planes_ok(height, { src_pitch, dst_pitch, dst_len })) {
                return ug::refuse_size("ug_hip_pixfmt_line_func");
        }
        if (!func || !src || !dst || width <= 0 || height <= 0 || dst_len < 0 || (((uintptr_t) src | (uintptr_t) src_pitch) & 3)) {
                ug::set_last_error_msg("ug_hip_pixfmt_line_func: bad arguments (the sources are read as 32-bit words)");
                return UG_HIP_EINVAL;
        }
        XArgs a = {};
        a.src = (const uint8_t *) src, a.dst = (uint8_t *) dst, a.spitch = src_pitch, a.dpitch = dst_pitch, a.width = width, a.height = height, a.L = dst_len;
        void (*k)(const XArgs) = nullptr;
        int nx = 0;
        if (!strcmp(func, "vc_copylineUYVYtoGrayscale")) {
                k = k_uyvy_to_grayscale, nx = dst_len / 2;
        } else if (!strcmp(func, "vc_copylineABGRtoRGB")) {
                k = k_rgba_to_rgb_shift, nx = dst_len / 3, a.rs = 24, a.gs = 16, a.bs = 8;
        } else if (!strcmp(func, "vc_copylineBGRAtoRGB")) {
                k = k_rgba_to_rgb_shift, nx = dst_len / 3, a.rs = 16, a.gs = 8, a.bs = 0;
        } else if (!strcmp(func, "vc_copylineToRGBA_inplace")) {
                if ((unsigned) rshift > 24 || (unsigned) gshift > 24 || (unsigned) bshift > 24 || (((uintptr_t) dst | (uintptr_t) dst_pitch) & 3)) return UG_HIP_EINVAL;
                k = k_to_rgba_inplace, nx = dst_len / 4, a.rs = rshift, a.gs = gshift, a.bs = bshift;
        } else {
                ug::set_last_error_msg("ug_hip_pixfmt_line_func: unknown function name");
                return UG_HIP_EUNSUPP;
        }
        if (nx <= 0) return UG_HIP_SUCCESS;
        hipLaunchKernelGGL(k, dim3((unsigned) ((nx + 63) / 64), (unsigned) ((height + 3) / 4), 1), dim3(64, 4, 1), 0, (hipStream_t) stream, a);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}
