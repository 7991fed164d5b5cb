// lavc_conv.hip -- SURVEY.md 8(f) N3: the pixel-format converters of src/libavcodec/to_lavc_vid_conv.c (UltraGrid codec -> the
// AVFrame an encoder takes) and src/libavcodec/from_lavc_vid_conv.c (a decoder's AVFrame -> UltraGrid codec), for the codecs of the
// hot path (UYVY, v210, RGB, RGBA) against the frame formats encoders and decoders really use (planar YUV 4:2:0 / 4:2:2 / 4:4:4 at 8
// and 10 bit, NV12, P010, P210, GBRP, XV30, Y210, VUYA ...).  This is what the reference's reserved GPU hook is for
// (to_lavc_vid_conv_cuda.h:55-66, from_lavc_vid_conv_cuda.h:55-66: both stubs that return NULL).
//
// Every kernel restates ONE reference function statement for statement, one lane per iteration of its inner loop (and per row or
// row pair), including the reference's slips where it has them -- the results must be identical, and the tests compare against
// the reference's own functions compiled from /root/reference (oracle/_ref/libugref_lavc.so).  Known slips kept: yuv420p_to_v210
// and yuv444p1Xle_to_v210 leave some luma samples unshifted (from_lavc_vid_conv.c:599,1099,1107), nv12_to_rgb gives both pixels
// of a pair the first one's colour (:797-811), yuv444p_to_rgb does not subtract the luma offset (:976), p210le_to_uyvy overwrites
// Cb and emits three bytes per pair (:1622-1625).
//
// Pure sample shuffles and Q14 integer colour arithmetic (color_space.h:100-110), HBM-bound.
#include <string.h>

#include "ug_common.h"

namespace ug {
const int kColorCoeffs[2][5][14] = {
        { { 4899, 9617, 1868, -2765, -5427, 8192, 8191, -6860, -1331, 16384, 22970, -5638, -11700, 29032 },
          { 4207, 8260, 1604, -2428, -4768, 7196, 7195, -6026, -1169, 19077, 26149, -6419, -13320, 33050 },
          { 4195, 8235, 1599, -2421, -4754, 7175, 7174, -6008, -1166, 19133, 26226, -6438, -13359, 33148 },
          { 4192, 8229, 1598, -2420, -4750, 7170, 7169, -6004, -1165, 19147, 26245, -6442, -13369, 33172 },
          { 4191, 8228, 1598, -2419, -4749, 7168, 7167, -6002, -1165, 19152, 26251, -6444, -13372, 33179 } },
        { { 3484, 11717, 1183, -1877, -6315, 8192, 8191, -7441, -750, 16384, 25800, -3069, -7671, 30402 },
          { 2992, 10063, 1016, -1649, -5547, 7196, 7195, -6536, -659, 19077, 29371, -3494, -8733, 34610 },
          { 2983, 10034, 1013, -1644, -5531, 7175, 7174, -6517, -657, 19133, 29457, -3504, -8758, 34712 },
          { 2981, 10026, 1012, -1643, -5527, 7170, 7169, -6512, -656, 19147, 29479, -3507, -8765, 34737 },
          { 2980, 10024, 1012, -1643, -5525, 7168, 7167, -6511, -656, 19152, 29486, -3507, -8767, 34745 } },
};
} // namespace ug

namespace {

constexpr auto &kCoeffs = ug::kColorCoeffs;
int depth_slot(int depth) { return ug::color_depth_slot(depth); }

struct Args {
        uint8_t *d[4]; // AVFrame::data
        int ls[4];     // AVFrame::linesize
        uint8_t *buf;  // packed UltraGrid buffer
        long pitch;    // its line size
        int w, h;
        int c[14];     // struct color_coeffs: y_r y_g y_b cb_r cb_g cb_b cr_r cr_g cr_b y_scale r_cr g_cb g_cr b_cb
        int rs, gs, bs;
        uint32_t am;   // alpha mask
};
enum { Y_R, Y_G, Y_B, CB_R, CB_G, CB_B, CR_R, CR_G, CR_B, Y_SCALE, R_CR, G_CB, G_CR, B_CB };
constexpr int kBase = 14; // COMP_BASE, color_space.h:71

#define UG_XY()                                                   \
        const int x = blockIdx.x * blockDim.x + threadIdx.x;      \
        const int y = blockIdx.y * blockDim.y + threadIdx.y
#define ROW(T, plane, row) ((T *) (a.d[plane] + (long) (row) * a.ls[plane]))
#define BUF(T, row) ((T *) (a.buf + (long) (row) * a.pitch))

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clamp_full(int v, int depth) { return clampi(v, 1 << (depth - 8), (255 << (depth - 8)) - 1); } // CLAMP_FULL
__device__ __forceinline__ uint32_t mk_rgba(const Args &a, int r, int g, int b)                                                  // MK_RGBA(.., 8)
{
        return a.am | (uint32_t) clamp_full(r, 8) << a.rs | (uint32_t) clamp_full(g, 8) << a.gs | (uint32_t) clamp_full(b, 8) << a.bs;
}
__device__ __forceinline__ uint32_t v210w(uint32_t lo, uint32_t mid, uint32_t hi) { return lo | mid << 10 | hi << 20; }
__device__ __forceinline__ void st4(uint32_t *dst, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
        if (((uintptr_t) dst & 15) == 0) {
                ug::st_stream((uint4 *) dst, make_uint4(w0, w1, w2, w3));
        } else {
                ug::st_stream(dst, w0), ug::st_stream(dst + 1, w1), ug::st_stream(dst + 2, w2), ug::st_stream(dst + 3, w3);
        }
}

// ================================ UltraGrid codec -> AVFrame (to_lavc_vid_conv.c) ==================================================
// a.buf / a.pitch = the packed input (pitch = vc_get_linesize(width, codec), as the reference strides it), a.d / a.ls = output planes

__global__ void k_uyvy_to_yuv444p(const Args a) // to_lavc_vid_conv.c:174-190
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint32_t s = BUF(const uint32_t, y)[x];
        const uint8_t cb = s & 0xff, y0 = (s >> 8) & 0xff, cr = (s >> 16) & 0xff, y1 = s >> 24;
        uint8_t *py = ROW(uint8_t, 0, y) + 2 * x, *pcb = ROW(uint8_t, 1, y) + 2 * x, *pcr = ROW(uint8_t, 2, y) + 2 * x;
        py[0] = y0, pcb[0] = cb, pcr[0] = cr;
        if (2 * x + 1 < a.ls[0]) py[1] = y1; // the second pixel of a lone last pair may not fit a tightly packed line
        if (2 * x + 1 < a.ls[1]) pcb[1] = cb;
        if (2 * x + 1 < a.ls[2]) pcr[1] = cr;
}

__global__ void k_uyvy_to_vuya(const Args a) // :155-172
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint32_t s = BUF(const uint32_t, y)[x];
        const uint32_t cb = s & 0xff, y0 = (s >> 8) & 0xff, cr = (s >> 16) & 0xff, y1 = s >> 24;
        uint32_t *dst = ROW(uint32_t, 0, y) + 2 * x;
        dst[0] = cr | cb << 8 | y0 << 16 | 0xff000000u;
        if (8 * x + 8 <= a.ls[0]) dst[1] = cr | cb << 8 | y1 << 16 | 0xff000000u;
}

struct V210Group {
        uint32_t y[6], cb[3], cr[3];
};
__device__ __forceinline__ V210Group v210_unpack(const uint32_t *src)
{
        const uint32_t w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
        V210Group g;
        g.y[0] = (w0 >> 10) & 0x3ff, g.y[1] = w1 & 0x3ff, g.y[2] = (w1 >> 20) & 0x3ff;
        g.y[3] = (w2 >> 10) & 0x3ff, g.y[4] = w3 & 0x3ff, g.y[5] = (w3 >> 20) & 0x3ff;
        g.cb[0] = w0 & 0x3ff, g.cb[1] = (w1 >> 10) & 0x3ff, g.cb[2] = (w2 >> 20) & 0x3ff;
        g.cr[0] = (w0 >> 20) & 0x3ff, g.cr[1] = w2 & 0x3ff, g.cr[2] = (w3 >> 10) & 0x3ff;
        return g;
}
// Several reference converters write whole groups past `width`; past the end of the LINE that lands in the next line, whose own
// conversion then overwrites it.  Lines are converted concurrently here, so such stores stop at the line size (`room` = samples that
// still fit): inside the frame the result is the same.
__device__ __forceinline__ int room16(long line_bytes, long first_sample, int want)
{
        const long fit = line_bytes / 2 - first_sample;
        return (int) (fit < want ? (fit < 0 ? 0 : fit) : want);
}
template <int N>
__device__ __forceinline__ void st16(uint16_t *p, const uint32_t (&v)[N], int shift, int n = N)
{
        if (n < N) {
#pragma unroll
                for (int i = 0; i < N; i++) {
                        if (i < n) p[i] = (uint16_t) (v[i] << shift);
                }
                return;
        }
        if (((uintptr_t) p & 3) == 0 && N % 2 == 0) {
#pragma unroll
                for (int i = 0; i < N / 2; i++) ug::st_stream((uint32_t *) p + i, (v[2 * i] << shift) | (v[2 * i + 1] << shift) << 16);
        } else {
#pragma unroll
                for (int i = 0; i < N; i++) ug::st_stream(p + i, (uint16_t) (v[i] << shift));
        }
}

enum { V_420P10, V_422P10, V_444P10, V_444P16, V_P210 };
// v210_to_yuv420p10le :197-262, v210_to_yuv422p10le :264-301, v210_to_yuv444p10le :303-346, v210_to_yuv444p16le :348-389,
// v210_to_p210le :577-609: width / 6 groups per line.  4:2:0 walks line pairs; with an odd height the reference reads and writes one
// line past the picture -- here the last line stands alone (its chroma is its own).
template <int MODE>
__global__ void k_v210_to_planar(const Args a)
{
        UG_XY();
        const int rows = MODE == V_420P10 ? (a.h + 1) / 2 : a.h;
        if (x >= a.w / 6 || y >= rows) return;
        if (MODE == V_420P10) {
                const int y0 = 2 * y, y1 = 2 * y + 1 < a.h ? 2 * y + 1 : 2 * y;
                const V210Group g0 = v210_unpack(BUF(const uint32_t, y0) + 4 * x), g1 = v210_unpack(BUF(const uint32_t, y1) + 4 * x);
                st16<6>(ROW(uint16_t, 0, y0) + 6 * x, g0.y, 0);
                if (y1 != y0) st16<6>(ROW(uint16_t, 0, y1) + 6 * x, g1.y, 0);
                uint16_t *cb = (uint16_t *) (a.d[1] + (long) a.ls[1] * y0 / 2) + 3 * x, *cr = (uint16_t *) (a.d[2] + (long) a.ls[2] * y0 / 2) + 3 * x;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        cb[i] = (uint16_t) ((g0.cb[i] + g1.cb[i]) / 2);
                        cr[i] = (uint16_t) ((g0.cr[i] + g1.cr[i]) / 2);
                }
                return;
        }
        const V210Group g = v210_unpack(BUF(const uint32_t, y) + 4 * x);
        if (MODE == V_422P10) {
                st16<6>(ROW(uint16_t, 0, y) + 6 * x, g.y, 0);
                uint16_t *cb = ROW(uint16_t, 1, y) + 3 * x, *cr = ROW(uint16_t, 2, y) + 3 * x;
#pragma unroll
                for (int i = 0; i < 3; i++) cb[i] = (uint16_t) g.cb[i], cr[i] = (uint16_t) g.cr[i];
        } else if (MODE == V_444P10 || MODE == V_444P16) {
                const int sh = MODE == V_444P16 ? 6 : 0;
                const uint32_t cb[6] = { g.cb[0], g.cb[0], g.cb[1], g.cb[1], g.cb[2], g.cb[2] }, cr[6] = { g.cr[0], g.cr[0], g.cr[1], g.cr[1], g.cr[2], g.cr[2] };
                st16<6>(ROW(uint16_t, 0, y) + 6 * x, g.y, sh);
                st16<6>(ROW(uint16_t, 1, y) + 6 * x, cb, sh);
                st16<6>(ROW(uint16_t, 2, y) + 6 * x, cr, sh);
        } else { // V_P210
                const uint32_t c[6] = { g.cb[0], g.cr[0], g.cb[1], g.cr[1], g.cb[2], g.cr[2] };
                st16<6>(ROW(uint16_t, 0, y) + 6 * x, g.y, 6);
                st16<6>(ROW(uint16_t, 1, y) + 6 * x, c, 6);
        }
}

__global__ void k_v210_to_xv30(const Args a) // :393-417, (width + 5) / 6 groups, six whole pixels each
{
        UG_XY();
        if (x >= (a.w + 5) / 6 || y >= a.h) return;
        const uint32_t *src = BUF(const uint32_t, y) + 4 * x;
        const V210Group g = v210_unpack(src);
        // XV30 pixel = U | Y << 10 | V << 20; chroma of a pair goes to both of its pixels.  The first two pixels are built there on top of
        // the first v210 word and so inherit its two padding bits
        const uint32_t pad = src[0] & 0xC0000000u;
        uint32_t *dst = ROW(uint32_t, 0, y) + 6 * x;
#pragma unroll
        for (int i = 0; i < 6; i++) {
                if (4 * (6 * x + i) + 4 <= a.ls[0]) dst[i] = g.cb[i / 2] | g.y[i] << 10 | g.cr[i / 2] << 20 | (i < 2 ? pad : 0u);
        }
}

__global__ void k_v210_to_y210(const Args a) // :421-450
{
        UG_XY();
        if (x >= (a.w + 5) / 6 || y >= a.h) return;
        const V210Group g = v210_unpack(BUF(const uint32_t, y) + 4 * x);
        const uint32_t o[12] = { g.y[0], g.cb[0], g.y[1], g.cr[0], g.y[2], g.cb[1], g.y[3], g.cr[1], g.y[4], g.cb[2], g.y[5], g.cr[2] };
        st16<12>(ROW(uint16_t, 0, y) + 12 * x, o, 6, room16(a.ls[0], 12L * x, 12));
}

template <int BPP>
__global__ void k_rgb_to_gbrp(const Args a) // rgb_rgba_to_gbrp :1318-1334 (source lines are bpp * width apart)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint8_t *src = a.buf + (long) y * (BPP * a.w) + (long) BPP * x;
        ROW(uint8_t, 2, y)[x] = src[0];
        ROW(uint8_t, 0, y)[x] = src[1];
        ROW(uint8_t, 1, y)[x] = src[2];
}

__global__ void k_rgb_to_yuv444p(const Args a) // :1187-1227
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint8_t *src = BUF(const uint8_t, y) + 3 * x;
        const int r = src[0], g = src[1], b = src[2];
        ROW(uint8_t, 0, y)[x] = (uint8_t) (((r * a.c[Y_R] + g * a.c[Y_G] + b * a.c[Y_B]) >> kBase) + 16);
        ROW(uint8_t, 1, y)[x] = (uint8_t) (((r * a.c[CB_R] + g * a.c[CB_G] + b * a.c[CB_B]) >> kBase) + 128);
        ROW(uint8_t, 2, y)[x] = (uint8_t) (((r * a.c[CR_R] + g * a.c[CR_G] + b * a.c[CR_B]) >> kBase) + 128);
}

// ---- sources beyond the hot-path four: Y216, Y416, R10k, R12L, RG48 -----------------------------------------------------------------

template <int DEPTH>
__global__ void k_y216_to_yuv422p(const Args a) // y216_to_yuv422pXXle :1236-1254, (width + 1) / 2 pairs
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint16_t *src = BUF(const uint16_t, y) + 4 * x;
        ROW(uint16_t, 0, y)[2 * x] = src[0] >> (16 - DEPTH);
        ROW(uint16_t, 1, y)[x] = src[1] >> (16 - DEPTH);
        if (room16(a.ls[0], 2L * x, 2) == 2) ROW(uint16_t, 0, y)[2 * x + 1] = src[2] >> (16 - DEPTH);
        ROW(uint16_t, 2, y)[x] = src[3] >> (16 - DEPTH);
}

__global__ void k_y216_to_yuv444p16le(const Args a) // :1266-1289
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint16_t *src = BUF(const uint16_t, y) + 4 * x;
        uint16_t *py = ROW(uint16_t, 0, y) + 2 * x, *pcb = ROW(uint16_t, 1, y) + 2 * x, *pcr = ROW(uint16_t, 2, y) + 2 * x;
        py[0] = src[0], pcb[0] = src[1], pcr[0] = src[3];
        if (room16(a.ls[0], 2L * x, 2) == 2) py[1] = src[2];
        if (room16(a.ls[1], 2L * x, 2) == 2) pcb[1] = src[1];
        if (room16(a.ls[2], 2L * x, 2) == 2) pcr[1] = src[3];
}

__global__ void k_y416_to_xv30(const Args a) // :454-470
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint16_t *src = BUF(const uint16_t, y) + 4 * x;
        const uint32_t u = src[0], Y = src[1], v = src[2], al = src[3];
        ROW(uint32_t, 0, y)[x] = (al >> 14U) << 30U | (v >> 6U) << 20U | (Y >> 6U) << 10U | (u >> 6U);
}

__global__ void k_y416_to_yuv444p(const Args a) // :476-498: the odd bytes (most significant of each little-endian sample)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint8_t *src = BUF(const uint8_t, y) + 8 * x;
        ROW(uint8_t, 1, y)[x] = src[1];
        ROW(uint8_t, 0, y)[x] = src[3];
        ROW(uint8_t, 2, y)[x] = src[5];
}

template <int DEPTH>
__global__ void k_y416_to_yuv444pXX(const Args a) // y416_to_yuv444pXXle :500-527
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint16_t *src = BUF(const uint16_t, y) + 4 * x;
        ROW(uint16_t, 1, y)[x] = src[0] >> (16 - DEPTH);
        ROW(uint16_t, 0, y)[x] = src[1] >> (16 - DEPTH);
        ROW(uint16_t, 2, y)[x] = src[2] >> (16 - DEPTH);
}

__device__ __forceinline__ void r10k_px(const uint8_t *src, int &r, int &g, int &b)
{
        r = src[0] << 2 | src[1] >> 6;
        g = (src[1] & 0x3f) << 4 | src[2] >> 4;
        b = (src[2] & 0x0f) << 6 | src[3] >> 2;
}
#define UG_RGB_TO_Y(r, g, b) ((r) * a.c[Y_R] + (g) * a.c[Y_G] + (b) * a.c[Y_B])
#define UG_RGB_TO_CB(r, g, b) ((r) * a.c[CB_R] + (g) * a.c[CB_G] + (b) * a.c[CB_B])
#define UG_RGB_TO_CR(r, g, b) ((r) * a.c[CR_R] + (g) * a.c[CR_G] + (b) * a.c[CR_B])

// r10k_to_yuv42Xp10le :616-687.  4:2:2 (VSUB 1): one lane per pixel pair.  4:2:0 (VSUB 2): one lane per pixel pair of a line pair, the two
// lines in the reference's order, because odd pairs average with what the chroma plane already holds (:670-677) -- first its previous
// content, then what the even line left there.
template <int VSUB>
__global__ void k_r10k_to_yuv42Xp10le(const Args a)
{
        UG_XY();
        if (x >= a.w / 2 || y >= (VSUB == 2 ? (a.h + 1) / 2 : a.h)) return;
#pragma unroll
        for (int l = 0; l < VSUB; l++) {
                const int row = VSUB * y + l;
                if (row >= a.h) break;
                const uint8_t *src = BUF(const uint8_t, row) + 8 * x;
                int r, g, b;
                r10k_px(src, r, g, b);
                ROW(uint16_t, 0, row)[2 * x] = (uint16_t) ((UG_RGB_TO_Y(r, g, b) >> kBase) + (1 << 6));
                int cb = (UG_RGB_TO_CB(r, g, b) >> kBase) + (1 << 9), cr = (UG_RGB_TO_CR(r, g, b) >> kBase) + (1 << 9);
                r10k_px(src + 4, r, g, b);
                ROW(uint16_t, 0, row)[2 * x + 1] = (uint16_t) ((UG_RGB_TO_Y(r, g, b) >> kBase) + (1 << 6));
                cb += (UG_RGB_TO_CB(r, g, b) >> kBase) + (1 << 9);
                cr += (UG_RGB_TO_CR(r, g, b) >> kBase) + (1 << 9);
                cb /= 2, cr /= 2;
                uint16_t *pcb = ROW(uint16_t, 1, row / VSUB) + x, *pcr = ROW(uint16_t, 2, row / VSUB) + x;
                if (VSUB == 1 || x % 2 == 0) {
                        *pcb = (uint16_t) cb, *pcr = (uint16_t) cr;
                } else {
                        *pcb = (uint16_t) ((*pcb + cb) / 2), *pcr = (uint16_t) ((*pcr + cr) / 2);
                }
        }
}

// r10k_to_yuv444pXXle :706-740 (IN_DEPTH 10), rg48_to_yuv444pXXle :1136-1170 (IN_DEPTH 16): one lane per pixel
template <int IN_DEPTH, int OUT_DEPTH>
__global__ void k_rgb_to_yuv444pXX(const Args a)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        int r, g, b;
        if (IN_DEPTH == 10) {
                r10k_px(BUF(const uint8_t, y) + 4 * x, r, g, b);
        } else {
                const uint16_t *src = BUF(const uint16_t, y) + 3 * x;
                r = src[0], g = src[1], b = src[2];
        }
        constexpr int sh = kBase + IN_DEPTH - OUT_DEPTH;
        ROW(uint16_t, 0, y)[x] = (uint16_t) ((UG_RGB_TO_Y(r, g, b) >> sh) + (1 << (OUT_DEPTH - 4)));
        ROW(uint16_t, 1, y)[x] = (uint16_t) ((UG_RGB_TO_CB(r, g, b) >> sh) + (1 << (OUT_DEPTH - 1)));
        ROW(uint16_t, 2, y)[x] = (uint16_t) ((UG_RGB_TO_CR(r, g, b) >> sh) + (1 << (OUT_DEPTH - 1)));
}

template <int DEPTH>
__global__ void k_r10k_to_gbrpXX(const Args a) // r10k_to_gbrpXXle :1373-1393
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        int r, g, b;
        r10k_px(BUF(const uint8_t, y) + 4 * x, r, g, b);
        ROW(uint16_t, 2, y)[x] = (uint16_t) (r << (DEPTH - 10));
        ROW(uint16_t, 0, y)[x] = (uint16_t) (g << (DEPTH - 10));
        ROW(uint16_t, 1, y)[x] = (uint16_t) (b << (DEPTH - 10));
}

__global__ void k_r10k_to_x2rgb10le(const Args a) // :1338-1348: htonl(word) >> 2
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        ROW(uint32_t, 0, y)[x] = __builtin_bswap32(BUF(const uint32_t, y)[x]) >> 2;
}

__global__ void k_rg48_to_gbrp12le(const Args a) // :1424-1438
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint16_t *src = BUF(const uint16_t, y) + 3 * x;
        ROW(uint16_t, 2, y)[x] = src[0] >> 4;
        ROW(uint16_t, 0, y)[x] = src[1] >> 4;
        ROW(uint16_t, 1, y)[x] = src[2] >> 4;
}

// 36 bytes of R12L = 8 pixels of 12-bit r, g, b in a little-endian bit stream (to_lavc_vid_conv.c:796-860 spells the bytes out)
__device__ __forceinline__ void r12l_unpack(const uint8_t *src, int (&c)[8][3])
{
        uint32_t w[9];
        if (((uintptr_t) src & 3) == 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) w[i] = ((const uint32_t *) src)[i];
        } else {
#pragma unroll
                for (int i = 0; i < 9; i++) w[i] = src[4 * i] | src[4 * i + 1] << 8 | src[4 * i + 2] << 16 | (uint32_t) src[4 * i + 3] << 24;
        }
#pragma unroll
        for (int k = 0; k < 24; k++) {
                const int bit = 12 * k, wi = bit / 32, sh = bit % 32;
                uint32_t v = w[wi] >> sh;
                if (sh > 20) v |= w[(wi + 1) % 9] << (32 - sh);
                c[k / 3][k % 3] = (int) (v & 0xfffu);
        }
}

// r12l_to_yuv4XXpYYle :761-866: one lane per 8-pixel group (whole groups, also past `width`); 4:2:2 takes the chroma of even pixels
template <int DEPTH, bool OUT_422>
__global__ void k_r12l_to_yuv(const Args a)
{
        UG_XY();
        if (x >= (a.w + 7) / 8 || y >= a.h) return;
        int c[8][3];
        r12l_unpack(BUF(const uint8_t, y) + 36 * x, c);
        constexpr int sh = kBase + 12 - DEPTH;
        uint32_t Y[8], cb[8], cr[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const int r = c[i][0], g = c[i][1], b = c[i][2];
                Y[i] = (uint16_t) ((UG_RGB_TO_Y(r, g, b) >> sh) + (1 << (DEPTH - 4)));
                cb[i] = (uint16_t) ((UG_RGB_TO_CB(r, g, b) >> sh) + (1 << (DEPTH - 1)));
                cr[i] = (uint16_t) ((UG_RGB_TO_CR(r, g, b) >> sh) + (1 << (DEPTH - 1)));
        }
        st16<8>(ROW(uint16_t, 0, y) + 8 * x, Y, 0, room16(a.ls[0], 8L * x, 8));
        if (OUT_422) {
                const uint32_t cb4[4] = { cb[0], cb[2], cb[4], cb[6] }, cr4[4] = { cr[0], cr[2], cr[4], cr[6] };
                st16<4>(ROW(uint16_t, 1, y) + 4 * x, cb4, 0, room16(a.ls[1], 4L * x, 4));
                st16<4>(ROW(uint16_t, 2, y) + 4 * x, cr4, 0, room16(a.ls[2], 4L * x, 4));
        } else {
                st16<8>(ROW(uint16_t, 1, y) + 8 * x, cb, 0, room16(a.ls[1], 8L * x, 8));
                st16<8>(ROW(uint16_t, 2, y) + 8 * x, cr, 0, room16(a.ls[2], 8L * x, 8));
        }
}

// r12l_to_p210le :899-1015 (the 10-bit coefficient set with 16-bit shifts and offsets, as written there), r12l_to_ayuv64le :1019-1130
template <bool AYUV>
__global__ void k_r12l_to_p210_ayuv64(const Args a)
{
        UG_XY();
        if (x >= (a.w + 7) / 8 || y >= a.h) return;
        int c[8][3];
        r12l_unpack(BUF(const uint8_t, y) + 36 * x, c);
        constexpr int sh = kBase + 12 - 16;
        uint32_t Y[8], cb[8], cr[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const int r = c[i][0], g = c[i][1], b = c[i][2];
                Y[i] = (uint16_t) ((UG_RGB_TO_Y(r, g, b) >> sh) + (1 << 12));
                cb[i] = (uint16_t) ((UG_RGB_TO_CB(r, g, b) >> sh) + (1 << 15));
                cr[i] = (uint16_t) ((UG_RGB_TO_CR(r, g, b) >> sh) + (1 << 15));
        }
        if (AYUV) {
                uint32_t o[32];
#pragma unroll
                for (int i = 0; i < 8; i++) o[4 * i] = 0xffff, o[4 * i + 1] = Y[i], o[4 * i + 2] = cb[i], o[4 * i + 3] = cr[i];
                st16<32>(ROW(uint16_t, 0, y) + 32 * x, o, 0, room16(a.ls[0], 32L * x, 32));
        } else {
                const uint32_t cc[8] = { cb[0], cr[0], cb[2], cr[2], cb[4], cr[4], cb[6], cr[6] };
                st16<8>(ROW(uint16_t, 0, y) + 8 * x, Y, 0, room16(a.ls[0], 8L * x, 8));
                st16<8>(ROW(uint16_t, 1, y) + 8 * x, cc, 0, room16(a.ls[1], 8L * x, 8));
        }
}

// ================================ AVFrame -> UltraGrid codec (from_lavc_vid_conv.c) ================================================
// a.d / a.ls = the decoder's planes, a.buf / a.pitch = output

// yuv420p_to_v210 :554-620, yuv422p_to_v210 :622-661, yuv420p10le_to_v210 :1007-1074, p010le_to_v210 :1498-1566, p210le_to_v210 :1450-1496
enum { S_420P8, S_422P8, S_420P10, S_P010, S_P210 };
template <int SRC>
__global__ void k_planar_to_v210(const Args a)
{
        UG_XY();
        constexpr bool k420 = SRC == S_420P8 || SRC == S_420P10 || SRC == S_P010;
        if (x >= a.w / 6 || y >= (k420 ? a.h / 2 : a.h)) return;
#pragma unroll
        for (int l = 0; l < (k420 ? 2 : 1); l++) {
                const int row = k420 ? 2 * y + l : y;
                uint32_t Y[6], cb[3], cr[3];
                if (SRC == S_420P8 || SRC == S_422P8) {
                        const uint8_t *sy = ROW(const uint8_t, 0, row) + 6 * x, *scb = ROW(const uint8_t, 1, y) + 3 * x, *scr = ROW(const uint8_t, 2, y) + 3 * x;
#pragma unroll
                        for (int i = 0; i < 6; i++) Y[i] = (uint32_t) sy[i] << 2;
                        if (SRC == S_420P8) Y[4] = sy[4]; // `w0_3 = *src_y1++;` -- not shifted in the reference (:599-600)
#pragma unroll
                        for (int i = 0; i < 3; i++) cb[i] = (uint32_t) scb[i] << 2, cr[i] = (uint32_t) scr[i] << 2;
                } else if (SRC == S_420P10) {
                        const uint16_t *sy = ROW(const uint16_t, 0, row) + 6 * x, *scb = ROW(const uint16_t, 1, y) + 3 * x, *scr = ROW(const uint16_t, 2, y) + 3 * x;
#pragma unroll
                        for (int i = 0; i < 6; i++) Y[i] = sy[i];
#pragma unroll
                        for (int i = 0; i < 3; i++) cb[i] = scb[i], cr[i] = scr[i];
                } else { // P010 / P210: samples in the high bits, Cb Cr interleaved
                        const uint16_t *sy = ROW(const uint16_t, 0, row) + 6 * x, *sc = ROW(const uint16_t, 1, y) + 6 * x;
#pragma unroll
                        for (int i = 0; i < 6; i++) Y[i] = sy[i] >> 6;
#pragma unroll
                        for (int i = 0; i < 3; i++) cb[i] = sc[2 * i] >> 6, cr[i] = sc[2 * i + 1] >> 6;
                }
                st4(BUF(uint32_t, row) + 4 * x, v210w(cb[0], Y[0], cr[0]), v210w(Y[1], cb[1], Y[2]), v210w(cr[1], Y[3], cb[2]), v210w(Y[4], cr[2], Y[5]));
        }
}

// yuv444p_to_v210 :714-762 (8 bit), yuv444p1Xle_to_v210 :1079-1123 (10 / 12 / 16 bit)
template <int DEPTH>
__global__ void k_yuv444_to_v210(const Args a)
{
        UG_XY();
        if (x >= a.w / 6 || y >= a.h) return;
        uint32_t Y[6], cb[3], cr[3];
        if (DEPTH == 8) {
                const uint8_t *sy = ROW(const uint8_t, 0, y) + 6 * x, *scb = ROW(const uint8_t, 1, y) + 6 * x, *scr = ROW(const uint8_t, 2, y) + 6 * x;
#pragma unroll
                for (int i = 0; i < 6; i++) Y[i] = (uint32_t) sy[i] << 2;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        cb[i] = (((uint32_t) scb[2 * i] << 2) + ((uint32_t) scb[2 * i + 1] << 2)) / 2;
                        cr[i] = (((uint32_t) scr[2 * i] << 2) + ((uint32_t) scr[2 * i + 1] << 2)) / 2;
                }
        } else {
                constexpr int sh = DEPTH - 10;
                const uint16_t *sy = ROW(const uint16_t, 0, y) + 6 * x, *scb = ROW(const uint16_t, 1, y) + 6 * x, *scr = ROW(const uint16_t, 2, y) + 6 * x;
#pragma unroll
                for (int i = 0; i < 6; i++) Y[i] = (uint32_t) sy[i] >> sh;
                Y[1] = sy[1], Y[4] = sy[4]; // `w0_1 = *src_y++;`, `w0_3 = *src_y++;` -- not shifted in the reference (:1099,1107)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        cb[i] = (((uint32_t) scb[2 * i] >> sh) + ((uint32_t) scb[2 * i + 1] >> sh)) / 2;
                        cr[i] = (((uint32_t) scr[2 * i] >> sh) + ((uint32_t) scr[2 * i + 1] >> sh)) / 2;
                }
        }
        st4(BUF(uint32_t, y) + 4 * x, v210w(cb[0], Y[0], cr[0]), v210w(Y[1], cb[1], Y[2]), v210w(cr[1], Y[3], cb[2]), v210w(Y[4], cr[2], Y[5]));
}

// -> UYVY, one lane per pixel pair.  yuv444p_to_uyvy :663-685, yuv444p16le_to_uyvy :687-712 (high bytes), yuv444p1Xle_to_uyvy :1183-1206,
// yuv420p10le_to_uyvy :1143-1178, p010le_to_uyvy :1568-1603, nv12_to_uyvy :150-169
enum { U_444P8, U_444P16, U_444P10, U_444P12, U_420P10, U_P010, U_NV12 };
template <int SRC>
__global__ void k_to_uyvy(const Args a)
{
        UG_XY();
        constexpr bool kPairRows = SRC == U_420P10 || SRC == U_P010;
        if (x >= a.w / 2 || y >= (kPairRows ? a.h / 2 : a.h)) return;
        if (SRC == U_444P8) {
                const uint8_t *sy = ROW(const uint8_t, 0, y) + 2 * x, *scb = ROW(const uint8_t, 1, y) + 2 * x, *scr = ROW(const uint8_t, 2, y) + 2 * x;
                BUF(uint32_t, y)[x] = (uint32_t) ((scb[0] + scb[1]) / 2) | (uint32_t) sy[0] << 8 | (uint32_t) ((scr[0] + scr[1]) / 2) << 16 | (uint32_t) sy[1] << 24;
        } else if (SRC == U_444P16) {
                const uint16_t *sy = ROW(const uint16_t, 0, y) + 2 * x, *scb = ROW(const uint16_t, 1, y) + 2 * x, *scr = ROW(const uint16_t, 2, y) + 2 * x;
                BUF(uint32_t, y)[x] = (uint32_t) (((scb[0] >> 8) + (scb[1] >> 8)) / 2) | (uint32_t) (sy[0] >> 8) << 8 |
                                      (uint32_t) (((scr[0] >> 8) + (scr[1] >> 8)) / 2) << 16 | (uint32_t) (sy[1] >> 8) << 24;
        } else if (SRC == U_444P10 || SRC == U_444P12) {
                constexpr int sh = SRC == U_444P10 ? 2 : 4;
                const uint16_t *sy = ROW(const uint16_t, 0, y) + 2 * x, *scb = ROW(const uint16_t, 1, y) + 2 * x, *scr = ROW(const uint16_t, 2, y) + 2 * x;
                BUF(uint32_t, y)[x] = (uint32_t) (((scb[0] + scb[1] + 1) / 2 >> sh) & 0xff) | (uint32_t) ((sy[0] >> sh) & 0xff) << 8 |
                                      (uint32_t) (((scr[0] + scr[1] + 1) / 2 >> sh) & 0xff) << 16 | (uint32_t) ((sy[1] >> sh) & 0xff) << 24;
        } else if (SRC == U_420P10) {
                const uint32_t u = (ROW(const uint16_t, 1, y)[x] >> 2) & 0xff, v = (ROW(const uint16_t, 2, y)[x] >> 2) & 0xff;
#pragma unroll
                for (int l = 0; l < 2; l++) {
                        const uint16_t *sy = ROW(const uint16_t, 0, 2 * y + l) + 2 * x;
                        BUF(uint32_t, 2 * y + l)[x] = u | (uint32_t) ((sy[0] >> 2) & 0xff) << 8 | v << 16 | (uint32_t) ((sy[1] >> 2) & 0xff) << 24;
                }
        } else if (SRC == U_P010) {
                const uint16_t *sc = ROW(const uint16_t, 1, y) + 2 * x;
                const uint32_t u = sc[0] >> 8, v = sc[1] >> 8;
#pragma unroll
                for (int l = 0; l < 2; l++) {
                        const uint16_t *sy = ROW(const uint16_t, 0, 2 * y + l) + 2 * x;
                        BUF(uint32_t, 2 * y + l)[x] = u | (uint32_t) (sy[0] >> 8) << 8 | v << 16 | (uint32_t) (sy[1] >> 8) << 24;
                }
        } else { // U_NV12
                const uint8_t *sy = ROW(const uint8_t, 0, y) + 2 * x, *sc = ROW(const uint8_t, 1, y / 2) + 2 * x;
                BUF(uint32_t, y)[x] = (uint32_t) sc[0] | (uint32_t) sy[0] << 8 | (uint32_t) sc[1] << 16 | (uint32_t) sy[1] << 24;
        }
}

__global__ void k_p210le_to_uyvy(const Args a) // :1605-1628: `*dst = Cb; *dst++ = Y0; *dst++ = Cr; *dst++ = Y1` -- three bytes per pair
{
        UG_XY();
        if (x >= a.w / 2 || y >= a.h) return;
        const uint16_t *sy = ROW(const uint16_t, 0, y) + 2 * x, *sc = ROW(const uint16_t, 1, y) + 2 * x;
        uint8_t *dst = BUF(uint8_t, y) + 3 * x;
        dst[0] = (uint8_t) (sy[0] >> 8);
        dst[1] = (uint8_t) (sc[1] >> 8);
        dst[2] = (uint8_t) (sy[1] >> 8);
}

__device__ __forceinline__ void put_rgb8(const Args &a, uint8_t *row, int px, bool rgba, int r, int g, int b)
{
        if (rgba) {
                ((uint32_t *) row)[px] = mk_rgba(a, r, g, b);
        } else {
                row[3 * px] = (uint8_t) clamp_full(r, 8), row[3 * px + 1] = (uint8_t) clamp_full(g, 8), row[3 * px + 2] = (uint8_t) clamp_full(b, 8);
        }
}

// yuv8p_to_rgb :837-919 (4:2:0 / 4:2:2 planar 8 bit -> RGB / RGBA): one lane per pixel pair of a line pair, height / 2 line pairs
template <int SUB, bool RGBA>
__global__ void k_yuv8p_to_rgb(const Args a)
{
        UG_XY();
        if (x >= a.w / 2 || y >= a.h / 2) return;
#pragma unroll
        for (int l = 0; l < 2; l++) {
                const int row = 2 * y + l, crow = SUB == 420 ? y : row;
                const int cb = ROW(const uint8_t, 1, crow)[x] - 128, cr = ROW(const uint8_t, 2, crow)[x] - 128;
                const uint8_t *sy = ROW(const uint8_t, 0, row) + 2 * x;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                        const int ys = (sy[i] - 16) * a.c[Y_SCALE];
                        put_rgb8(a, BUF(uint8_t, row), 2 * x + i, RGBA, (ys + cr * a.c[R_CR]) >> kBase, (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> kBase, (ys + cb * a.c[B_CB]) >> kBase);
                }
        }
}

template <bool RGBA>
__global__ void k_yuv444p_to_rgb(const Args a) // :953-993 (no luma offset, RGB clamped to 1..254)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const int cb = ROW(const uint8_t, 1, y)[x] - 128, cr = ROW(const uint8_t, 2, y)[x] - 128, ys = ROW(const uint8_t, 0, y)[x] * a.c[Y_SCALE];
        put_rgb8(a, BUF(uint8_t, y), x, RGBA, (ys + cr * a.c[R_CR]) >> kBase, (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> kBase, (ys + cb * a.c[B_CB]) >> kBase);
}

template <bool RGBA>
__global__ void k_nv12_to_rgb(const Args a) // :770-819: the second pixel of a pair repeats the first one's r, g, b
{
        UG_XY();
        if (x >= a.w / 2 || y >= a.h) return;
        const uint8_t *sc = ROW(const uint8_t, 1, y / 2) + 2 * x;
        const int cb = sc[0] - 128, cr = sc[1] - 128, ys = (ROW(const uint8_t, 0, y)[2 * x] - 16) * a.c[Y_SCALE];
        const int r = (ys + cr * a.c[R_CR]) >> kBase, g = (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> kBase, b = (ys + cb * a.c[B_CB]) >> kBase;
        put_rgb8(a, BUF(uint8_t, y), 2 * x, RGBA, r, g, b);
        put_rgb8(a, BUF(uint8_t, y), 2 * x + 1, RGBA, r, g, b);
}

template <bool RGBA>
__global__ void k_gbrp_to_rgb(const Args a) // gbrp_to_rgb :221-237, gbrp_to_rgba :239-262: every plane indexed with linesize[0]
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const long idx = (long) y * a.ls[0] + x;
        const uint32_t r = a.d[2][idx], g = a.d[0][idx], b = a.d[1][idx];
        if (RGBA) {
                BUF(uint32_t, y)[x] = a.am | r << a.rs | g << a.gs | b << a.bs;
        } else {
                uint8_t *o = BUF(uint8_t, y) + 3 * x;
                o[0] = (uint8_t) r, o[1] = (uint8_t) g, o[2] = (uint8_t) b;
        }
}

// yuvp10le_to_rgb :1273-1374 (4:2:0 / 4:2:2 planar 10 bit -> RGB 24 / R10k 30 / RGBA 32)
template <int SUB, int OUT_BITS>
__global__ void k_yuvp10le_to_rgb(const Args a)
{
        UG_XY();
        if (x >= a.w / 2 || y >= a.h / 2) return;
        constexpr int bpp = OUT_BITS == 30 ? 10 : 8;
        constexpr int sh = kBase + (10 - bpp);
#pragma unroll
        for (int l = 0; l < 2; l++) {
                const int row = 2 * y + l, crow = SUB == 420 ? y : row;
                const int cr = ROW(const uint16_t, 2, crow)[x] - (1 << 9), cb = ROW(const uint16_t, 1, crow)[x] - (1 << 9);
                const int rr = (cr * a.c[R_CR]) >> sh, gg = (cb * a.c[G_CB] + cr * a.c[G_CR]) >> sh, bb = (cb * a.c[B_CB]) >> sh;
                const uint16_t *sy = ROW(const uint16_t, 0, row) + 2 * x;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                        const int ys = (a.c[Y_SCALE] * (sy[i] - (1 << 6))) >> sh;
                        const uint32_t r = clamp_full(ys + rr, bpp), g = clamp_full(ys + gg, bpp), b = clamp_full(ys + bb, bpp);
                        const int px = 2 * x + i;
                        if (OUT_BITS == 32) {
                                BUF(uint32_t, row)[px] = a.am | (r << a.rs | g << a.gs | b << a.bs);
                        } else if (OUT_BITS == 24) {
                                uint8_t *o = BUF(uint8_t, row) + 3 * px;
                                o[0] = (uint8_t) r, o[1] = (uint8_t) g, o[2] = (uint8_t) b;
                        } else {
                                // R10k: big-endian 10-bit r, g, b, then two padding bits set
                                const uint32_t b0 = r >> 2, b1 = (r & 3u) << 6 | g >> 4, b2 = (g & 0xfu) << 4 | b >> 6, b3 = (b & 0x3fu) << 2 | 3u;
                                BUF(uint32_t, row)[px] = b0 | b1 << 8 | b2 << 16 | b3 << 24;
                        }
                }
        }
}

template <bool RGBA>
__global__ void k_yuv444p10le_to_rgb(const Args a) // :1393-1435
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const int cb = ROW(const uint16_t, 1, y)[x] - (1 << 9), cr = ROW(const uint16_t, 2, y)[x] - (1 << 9);
        const int ys = (ROW(const uint16_t, 0, y)[x] - (1 << 6)) * a.c[Y_SCALE];
        put_rgb8(a, BUF(uint8_t, y), x, RGBA, (ys + cr * a.c[R_CR]) >> (kBase + 2), (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> (kBase + 2), (ys + cb * a.c[B_CB]) >> (kBase + 2));
}

// yuv444pXXle_to_r10k :295-334, _to_r12l :363-437, _to_rg48 :464-499: planar 4:4:4 at 10 / 12 / 16 bit -> packed RGB at 10 / 12 / 16 bit.
// The sums are 32-bit and wrap for extreme 16-bit samples exactly as the reference's comp_type_t arithmetic does on this compiler.
template <int DEPTH, int OUT_BITS>
__device__ __forceinline__ void yuv444_px_to_rgb(const Args &a, int row, int col, int &r, int &g, int &b)
{
        const int ys = a.c[Y_SCALE] * (ROW(const uint16_t, 0, row)[col] - (1 << (DEPTH - 4)));
        const int cr = ROW(const uint16_t, 2, row)[col] - (1 << (DEPTH - 1)), cb = ROW(const uint16_t, 1, row)[col] - (1 << (DEPTH - 1));
        constexpr int sh = kBase - OUT_BITS + DEPTH;
        r = clamp_full((ys + cr * a.c[R_CR]) >> sh, OUT_BITS);
        g = clamp_full((ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> sh, OUT_BITS);
        b = clamp_full((ys + cb * a.c[B_CB]) >> sh, OUT_BITS);
}

template <int DEPTH>
__global__ void k_yuv444pXX_to_r10k(const Args a)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        int r, g, b;
        yuv444_px_to_rgb<DEPTH, 10>(a, y, x, r, g, b);
        BUF(uint32_t, y)[x] = (uint32_t) (r >> 2) | (uint32_t) (((r & 0x3) << 6 | g >> 4) & 0xff) << 8 | (uint32_t) (((g & 0xF) << 4 | b >> 6) & 0xff) << 16 |
                              (uint32_t) (((b & 0x3F) << 2 | 0x3) & 0xff) << 24;
}

template <int DEPTH>
__global__ void k_yuv444pXX_to_rg48(const Args a)
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        int r, g, b;
        yuv444_px_to_rgb<DEPTH, 16>(a, y, x, r, g, b);
        uint16_t *dst = BUF(uint16_t, y) + 3 * x;
        dst[0] = (uint16_t) r, dst[1] = (uint16_t) g, dst[2] = (uint16_t) b;
}

template <int DEPTH>
__global__ void k_yuv444pXX_to_r12l(const Args a) // whole groups of 8, also past `width` (the reference reads and writes them too)
{
        UG_XY();
        if (x >= (a.w + 7) / 8 || y >= a.h) return;
        uint32_t v[24], bytes[36];
#pragma unroll
        for (int j = 0; j < 8; j++) {
                int r, g, b;
                yuv444_px_to_rgb<DEPTH, 12>(a, y, 8 * x + j, r, g, b);
                v[3 * j] = r, v[3 * j + 1] = g, v[3 * j + 2] = b;
        }
#pragma unroll
        for (int k = 0; k < 12; k++) { // :391-433 byte by byte = a little-endian stream of 12-bit values
                const uint32_t e = v[2 * k], o = v[2 * k + 1];
                bytes[3 * k] = e & 0xff, bytes[3 * k + 1] = ((o & 0xf) << 4 | e >> 8) & 0xff, bytes[3 * k + 2] = (o >> 4) & 0xff;
        }
        uint8_t *dst = BUF(uint8_t, y) + 36 * x;
        if (((uintptr_t) dst & 3) == 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) ((uint32_t *) dst)[i] = bytes[4 * i] | bytes[4 * i + 1] << 8 | bytes[4 * i + 2] << 16 | bytes[4 * i + 3] << 24;
        } else {
#pragma unroll
                for (int i = 0; i < 36; i++) dst[i] = (uint8_t) bytes[i];
        }
}

template <int DEPTH>
__global__ void k_yuv444pXX_to_y416(const Args a) // yuv444p1Xle_to_y416 :1223-1248
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        uint16_t *dst = BUF(uint16_t, y) + 4 * x;
        dst[0] = (uint16_t) (ROW(const uint16_t, 1, y)[x] << (16 - DEPTH));
        dst[1] = (uint16_t) (ROW(const uint16_t, 0, y)[x] << (16 - DEPTH));
        dst[2] = (uint16_t) (ROW(const uint16_t, 2, y)[x] << (16 - DEPTH));
        dst[3] = 0xFFFF;
}

__global__ void k_xv30_to_uyvy(const Args a) // :1631-1658: pairs, then a lone last pixel as U Y V 0
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint32_t *src = ROW(const uint32_t, 0, y) + 2 * x;
        const uint32_t in1 = src[0];
        if (2 * x + 1 < a.w) {
                const uint32_t in2 = src[1];
                const uint32_t u = ((((in1 >> 2) & 0xFFu) + (((in2 >> 2) & 0xFFu) + 1)) >> 1) & 0xFFu, v = ((((in1 >> 22) & 0xFFu) + (((in2 >> 22) & 0xFFu) + 1)) >> 1) & 0xFFu;
                BUF(uint32_t, y)[x] = u | ((in1 >> 12) & 0xFFu) << 8 | v << 16 | ((in2 >> 12) & 0xFFu) << 24;
        } else {
                BUF(uint32_t, y)[x] = ((in1 >> 2U) & 0xFFU) | ((in1 >> 12U) & 0xFFU) << 8 | ((in1 >> 22U) & 0xFFU) << 16;
        }
}

// xv30_to_v210 :1660-1698 (width / 6 groups), y210_to_v210 :1724-1760 ((width + 5) / 6 groups)
template <bool XV30>
__global__ void k_packed_to_v210(const Args a)
{
        UG_XY();
        if (x >= (XV30 ? a.w / 6 : (a.w + 5) / 6) || y >= a.h) return;
        uint32_t u[3], y0[3], v[3], y1[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
                if (XV30) {
                        const uint32_t *src = ROW(const uint32_t, 0, y) + 6 * x + 2 * i;
                        const uint32_t in0 = src[0], in1 = src[1];
                        u[i] = ((in0 & 0x3FFU) + (in1 & 0x3FFU) + 1) >> 1;
                        y0[i] = (in0 >> 10U) & 0x3FFU;
                        v[i] = ((in0 >> 20U & 0x3FFU) + ((in1 >> 20U & 0x3FFU) + 1)) >> 1;
                        y1[i] = (in1 >> 10U) & 0x3FFU;
                } else {
                        const uint16_t *src = ROW(const uint16_t, 0, y) + 12 * x + 4 * i;
                        y0[i] = src[0] >> 6, u[i] = src[1] >> 6, y1[i] = src[2] >> 6, v[i] = src[3] >> 6;
                }
        }
        st4(BUF(uint32_t, y) + 4 * x, v[0] << 20U | y0[0] << 10U | u[0], y0[1] << 20U | u[1] << 10U | y1[0], u[2] << 20U | (y1[1] << 10U | v[1]),
            y1[2] << 20U | v[2] << 10U | y0[2]);
}

__global__ void k_xv30_to_y416(const Args a) // :1700-1720
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint32_t in = ROW(const uint32_t, 0, y)[x];
        uint16_t *dst = BUF(uint16_t, y) + 4 * x;
        dst[0] = (in & 0x3FFU) << 6U, dst[1] = ((in >> 10U) & 0x3FFU) << 6U, dst[2] = ((in >> 20U) & 0x3FFU) << 6U, dst[3] = 0xFFFFU;
}

__global__ void k_y210_to_y416(const Args a) // :1762-1792, (width + 1) / 2 pairs, two whole pixels each
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint16_t *src = ROW(const uint16_t, 0, y) + 4 * x;
        uint16_t *dst = BUF(uint16_t, y) + 8 * x;
        dst[0] = src[1], dst[1] = src[0], dst[2] = src[3], dst[3] = 0xFFFFU;
        if (room16(a.pitch, 8L * x, 8) == 8) dst[4] = src[1], dst[5] = src[2], dst[6] = src[3], dst[7] = 0xFFFFU; // second pixel of a lone last pair

}

__global__ void k_y210_to_uyvy(const Args a) // :1794-1818: the high bytes
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint8_t *src = ROW(const uint8_t, 0, y) + 8 * x;
        BUF(uint32_t, y)[x] = (uint32_t) src[3] | (uint32_t) src[1] << 8 | (uint32_t) src[7] << 16 | (uint32_t) src[5] << 24;
}

__global__ void k_ayuv64_to_y416(const Args a) // :1904-1925
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint16_t *src = ROW(const uint16_t, 0, y) + 4 * x;
        uint16_t *dst = BUF(uint16_t, y) + 4 * x;
        dst[0] = src[2], dst[1] = src[1], dst[2] = src[3], dst[3] = src[0];
}

__global__ void k_ayuv64_to_v210(const Args a) // :1927-1967: six A Y U V pixels -> one v210 group
{
        UG_XY();
        if (x >= (a.w + 5) / 6 || y >= a.h) return;
        const uint16_t *px = ROW(const uint16_t, 0, y) + 24 * x; // pixel i: A = px[4i], Y = px[4i + 1], U = px[4i + 2], V = px[4i + 3]
        uint32_t Y[6], U[3], V[3];
#pragma unroll
        for (int i = 0; i < 6; i++) Y[i] = px[4 * i + 1] >> 6;
        // the luma of the 2nd and 5th pixel enters its word with all 16 bits (`w = src[5]`, `w = src[1]` there), spilling into the fields above
        Y[1] = px[4 * 1 + 1], Y[4] = px[4 * 4 + 1];
#pragma unroll
        for (int k = 0; k < 3; k++) {
                U[k] = ((px[8 * k + 2] >> 6) + (px[8 * k + 6] >> 6)) / 2;
                V[k] = ((px[8 * k + 3] >> 6) + (px[8 * k + 7] >> 6)) / 2;
        }
        st4(BUF(uint32_t, y) + 4 * x, v210w(U[0], Y[0], V[0]), v210w(Y[1], U[1], Y[2]), v210w(V[1], Y[3], U[2]), v210w(Y[4], V[2], Y[5]));
}

__global__ void k_vuya_to_uyvy(const Args a) // :1971-1996
{
        UG_XY();
        if (x >= (a.w + 1) / 2 || y >= a.h) return;
        const uint8_t *src = ROW(const uint8_t, 0, y) + 8 * x;
        if (2 * x + 1 < a.w) {
                BUF(uint32_t, y)[x] = ((src[1] + src[5] + 1U) >> 1U) | (uint32_t) src[2] << 8 | ((src[0] + src[4] + 1U) >> 1U) << 16 | (uint32_t) src[6] << 24;
        } else {
                BUF(uint32_t, y)[x] = (uint32_t) src[1] | (uint32_t) src[2] << 8 | (uint32_t) src[0] << 16;
        }
}

template <bool ALPHA>
__global__ void k_vuyax_to_y416(const Args a) // :2001-2015
{
        UG_XY();
        if (x >= a.w || y >= a.h) return;
        const uint8_t *src = ROW(const uint8_t, 0, y) + 4 * x;
        uint16_t *dst = BUF(uint16_t, y) + 4 * x;
        dst[0] = src[1] << 8U, dst[1] = src[2] << 8U, dst[2] = src[0] << 8U, dst[3] = ALPHA ? src[3] << 8U : 0xFFFF;
}

// ---- 8-pixel-per-lane variants of the most used 8-bit conversions (same arithmetic, 64-128 bit accesses) ----------------------------------
// Taken when every pointer and line size is a multiple of 16 and the lane count divides the line (launch()); otherwise the kernels above.

__device__ __forceinline__ uint32_t byte_of(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }

// The W output words of unit x of a packed line (the fast variants: blockDim = (64, 4), a wave = 64 consecutive units, lanes past the line
// have returned, 16-byte aligned lines): the wave's units leave as one contiguous run of whole lines, streamed (ug::WaveWords)
template <int W>
__device__ __forceinline__ void store_unit(uint8_t *row, int x, int nunits, const uint32_t (&w)[W])
{
        using WS = ug::WaveWords<W>;
        __shared__ uint32_t lds[WS::LDS_DWORDS ? 4 * WS::LDS_DWORDS : 1];
        const int lane = threadIdx.x, x0 = x - lane, units = min(64, nunits - x0);
        WS::store(row + (long) x0 * (4 * W), w, lds + threadIdx.y * WS::LDS_DWORDS, lane, units, units);
}

template <bool RGBA>
__device__ __forceinline__ void store_px8(const Args &a, uint8_t *row, int x, const int (&r)[8], const int (&g)[8], const int (&b)[8])
{
        if (RGBA) {
                uint32_t w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) w[i] = mk_rgba(a, r[i], g[i], b[i]);
                store_unit<8>(row, x, a.w / 8, w);
        } else {
                uint32_t by[24], w[6];
#pragma unroll
                for (int i = 0; i < 8; i++) by[3 * i] = clamp_full(r[i], 8), by[3 * i + 1] = clamp_full(g[i], 8), by[3 * i + 2] = clamp_full(b[i], 8);
#pragma unroll
                for (int i = 0; i < 6; i++) w[i] = by[4 * i] | by[4 * i + 1] << 8 | by[4 * i + 2] << 16 | by[4 * i + 3] << 24;
                store_unit<6>(row, x, a.w / 8, w);
        }
}

template <int SUB, bool RGBA>
__global__ void k_yuv8p_to_rgb_x8(const Args a) // yuv8p_to_rgb, 4 pixel pairs of a line pair per lane
{
        UG_XY();
        if (x >= a.w / 8 || y >= a.h / 2) return;
#pragma unroll
        for (int l = 0; l < 2; l++) {
                const int row = 2 * y + l, crow = SUB == 420 ? y : row;
                const uint2 yy = *(const uint2 *) (ROW(const uint8_t, 0, row) + 8 * x);
                const uint32_t cb4 = *(const uint32_t *) (ROW(const uint8_t, 1, crow) + 4 * x), cr4 = *(const uint32_t *) (ROW(const uint8_t, 2, crow) + 4 * x);
                int r[8], g[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        const int cb = (int) byte_of(cb4, i / 2) - 128, cr = (int) byte_of(cr4, i / 2) - 128;
                        const int ys = ((int) byte_of(i < 4 ? yy.x : yy.y, i % 4) - 16) * a.c[Y_SCALE];
                        r[i] = (ys + cr * a.c[R_CR]) >> kBase, g[i] = (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> kBase, b[i] = (ys + cb * a.c[B_CB]) >> kBase;
                }
                store_px8<RGBA>(a, BUF(uint8_t, row), x, r, g, b);
        }
}

template <int OUT> // 0 UYVY, 1 RGB, 2 RGBA
__global__ void k_nv12_x8(const Args a) // nv12_to_uyvy / nv12_to_rgb, 4 pixel pairs per lane
{
        UG_XY();
        if (x >= a.w / 8 || y >= a.h) return;
        const uint2 yy = *(const uint2 *) (ROW(const uint8_t, 0, y) + 8 * x), cc = *(const uint2 *) (ROW(const uint8_t, 1, y / 2) + 8 * x);
        if (OUT == 0) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        const uint32_t c2 = i < 2 ? cc.x : cc.y, y2 = i < 2 ? yy.x : yy.y;
                        w[i] = byte_of(c2, 2 * (i % 2)) | byte_of(y2, 2 * (i % 2)) << 8 | byte_of(c2, 2 * (i % 2) + 1) << 16 | byte_of(y2, 2 * (i % 2) + 1) << 24;
                }
                ug::st_stream((uint4 *) (BUF(uint8_t, y) + 16L * x), make_uint4(w[0], w[1], w[2], w[3]));
        } else {
                int r[8], g[8], b[8];
#pragma unroll
                for (int i = 0; i < 4; i++) { // both pixels of a pair take the first one's colour (:797-811)
                        const uint32_t c2 = i < 2 ? cc.x : cc.y, y2 = i < 2 ? yy.x : yy.y;
                        const int cb = (int) byte_of(c2, 2 * (i % 2)) - 128, cr = (int) byte_of(c2, 2 * (i % 2) + 1) - 128;
                        const int ys = ((int) byte_of(y2, 2 * (i % 2)) - 16) * a.c[Y_SCALE];
                        r[2 * i] = r[2 * i + 1] = (ys + cr * a.c[R_CR]) >> kBase;
                        g[2 * i] = g[2 * i + 1] = (ys + cb * a.c[G_CB] + cr * a.c[G_CR]) >> kBase;
                        b[2 * i] = b[2 * i + 1] = (ys + cb * a.c[B_CB]) >> kBase;
                }
                store_px8<OUT == 2>(a, BUF(uint8_t, y), x, r, g, b);
        }
}

template <int BPP>
__global__ void k_rgb_to_gbrp_x8(const Args a) // rgb_rgba_to_gbrp, 8 pixels per lane
{
        UG_XY();
        if (x >= a.w / 8 || y >= a.h) return;
        const uint32_t *src = (const uint32_t *) (a.buf + (long) y * (BPP * a.w)) + 2 * BPP * x;
        uint32_t w[2 * BPP];
#pragma unroll
        for (int i = 0; i < 2 * BPP; i++) w[i] = src[i];
        uint32_t c[3][2] = {};
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                        const int bi = BPP * i + k;
                        c[k][i / 4] |= byte_of(w[bi / 4], bi % 4) << (8 * (i % 4));
                }
        }
        ug::st_stream((uint2 *) (ROW(uint8_t, 2, y) + 8 * x), make_uint2(c[0][0], c[0][1]));
        ug::st_stream((uint2 *) (ROW(uint8_t, 0, y) + 8 * x), make_uint2(c[1][0], c[1][1]));
        ug::st_stream((uint2 *) (ROW(uint8_t, 1, y) + 8 * x), make_uint2(c[2][0], c[2][1]));
}

template <int SRC> // S_420P8 / S_422P8: four 6-pixel groups per lane
__global__ void k_planar8_to_v210_x4(const Args a)
{
        UG_XY();
        constexpr bool k420 = SRC == S_420P8;
        if (x >= a.w / 24 || y >= (k420 ? a.h / 2 : a.h)) return;
        const uint8_t *pcb = ROW(const uint8_t, 1, y) + 12 * x, *pcr = ROW(const uint8_t, 2, y) + 12 * x;
        uint32_t cbw[3], crw[3];
#pragma unroll
        for (int i = 0; i < 3; i++) cbw[i] = ((const uint32_t *) pcb)[i], crw[i] = ((const uint32_t *) pcr)[i];
#pragma unroll
        for (int l = 0; l < (k420 ? 2 : 1); l++) {
                const int row = k420 ? 2 * y + l : y;
                const uint2 *py = (const uint2 *) (ROW(const uint8_t, 0, row) + 24 * x);
                uint32_t yw[6];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        const uint2 v = py[i];
                        yw[2 * i] = v.x, yw[2 * i + 1] = v.y;
                }
                uint32_t ow[16];
#pragma unroll
                for (int gidx = 0; gidx < 4; gidx++) {
                        uint32_t Y[6], cb[3], cr[3];
#pragma unroll
                        for (int i = 0; i < 6; i++) Y[i] = byte_of(yw[(6 * gidx + i) / 4], (6 * gidx + i) % 4) << 2;
                        if (k420) Y[4] >>= 2; // unshifted in the reference (:599-600)
#pragma unroll
                        for (int i = 0; i < 3; i++) cb[i] = byte_of(cbw[(3 * gidx + i) / 4], (3 * gidx + i) % 4) << 2, cr[i] = byte_of(crw[(3 * gidx + i) / 4], (3 * gidx + i) % 4) << 2;
                        ow[4 * gidx] = v210w(cb[0], Y[0], cr[0]), ow[4 * gidx + 1] = v210w(Y[1], cb[1], Y[2]);
                        ow[4 * gidx + 2] = v210w(cr[1], Y[3], cb[2]), ow[4 * gidx + 3] = v210w(Y[4], cr[2], Y[5]);
                }
                store_unit<16>(BUF(uint8_t, row), x, a.w / 24, ow);
        }
}

// k_v210_to_planar for 4:2:0 / 4:2:2 / P210, two 6-pixel groups per lane (32 bytes in): 24 bytes of luma and 12 of each chroma plane (24 of
// interleaved chroma for P210) per line and lane, every plane's units of a wave stored as one contiguous run (store_unit)
template <int MODE>
__global__ void k_v210_to_planar_x2(const Args a)
{
        UG_XY();
        const int rows = MODE == V_420P10 ? (a.h + 1) / 2 : a.h, nunits = a.w / 12;
        if (x >= nunits || y >= rows) return;
        const int y0 = MODE == V_420P10 ? 2 * y : y, y1 = MODE == V_420P10 && 2 * y + 1 < a.h ? 2 * y + 1 : y0;
        constexpr int sh = MODE == V_P210 ? 6 : 0;
        V210Group g[2][2]; // [line][group]
#pragma unroll
        for (int k = 0; k < 2; k++) {
                g[0][k] = v210_unpack(BUF(const uint32_t, y0) + 8 * x + 4 * k);
                g[1][k] = MODE == V_420P10 ? v210_unpack(BUF(const uint32_t, y1) + 8 * x + 4 * k) : g[0][k];
        }
#pragma unroll
        for (int l = 0; l < (MODE == V_420P10 ? 2 : 1); l++) {
                if (l == 1 && y1 == y0) break; // odd height: the last line stands alone (wave-uniform)
                uint32_t yw[6];
#pragma unroll
                for (int i = 0; i < 6; i++) yw[i] = (g[l][i / 3].y[2 * (i % 3)] << sh) | (g[l][i / 3].y[2 * (i % 3) + 1] << sh) << 16;
                store_unit<6>(ROW(uint8_t, 0, l ? y1 : y0), x, nunits, yw);
        }
        if (MODE == V_P210) {
                uint32_t cw[6];
#pragma unroll
                for (int i = 0; i < 6; i++) cw[i] = (g[0][i / 3].cb[i % 3] << sh) | (g[0][i / 3].cr[i % 3] << sh) << 16;
                store_unit<6>(ROW(uint8_t, 1, y0), x, nunits, cw);
        } else {
                uint32_t cb[6], cr[6], cbw[3], crw[3];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                        cb[i] = MODE == V_420P10 ? (g[0][i / 3].cb[i % 3] + g[1][i / 3].cb[i % 3]) / 2 : g[0][i / 3].cb[i % 3];
                        cr[i] = MODE == V_420P10 ? (g[0][i / 3].cr[i % 3] + g[1][i / 3].cr[i % 3]) / 2 : g[0][i / 3].cr[i % 3];
                }
#pragma unroll
                for (int i = 0; i < 3; i++) cbw[i] = (cb[2 * i] & 0xffffu) | cb[2 * i + 1] << 16, crw[i] = (cr[2 * i] & 0xffffu) | cr[2 * i + 1] << 16;
                uint8_t *const rcb = MODE == V_420P10 ? a.d[1] + (long) a.ls[1] * y0 / 2 : (uint8_t *) ROW(uint8_t, 1, y0);
                uint8_t *const rcr = MODE == V_420P10 ? a.d[2] + (long) a.ls[2] * y0 / 2 : (uint8_t *) ROW(uint8_t, 2, y0);
                store_unit<3>(rcb, x, nunits, cbw);
                store_unit<3>(rcr, x, nunits, crw);
        }
}

// k_planar_to_v210 for the 16-bit sources (yuv420p10le, P010, P210), two 6-pixel groups per lane: the 24 luma bytes and the 12 + 12 (P010 / P210:
// 24 interleaved) chroma bytes read as 32-bit words, 32 bytes of v210 out through store_unit
template <int SRC>
__global__ void k_planar16_to_v210_x2(const Args a)
{
        UG_XY();
        constexpr bool k420 = SRC == S_420P10 || SRC == S_P010;
        const int nunits = a.w / 12;
        if (x >= nunits || y >= (k420 ? a.h / 2 : a.h)) return;
        uint32_t cb[6], cr[6];
        if (SRC == S_420P10) {
                const uint32_t *scb = (const uint32_t *) (ROW(const uint8_t, 1, y) + 12 * x), *scr = (const uint32_t *) (ROW(const uint8_t, 2, y) + 12 * x);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                        const uint32_t u = scb[i], v = scr[i];
                        cb[2 * i] = u & 0xffffu, cb[2 * i + 1] = u >> 16, cr[2 * i] = v & 0xffffu, cr[2 * i + 1] = v >> 16;
                }
        } else {
                const uint32_t *sc = (const uint32_t *) (ROW(const uint8_t, 1, y) + 24 * x);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                        const uint32_t c = sc[i];
                        cb[i] = (c & 0xffffu) >> 6, cr[i] = c >> 22;
                }
        }
#pragma unroll
        for (int l = 0; l < (k420 ? 2 : 1); l++) {
                const int row = k420 ? 2 * y + l : y;
                const uint32_t *sy = (const uint32_t *) (ROW(const uint8_t, 0, row) + 24 * x);
                uint32_t Y[12], ow[8];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                        const uint32_t v = sy[i];
                        Y[2 * i] = SRC == S_420P10 ? v & 0xffffu : (v & 0xffffu) >> 6;
                        Y[2 * i + 1] = SRC == S_420P10 ? v >> 16 : v >> 22;
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                        ow[4 * k] = v210w(cb[3 * k], Y[6 * k], cr[3 * k]), ow[4 * k + 1] = v210w(Y[6 * k + 1], cb[3 * k + 1], Y[6 * k + 2]);
                        ow[4 * k + 2] = v210w(cr[3 * k + 1], Y[6 * k + 3], cb[3 * k + 2]), ow[4 * k + 3] = v210w(Y[6 * k + 4], cr[3 * k + 2], Y[6 * k + 5]);
                }
                store_unit<8>(BUF(uint8_t, row), x, nunits, ow);
        }
}

__global__ void k_uyvy_to_yuv444p_x8(const Args a) // k_uyvy_to_yuv444p, 8 pixels per lane: 16 bytes in, 8 bytes to each of the three planes
{
        UG_XY();
        if (x >= a.w / 8 || y >= a.h) return;
        const uint4 q = BUF(const uint4, y)[x];
        const uint32_t s[4] = { q.x, q.y, q.z, q.w };
        uint32_t yy[2], cb[2], cr[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
                const uint32_t a0 = s[2 * h], a1 = s[2 * h + 1];
                yy[h] = byte_of(a0, 1) | byte_of(a0, 3) << 8 | byte_of(a1, 1) << 16 | byte_of(a1, 3) << 24;
                cb[h] = byte_of(a0, 0) * 0x0101u | byte_of(a1, 0) * 0x01010000u;
                cr[h] = byte_of(a0, 2) * 0x0101u | byte_of(a1, 2) * 0x01010000u;
        }
        ug::st_stream((uint2 *) (ROW(uint8_t, 0, y) + 8 * x), make_uint2(yy[0], yy[1]));
        ug::st_stream((uint2 *) (ROW(uint8_t, 1, y) + 8 * x), make_uint2(cb[0], cb[1]));
        ug::st_stream((uint2 *) (ROW(uint8_t, 2, y) + 8 * x), make_uint2(cr[0], cr[1]));
}

// ---------------------------------------------------------------------------------------------------------------------------------
enum Nx { NX_W, NX_W2, NX_W2UP, NX_W6, NX_W6UP, NX_W8UP };
enum Ny { NY_H, NY_H2, NY_H2UP };
enum Fwd { F_NONE, F_MEMCPY, F_PIXFMT_R10K_BGR0, F_PIXFMT_RG48_RGBA, F_PIXFMT_RG48_R12L, F_PIXFMT_RGB_BGR0, F_TO_PLANAR, F_I420, F_I422, F_FROM_PLANAR, F_PIXFMT_RGB_UYVY, F_PIXFMT_RGB_RGBA };

struct Conv {
        const char *uv, *av;
        void (*kernel)(const Args);
        Nx nx;
        Ny ny;
        int coeff_depth; // bit depth of the colour coefficients (0 = no colour arithmetic)
        Fwd fwd;
        const char *fwd_name;
        int min_planes;
        void (*fast)(const Args); // optional wide variant: fast_div iterations of `kernel` per lane
        int fast_div;
};

// get_uv_to_av_conversion table, to_lavc_vid_conv.c:1458-1529 (rows for UYVY, v210, RGB, RGBA)
const Conv kToAv[] = {
        { "UYVY", "yuv420p", nullptr, NX_W, NY_H, 0, F_I420, nullptr, 3 },
        { "UYVY", "yuvj420p", nullptr, NX_W, NY_H, 0, F_I420, nullptr, 3 },
        { "UYVY", "yuv422p", nullptr, NX_W, NY_H, 0, F_I422, nullptr, 3 },
        { "UYVY", "yuvj422p", nullptr, NX_W, NY_H, 0, F_I422, nullptr, 3 },
        { "UYVY", "yuv444p", k_uyvy_to_yuv444p, NX_W2UP, NY_H, 0, F_NONE, nullptr, 3 , k_uyvy_to_yuv444p_x8, 4 },
        { "UYVY", "yuvj444p", k_uyvy_to_yuv444p, NX_W2UP, NY_H, 0, F_NONE, nullptr, 3 , k_uyvy_to_yuv444p_x8, 4 },
        { "UYVY", "nv12", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "uyvy_to_nv12", 2 },
        { "UYVY", "vuya", k_uyvy_to_vuya, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "UYVY", "vuyx", k_uyvy_to_vuya, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "yuv420p10le", k_v210_to_planar<V_420P10>, NX_W6, NY_H2UP, 0, F_NONE, nullptr, 3 , k_v210_to_planar_x2<V_420P10>, 2 },
        { "v210", "yuv422p10le", k_v210_to_planar<V_422P10>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 , k_v210_to_planar_x2<V_422P10>, 2 },
        { "v210", "yuv444p10le", k_v210_to_planar<V_444P10>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "v210", "yuv444p16le", k_v210_to_planar<V_444P16>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "v210", "p010le", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "v210_to_p010le", 2 },
        { "v210", "p210le", k_v210_to_planar<V_P210>, NX_W6, NY_H, 0, F_NONE, nullptr, 2 }, // (the two-group form measured 0.68 against 0.72 here: both planes already leave in 12-byte pieces of whole lines)
        { "v210", "xv30le", k_v210_to_xv30, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "y210le", k_v210_to_y210, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "y212le", k_v210_to_y210, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "RGB", "bgr0", nullptr, NX_W, NY_H, 0, F_PIXFMT_RGB_BGR0, nullptr, 1 },
        { "RGB", "gbrp", k_rgb_to_gbrp<3>, NX_W, NY_H, 0, F_NONE, nullptr, 3 , k_rgb_to_gbrp_x8<3>, 8 },
        { "RGB", "yuv444p", k_rgb_to_yuv444p, NX_W, NY_H, 8, F_NONE, nullptr, 3 },
        { "RGBA", "gbrp", k_rgb_to_gbrp<4>, NX_W, NY_H, 0, F_NONE, nullptr, 3 , k_rgb_to_gbrp_x8<4>, 8 },
        { "RGBA", "bgra", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "rgba_to_bgra", 1 },
        { "Y216", "y210le", nullptr, NX_W, NY_H, 0, F_MEMCPY, nullptr, 1 },
        { "Y216", "y212le", nullptr, NX_W, NY_H, 0, F_MEMCPY, nullptr, 1 },
        { "Y216", "p010le", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "y216_to_p010le", 2 },
        { "Y216", "yuv422p10le", k_y216_to_yuv422p<10>, NX_W2UP, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y216", "yuv422p16le", k_y216_to_yuv422p<16>, NX_W2UP, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y216", "yuv444p16le", k_y216_to_yuv444p16le, NX_W2UP, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y416", "xv30le", k_y416_to_xv30, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "yuv444p", k_y416_to_yuv444p, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p10le", k_y416_to_yuv444pXX<10>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p12le", k_y416_to_yuv444pXX<12>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p16le", k_y416_to_yuv444pXX<16>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "R10k", "yuv444p10le", k_rgb_to_yuv444pXX<10, 10>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "R10k", "yuv444p12le", k_rgb_to_yuv444pXX<10, 12>, NX_W, NY_H, 12, F_NONE, nullptr, 3 },
        { "R10k", "yuv444p16le", k_rgb_to_yuv444pXX<10, 16>, NX_W, NY_H, 16, F_NONE, nullptr, 3 },
        { "R10k", "yuv422p10le", k_r10k_to_yuv42Xp10le<1>, NX_W2, NY_H, 10, F_NONE, nullptr, 3 },
        { "R10k", "yuv420p10le", k_r10k_to_yuv42Xp10le<2>, NX_W2, NY_H2UP, 10, F_NONE, nullptr, 3 },
        { "R10k", "gbrp10le", k_r10k_to_gbrpXX<10>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "R10k", "gbrp16le", k_r10k_to_gbrpXX<16>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "R10k", "x2rgb10le", k_r10k_to_x2rgb10le, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
        { "R10k", "bgr0", nullptr, NX_W, NY_H, 0, F_PIXFMT_R10K_BGR0, nullptr, 1 },
        { "R12L", "yuv444p10le", k_r12l_to_yuv<10, false>, NX_W8UP, NY_H, 10, F_NONE, nullptr, 3 },
        { "R12L", "yuv444p12le", k_r12l_to_yuv<12, false>, NX_W8UP, NY_H, 12, F_NONE, nullptr, 3 },
        { "R12L", "yuv444p16le", k_r12l_to_yuv<16, false>, NX_W8UP, NY_H, 16, F_NONE, nullptr, 3 },
        { "R12L", "yuv422p10le", k_r12l_to_yuv<10, true>, NX_W8UP, NY_H, 10, F_NONE, nullptr, 3 },
        { "R12L", "yuv422p12le", k_r12l_to_yuv<12, true>, NX_W8UP, NY_H, 12, F_NONE, nullptr, 3 },
        { "R12L", "yuv422p16le", k_r12l_to_yuv<16, true>, NX_W8UP, NY_H, 16, F_NONE, nullptr, 3 },
        { "R12L", "p210le", k_r12l_to_p210_ayuv64<false>, NX_W8UP, NY_H, 10, F_NONE, nullptr, 2 },
        { "R12L", "ayuv64le", k_r12l_to_p210_ayuv64<true>, NX_W8UP, NY_H, 16, F_NONE, nullptr, 1 },
        { "R12L", "gbrp12le", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "r12l_to_gbrp12le", 3 },
        { "R12L", "gbrp16le", nullptr, NX_W, NY_H, 0, F_TO_PLANAR, "r12l_to_gbrp16le", 3 },
        { "RG48", "yuv444p10le", k_rgb_to_yuv444pXX<16, 10>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "RG48", "yuv444p12le", k_rgb_to_yuv444pXX<16, 12>, NX_W, NY_H, 12, F_NONE, nullptr, 3 },
        { "RG48", "yuv444p16le", k_rgb_to_yuv444pXX<16, 16>, NX_W, NY_H, 16, F_NONE, nullptr, 3 },
        { "RG48", "gbrp12le", k_rg48_to_gbrp12le, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
};

// av_to_uv_conversions table, from_lavc_vid_conv.c:2049-2172 (rows whose output is UYVY, v210, RGB, RGBA or R10k)
#define UG_FP(av, uv, name) { uv, av, nullptr, NX_W, NY_H, 0, F_FROM_PLANAR, name, 3 }
const Conv kFromAv[] = {
        { "v210", "yuv420p10le", k_planar_to_v210<S_420P10>, NX_W6, NY_H2, 0, F_NONE, nullptr, 3 , k_planar16_to_v210_x2<S_420P10>, 2 },
        { "UYVY", "yuv420p10le", k_to_uyvy<U_420P10>, NX_W2, NY_H2, 0, F_NONE, nullptr, 3 },
        { "RGB", "yuv420p10le", k_yuvp10le_to_rgb<420, 24>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        { "RGBA", "yuv420p10le", k_yuvp10le_to_rgb<420, 32>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        { "R10k", "yuv420p10le", k_yuvp10le_to_rgb<420, 30>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        UG_FP("yuv422p10le", "v210", "yuv422p10le_to_v210"),
        UG_FP("yuv422p10le", "UYVY", "yuv422p10le_to_uyvy"),
        { "RGB", "yuv422p10le", k_yuvp10le_to_rgb<422, 24>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        { "RGBA", "yuv422p10le", k_yuvp10le_to_rgb<422, 32>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        { "R10k", "yuv422p10le", k_yuvp10le_to_rgb<422, 30>, NX_W2, NY_H2, 10, F_NONE, nullptr, 3 },
        { "v210", "yuv444p10le", k_yuv444_to_v210<10>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "yuv444p10le", k_to_uyvy<U_444P10>, NX_W2, NY_H, 0, F_NONE, nullptr, 3 },
        { "RGB", "yuv444p10le", k_yuv444p10le_to_rgb<false>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "RGBA", "yuv444p10le", k_yuv444p10le_to_rgb<true>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "v210", "yuv444p12le", k_yuv444_to_v210<12>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "yuv444p12le", k_to_uyvy<U_444P12>, NX_W2, NY_H, 0, F_NONE, nullptr, 3 },
        { "v210", "yuv444p16le", k_yuv444_to_v210<16>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "yuv444p16le", k_to_uyvy<U_444P16>, NX_W2, NY_H, 0, F_NONE, nullptr, 3 },
        { "v210", "p210le", k_planar_to_v210<S_P210>, NX_W6, NY_H, 0, F_NONE, nullptr, 2 , k_planar16_to_v210_x2<S_P210>, 2 },
        { "UYVY", "p210le", k_p210le_to_uyvy, NX_W2, NY_H, 0, F_NONE, nullptr, 2 },
        { "v210", "p010le", k_planar_to_v210<S_P010>, NX_W6, NY_H2, 0, F_NONE, nullptr, 2 , k_planar16_to_v210_x2<S_P010>, 2 },
        { "UYVY", "p010le", k_to_uyvy<U_P010>, NX_W2, NY_H2, 0, F_NONE, nullptr, 2 },
        { "v210", "yuv420p", k_planar_to_v210<S_420P8>, NX_W6, NY_H2, 0, F_NONE, nullptr, 3 , k_planar8_to_v210_x4<S_420P8>, 4 },
        UG_FP("yuv420p", "UYVY", "yuv420p_to_uyvy"),
        { "RGB", "yuv420p", k_yuv8p_to_rgb<420, false>, NX_W2, NY_H2, 8, F_NONE, nullptr, 3 , k_yuv8p_to_rgb_x8<420, false>, 4 },
        { "RGBA", "yuv420p", k_yuv8p_to_rgb<420, true>, NX_W2, NY_H2, 8, F_NONE, nullptr, 3 , k_yuv8p_to_rgb_x8<420, true>, 4 },
        { "v210", "yuv422p", k_planar_to_v210<S_422P8>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 , k_planar8_to_v210_x4<S_422P8>, 4 },
        UG_FP("yuv422p", "UYVY", "yuv422p_to_uyvy"),
        { "RGB", "yuv422p", k_yuv8p_to_rgb<422, false>, NX_W2, NY_H2, 8, F_NONE, nullptr, 3 , k_yuv8p_to_rgb_x8<422, false>, 4 },
        { "RGBA", "yuv422p", k_yuv8p_to_rgb<422, true>, NX_W2, NY_H2, 8, F_NONE, nullptr, 3 , k_yuv8p_to_rgb_x8<422, true>, 4 },
        { "v210", "yuv444p", k_yuv444_to_v210<8>, NX_W6, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "yuv444p", k_to_uyvy<U_444P8>, NX_W2, NY_H, 0, F_NONE, nullptr, 3 },
        { "RGB", "yuv444p", k_yuv444p_to_rgb<false>, NX_W, NY_H, 8, F_NONE, nullptr, 3 },
        { "RGBA", "yuv444p", k_yuv444p_to_rgb<true>, NX_W, NY_H, 8, F_NONE, nullptr, 3 },
        UG_FP("yuv444p", "VUYA", "yuv444p_to_vuya"),
        { "UYVY", "nv12", k_to_uyvy<U_NV12>, NX_W2, NY_H, 0, F_NONE, nullptr, 2 , k_nv12_x8<0>, 4 },
        { "RGB", "nv12", k_nv12_to_rgb<false>, NX_W2, NY_H, 8, F_NONE, nullptr, 2 , k_nv12_x8<1>, 4 },
        { "RGBA", "nv12", k_nv12_to_rgb<true>, NX_W2, NY_H, 8, F_NONE, nullptr, 2 , k_nv12_x8<2>, 4 },
        { "RGB", "gbrap", nullptr, NX_W, NY_H, 0, F_FROM_PLANAR, "gbrap_to_rgb", 4 },
        { "RGBA", "gbrap", nullptr, NX_W, NY_H, 0, F_FROM_PLANAR, "gbrap_to_rgba", 4 },
        { "RGB", "gbrp", k_gbrp_to_rgb<false>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "RGBA", "gbrp", k_gbrp_to_rgb<true>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "rgb24", nullptr, NX_W, NY_H, 0, F_PIXFMT_RGB_UYVY, nullptr, 1 },
        { "RGBA", "rgb24", nullptr, NX_W, NY_H, 0, F_PIXFMT_RGB_RGBA, nullptr, 1 },
        { "RGBA", "rgb48le", nullptr, NX_W, NY_H, 0, F_PIXFMT_RG48_RGBA, nullptr, 1 },
        { "R12L", "rgb48le", nullptr, NX_W, NY_H, 0, F_PIXFMT_RG48_R12L, nullptr, 1 },
        UG_FP("gbrp10le", "R10k", "gbrp10le_to_r10k"), UG_FP("gbrp10le", "RGB", "gbrp10le_to_rgb"), UG_FP("gbrp10le", "RGBA", "gbrp10le_to_rgba"),
        UG_FP("gbrp10le", "RG48", "gbrp10le_to_rg48"), UG_FP("gbrp12le", "R12L", "gbrp12le_to_r12l"), UG_FP("gbrp12le", "R10k", "gbrp12le_to_r10k"),
        UG_FP("gbrp12le", "RGB", "gbrp12le_to_rgb"), UG_FP("gbrp12le", "RGBA", "gbrp12le_to_rgba"), UG_FP("gbrp12le", "RG48", "gbrp12le_to_rg48"),
        UG_FP("gbrp16le", "R12L", "gbrp16le_to_r12l"), UG_FP("gbrp16le", "R10k", "gbrp16le_to_r10k"), UG_FP("gbrp16le", "RG48", "gbrp16le_to_rg48"),
        { "R10k", "yuv444p10le", k_yuv444pXX_to_r10k<10>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "R12L", "yuv444p10le", k_yuv444pXX_to_r12l<10>, NX_W8UP, NY_H, 10, F_NONE, nullptr, 3 },
        { "RG48", "yuv444p10le", k_yuv444pXX_to_rg48<10>, NX_W, NY_H, 10, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p10le", k_yuv444pXX_to_y416<10>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "R10k", "yuv444p12le", k_yuv444pXX_to_r10k<12>, NX_W, NY_H, 12, F_NONE, nullptr, 3 },
        { "R12L", "yuv444p12le", k_yuv444pXX_to_r12l<12>, NX_W8UP, NY_H, 12, F_NONE, nullptr, 3 },
        { "RG48", "yuv444p12le", k_yuv444pXX_to_rg48<12>, NX_W, NY_H, 12, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p12le", k_yuv444pXX_to_y416<12>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "R10k", "yuv444p16le", k_yuv444pXX_to_r10k<16>, NX_W, NY_H, 16, F_NONE, nullptr, 3 },
        { "R12L", "yuv444p16le", k_yuv444pXX_to_r12l<16>, NX_W8UP, NY_H, 16, F_NONE, nullptr, 3 },
        { "RG48", "yuv444p16le", k_yuv444pXX_to_rg48<16>, NX_W, NY_H, 16, F_NONE, nullptr, 3 },
        { "Y416", "yuv444p16le", k_yuv444pXX_to_y416<16>, NX_W, NY_H, 0, F_NONE, nullptr, 3 },
        { "UYVY", "xv30le", k_xv30_to_uyvy, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "xv30le", k_packed_to_v210<true>, NX_W6, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "xv30le", k_xv30_to_y416, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
        { "UYVY", "y210le", k_y210_to_uyvy, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "y210le", k_packed_to_v210<false>, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "y210le", k_y210_to_y416, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "UYVY", "y212le", k_y210_to_uyvy, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "y212le", k_packed_to_v210<false>, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "y212le", k_y210_to_y416, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "v210", "ayuv64le", k_ayuv64_to_v210, NX_W6UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "ayuv64le", k_ayuv64_to_y416, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
        { "UYVY", "vuya", k_vuya_to_uyvy, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "UYVY", "vuyx", k_vuya_to_uyvy, NX_W2UP, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "vuya", k_vuyax_to_y416<true>, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
        { "Y416", "vuyx", k_vuyax_to_y416<false>, NX_W, NY_H, 0, F_NONE, nullptr, 1 },
};

// yuvj* frames take the rows of their yuv* twins (from_lavc_vid_conv.c:2108-2119, to_lavc_vid_conv.c:1496-1505)
const char *canonical_av(const char *av, char (&tmp)[32])
{
        if (av && !strncmp(av, "yuvj", 4) && strlen(av) < sizeof tmp - 1) {
                snprintf(tmp, sizeof tmp, "yuv%s", av + 4);
                return tmp;
        }
        return av;
}

template <size_t N>
const Conv *find(const Conv (&tab)[N], const char *uv, const char *av)
{
        char tmp[32];
        if (!uv || !av) return nullptr;
        for (const Conv &c : tab) {
                if (!strcmp(c.uv, uv) && !strcmp(c.av, av)) return &c;
        }
        const char *canon = canonical_av(av, tmp);
        for (const Conv &c : tab) {
                if (!strcmp(c.uv, uv) && !strcmp(c.av, canon)) return &c;
        }
        return nullptr;
}

int launch(const Conv &c, const Args &a, hipStream_t st)
{
        const int w = a.w, h = a.h;
        const int nx = c.nx == NX_W ? w : c.nx == NX_W2 ? w / 2 : c.nx == NX_W2UP ? (w + 1) / 2 : c.nx == NX_W6 ? w / 6 : c.nx == NX_W6UP ? (w + 5) / 6 : (w + 7) / 8;
        const int ny = c.ny == NY_H ? h : c.ny == NY_H2 ? h / 2 : (h + 1) / 2;
        if (nx <= 0 || ny <= 0) return UG_HIP_SUCCESS;
        const dim3 block(64, 4, 1);
        if (c.fast && nx % c.fast_div == 0) {
                uintptr_t bits = (uintptr_t) a.buf | (uintptr_t) a.pitch;
                for (int i = 0; i < c.min_planes; i++) bits |= (uintptr_t) a.d[i] | (uintptr_t) a.ls[i];
                if (!strcmp(c.av, "gbrp") && !strcmp(c.uv, "RGB") && (a.w % 16)) bits |= 1; // 3 * width source lines must stay 16-aligned
                if (c.fast == k_uyvy_to_yuv444p_x8 && (a.w % 8)) bits |= 1;                 // (w + 1) / 2 pairs: an odd width can pass the divisibility test
                if ((bits & 15) == 0) {
                        const int fx = nx / c.fast_div;
                        hipLaunchKernelGGL(c.fast, dim3((unsigned) ((fx + 63) / 64), (unsigned) ((ny + 3) / 4), 1), block, 0, st, a);
                        UG_HIP_LAUNCH_CHECK();
                        return UG_HIP_SUCCESS;
                }
        }
        const dim3 grid((unsigned) ((nx + 63) / 64), (unsigned) ((ny + 3) / 4), 1);
        hipLaunchKernelGGL(c.kernel, grid, block, 0, st, a);
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

// bytes per sample of a frame format's planes as the kernels address them: 16-bit planes must sit at even addresses, the packed
// 32-bit formats at multiples of four (the reference asserts the same, e.g. to_lavc_vid_conv.c:199-202,396-397)
int frame_align(const char *av)
{
        if (strstr(av, "xv30") || strstr(av, "vuy") || strstr(av, "x2rgb")) return 4;
        if (strstr(av, "10le") || strstr(av, "12le") || strstr(av, "16le") || strstr(av, "y21") || strstr(av, "ayuv64")) return 2;
        return 1;
}

// vc_get_linesize (video_codec.c:507-521) for the codecs of the to_lavc table
long uv_linesize(const char *uv, int w)
{
        if (!strcmp(uv, "UYVY")) return ug::linesize(UG_PF_UYVY, w);
        if (!strcmp(uv, "v210")) return ug::linesize(UG_PF_V210, w);
        if (!strcmp(uv, "RGB")) return 3L * w;
        if (!strcmp(uv, "RGBA")) return 4L * w;
        if (!strcmp(uv, "Y216")) return (w + 1) / 2 * 8L;
        if (!strcmp(uv, "Y416")) return 8L * w;
        if (!strcmp(uv, "R10k")) return (w + 63) / 64 * 64 * 4L;
        if (!strcmp(uv, "R12L")) return (w + 7) / 8 * 36L;
        if (!strcmp(uv, "RG48")) return 6L * w;
        return 0;
}

// how the kernels address a packed UltraGrid buffer: 32-bit words (UYVY, v210, RGBA, R10k, VUYA), 16-bit samples (RG48, Y216, Y416), bytes
int uv_align(const char *uv)
{
        if (!strcmp(uv, "RGB") || !strcmp(uv, "R12L")) return 1;
        if (!strcmp(uv, "RG48") || !strcmp(uv, "Y216") || !strcmp(uv, "Y416")) return 2;
        return 4;
}

bool fill_frame(Args &a, const ug_av_frame *f, int planes, int align)
{
        for (int i = 0; i < planes; i++) {
                if (!f->data[i] || f->linesize[i] <= 0) return false;
                if (((uintptr_t) f->data[i] | (uintptr_t) f->linesize[i]) & (uintptr_t) (align - 1)) return false;
                a.d[i] = (uint8_t *) f->data[i];
                a.ls[i] = f->linesize[i];
        }
        a.w = f->width;
        a.h = f->height;
        if (!ug::dims_ok(a.w, a.h)) return false; // (the C ABI's size bound, ug_common.h)
        for (int i = 0; i < planes; i++) {
                if (!ug::span_ok(f->linesize[i], a.h)) return false;
        }
        return true;
}

} // namespace

extern "C" {

int ug_hip_color_coeffs(int cs, int depth, int out[14])
{
        const int slot = depth_slot(depth);
        if (cs == 0) cs = 2; // CS_DFL: BT.709 unless UltraGrid is started with --param color-601 (color_space.c:152-157)
        if ((cs != 1 && cs != 2) || slot < 0 || !out) return UG_HIP_EINVAL;
        memcpy(out, kCoeffs[cs - 1][slot], sizeof kCoeffs[0][0]);
        return UG_HIP_SUCCESS;
}

// compute_color_coeffs(kr, kb, depth) (color_space.c:193-197 over the COEFFS macro, :60-131): Q14 coefficients for arbitrary luma weights;
// depth 0 = full range.  Doubles, as there; the 13 narrow fields of struct color_coeffs are `short` (color_space.h:135-148).
int ug_hip_compute_color_coeffs(double kr, double kb, int depth, int out[14])
{
        if (!out || depth < 0 || depth > 16 || (depth > 0 && depth < 8)) return UG_HIP_EINVAL;
        const double kg = 1. - kr - kb, D = 2. * (kr + kg), E = 2. * (1. - kr);
        const double ylim = depth == 0 ? 1.0 : 219. * (1 << (depth - 8)) / ((1 << depth) - 1);
        const double clim = depth == 0 ? 1.0 : 224. * (1 << (depth - 8)) / ((1 << depth) - 1);
        const double base = 1 << kBase, eps = 0.5;
        auto scaled = [&](double x) { return (int) ((x * base) + (x > 0 ? 1. : -1.) * eps); };
        const int v[14] = {
                (int) (((kr * ylim) * base) + eps), (int) (((kg * ylim) * base) + eps), (int) (((kb * ylim) * base) + eps),
                (int) (((-kr / D * clim) * base) - eps), (int) (((-kg / D * clim) * base) - eps), (int) ((((1 - kb) / D * clim) * base) + eps),
                (int) ((((1 - kr) / E * clim) * base) - eps), (int) (((-kg / E * clim) * base) - eps), (int) (((-kb / E * clim) * base) + eps),
                scaled(1. / ylim), scaled((2. * (1. - kr)) / clim), scaled((-kb * (2. * (kr + kg)) / kg) / clim), scaled((-kr * (2. * (1. - kr)) / kg) / clim),
                scaled((2. * (kr + kg)) / clim),
        };
        for (int i = 0; i < 13; i++) out[i] = (short) v[i];
        out[13] = v[13];
        return UG_HIP_SUCCESS;
}

int ug_hip_uv_to_av_supported(const char *uv_codec, const char *av_pixfmt) { return find(kToAv, uv_codec, av_pixfmt) != nullptr; }
int ug_hip_av_to_uv_supported(const char *av_pixfmt, const char *uv_codec) { return find(kFromAv, uv_codec, av_pixfmt) != nullptr; }

int ug_hip_uv_to_av(const char *uv_codec, const char *av_pixfmt, const void *in_data, const struct ug_av_frame *out, ug_hip_stream_t stream)
{
        const Conv *c = find(kToAv, uv_codec, av_pixfmt);
        if (!c) {
                ug::set_last_error_msg("ug_hip_uv_to_av: no such conversion");
                return UG_HIP_EUNSUPP;
        }
        Args a = {};
        if (!in_data || !out || !fill_frame(a, out, c->min_planes, frame_align(c->av)) || ((uintptr_t) in_data & 3)) {
                ug::set_last_error_msg("ug_hip_uv_to_av: null pointer, bad geometry or misaligned plane");
                return UG_HIP_EINVAL;
        }
        const int w = a.w, h = a.h;
        if (!ug::span_ok(uv_linesize(c->uv, w), h)) return ug::refuse_size("ug_hip_uv_to_av");
        switch (c->fwd) {
        case F_I420:
                return ug_hip_uyvy_to_i420(in_data, 0, out->data[0], out->linesize[0], out->data[1], out->linesize[1], out->data[2], out->linesize[2], w, h, stream);
        case F_I422:
                return ug_hip_uyvy_to_i422(in_data, 0, out->data[0], out->linesize[0], out->data[1], out->linesize[1], out->data[2], out->linesize[2], w, h, stream);
        case F_TO_PLANAR: {
                ug_to_planar_data d = {};
                d.width = w, d.height = h, d.in_data = in_data;
                for (int i = 0; i < c->min_planes; i++) d.out_data[i] = out->data[i], d.out_linesize[i] = (unsigned) out->linesize[i];
                return ug_hip_to_planar(c->fwd_name, &d, stream);
        }
        case F_MEMCPY: { // to_lavc_memcpy_data :1847-1860: vc_get_size(width) bytes of every line
                const size_t ls = (size_t) uv_linesize(c->uv, w);
                UG_HIP_TRY(hipMemcpy2DAsync(out->data[0], (size_t) out->linesize[0], in_data, ls, ls, (size_t) h, hipMemcpyDeviceToDevice, (hipStream_t) stream));
                return UG_HIP_SUCCESS;
        }
        case F_PIXFMT_R10K_BGR0: // r10k_to_bgr0 :1302-1312: vc_copyliner10k(dst, src, linesize(RGBA), 16, 8, 0)
                return ug_hip_pixfmt_convert(UG_PF_R10K, UG_PF_RGBA, in_data, out->data[0], w, h, 0, out->linesize[0], 16, 8, 0, stream);
        case F_PIXFMT_RGB_BGR0: // rgb_to_bgr0, to_lavc_vid_conv.c:1291-1300: vc_copylineRGBtoRGBA(dst, src, linesize(RGBA), 16, 8, 0)
                return ug_hip_pixfmt_convert(UG_PF_RGB, UG_PF_RGBA, in_data, out->data[0], w, h, 0, out->linesize[0], 16, 8, 0, stream);
        default: break;
        }
        a.buf = (uint8_t *) in_data;
        a.pitch = uv_linesize(c->uv, w);
        if (c->coeff_depth) memcpy(a.c, kCoeffs[1][depth_slot(c->coeff_depth)], sizeof a.c); // get_color_coeffs(CS_DFL, depth): BT.709
        return launch(*c, a, (hipStream_t) stream);
}

int ug_hip_av_to_uv(const char *av_pixfmt, const char *uv_codec, void *dst, int pitch, const struct ug_av_frame *in, const int rgb_shift[3],
                    ug_hip_stream_t stream)
{
        const Conv *c = find(kFromAv, uv_codec, av_pixfmt);
        if (!c) {
                ug::set_last_error_msg("ug_hip_av_to_uv: no such conversion");
                return UG_HIP_EUNSUPP;
        }
        Args a = {};
        if (!dst || !in || pitch <= 0 || !fill_frame(a, in, c->min_planes, frame_align(c->av)) || !ug::span_ok(pitch, a.h) || (((uintptr_t) dst | (uintptr_t) pitch) & (uintptr_t) (uv_align(c->uv) - 1))) {
                ug::set_last_error_msg("ug_hip_av_to_uv: null pointer, bad geometry or misaligned buffer");
                return UG_HIP_EINVAL;
        }
        static const int kDefaultShift[3] = { 0, 8, 16 };
        const int *sh = rgb_shift ? rgb_shift : kDefaultShift;
        const int w = a.w, h = a.h;
        switch (c->fwd) {
        case F_FROM_PLANAR: {
                ug_from_planar_data d = {};
                d.width = w, d.height = h, d.out_data = dst, d.out_pitch = (unsigned) pitch;
                for (int i = 0; i < c->min_planes; i++) d.in_data[i] = in->data[i], d.in_linesize[i] = (unsigned) in->linesize[i];
                d.rgb_shift[0] = sh[0], d.rgb_shift[1] = sh[1], d.rgb_shift[2] = sh[2];
                return ug_hip_from_planar(c->fwd_name, &d, stream);
        }
        case F_PIXFMT_RGB_UYVY: // rgb24_to_uyvy :171-184: vc_copylineRGBtoUYVY per line, dst_len = vc_get_linesize(width, UYVY)
                return ug_hip_pixfmt_convert(UG_PF_RGB, UG_PF_UYVY, in->data[0], dst, w, h, in->linesize[0], pitch, 0, 8, 16, stream);
        case F_PIXFMT_RG48_RGBA: // rgb48le_to_rgba :519-534
                return ug_hip_pixfmt_convert(UG_PF_RG48, UG_PF_RGBA, in->data[0], dst, w, h, in->linesize[0], pitch, sh[0], sh[1], sh[2], stream);
        case F_PIXFMT_RG48_R12L: // rgb48le_to_r12l :536-552
                return ug_hip_pixfmt_convert(UG_PF_RG48, UG_PF_R12L, in->data[0], dst, w, h, in->linesize[0], pitch, sh[0], sh[1], sh[2], stream);
        case F_PIXFMT_RGB_RGBA: // rgb24_to_rgb32 :205-219
                return ug_hip_pixfmt_convert(UG_PF_RGB, UG_PF_RGBA, in->data[0], dst, w, h, in->linesize[0], pitch, sh[0], sh[1], sh[2], stream);
        default: break;
        }
        if ((unsigned) sh[0] > 24 || (unsigned) sh[1] > 24 || (unsigned) sh[2] > 24) {
                ug::set_last_error_msg("ug_hip_av_to_uv: rgb_shift out of range");
                return UG_HIP_EINVAL;
        }
        a.buf = (uint8_t *) dst;
        a.pitch = pitch;
        a.rs = sh[0], a.gs = sh[1], a.bs = sh[2];
        a.am = 0xFFFFFFFFu ^ (0xFFu << a.rs) ^ (0xFFu << a.gs) ^ (0xFFu << a.bs);
        if (c->coeff_depth) {
                // get_cs_for_conv, from_lavc_vid_conv.c:2614-2658: BT.601 for BT470BG / SMPTE170M / SMPTE240M frames, else BT.709 (the default);
                // full-range (AVCOL_RANGE_JPEG) frames take the depth-0 table
                const bool src_601 = in->colorspace == 5 || in->colorspace == 6 || in->colorspace == 7;
                const bool full = in->color_range == 2;
                memcpy(a.c, kCoeffs[src_601 ? 0 : 1][full ? 0 : depth_slot(c->coeff_depth)], sizeof a.c);
        }
        return launch(*c, a, (hipStream_t) stream);
}

} // extern "C"
