// planar_api.hip -- the whole of src/from_planar.h (decode_planar_func_t, :73-113) and src/to_planar.h (decode_buffer_func_t,
// :62-74) on the GPU, called by the reference's own function names over the reference's own argument structs:
//
//   ug_hip_from_planar("gbrp12le_to_r12l", &d, stream)   <->   gbrp12le_to_r12l(d)          (struct from_planar_data, from_planar.h:58-70)
//   ug_hip_to_planar("r12l_to_gbrp16le", &d, stream)     <->   r12l_to_gbrp16le(d)          (struct to_planar_data, to_planar.h:53-59)
//
// These are the building blocks of libavcodec/{to,from}_lavc_vid_conv.c: planar <-> packed shuffles with
// bit-depth shifts.  Pure byte movement, HBM-bound: every input sample is read once and every output byte written once.  One lane
// moves 8 pixels; planes and packed lines are fetched / stored with 4-16 byte accesses when pointers and line sizes allow, else
// sample by sample (ragged last group, odd pitches).  Bit-exact against the compiled reference (tests/test_planar_api.py),
// including what the reference does with samples that carry bits above the nominal depth (it never masks them).
//
// Families (reference lines):
//   planar RGB -> RGB / RGBA / RG48 / R10k / R12L    from_planar.c:60-262,335-369,477-563
//   planar YUV 4:4:4 -> VUYA                         from_planar.c:565-581
//   planar YUV 4:2:2 (8..16 bit) -> UYVY / YUYV      from_planar.c:391-475
//   planar YUV 4:2:0 -> UYVY, planar 4:2:2 10 -> v210  planar.hip (ug_hip_yuv420p_to_uyvy, ug_hip_yuv422p10le_to_v210)
//   planar YUV 4:2:0 -> I420                         from_planar.c:371-389
//   R12L -> planar RGB 12/16                         to_planar.c:380-476
//   Y216 -> P010, RGBA -> BGRA, VUYA -> planar 4:4:4 to_planar.c:157-203,304-341
//   v210 -> P010, UYVY -> NV12, UYVY -> I420         pixfmt.hip / planar.hip entries
#include <string.h>

#include "ug_common.h"

namespace {

enum Out { O_RGB, O_RGBA_SHIFT, O_RGBA_BYTES, O_RG48, O_R10K, O_R12L };

struct RgbpArgs {
        const uint8_t *in[4];
        uint32_t ls[4];
        uint8_t *out;
        uint32_t pitch;
        int width, height;
        int depth;
        int rs, gs, bs;
        uint32_t alpha_mask;
        int in_align;  // guaranteed alignment of (plane row + 8 samples * group): 1 = none
        int out_align; // same for (out row + group bytes)
};

// N samples of T starting at sample x0 of the row; samples at or beyond n_valid read as 0
template <typename T, int N>
__device__ __forceinline__ void load_samples(const uint8_t *row, int x0, int n_valid, int align, uint32_t (&s)[N])
{
        constexpr int kBytes = N * (int) sizeof(T);
        const uint8_t *p = row + (size_t) x0 * sizeof(T);
        if (n_valid >= N && align >= kBytes) {
                uint32_t w[kBytes / 4];
                if (kBytes == 16) {
                        const uint4 v = *(const uint4 *) p;
                        w[0] = v.x; w[1 % (kBytes / 4)] = v.y; w[2 % (kBytes / 4)] = v.z; w[3 % (kBytes / 4)] = v.w;
                } else if (kBytes == 8) {
                        const uint2 v = *(const uint2 *) p;
                        w[0] = v.x; w[1 % (kBytes / 4)] = v.y;
                } else {
                        w[0] = *(const uint32_t *) p;
                }
#pragma unroll
                for (int i = 0; i < N; i++) {
                        s[i] = sizeof(T) == 1 ? (w[i / 4] >> (8 * (i % 4))) & 0xffu : (w[i / 2] >> (16 * (i % 2))) & 0xffffu;
                }
        } else {
#pragma unroll
                for (int i = 0; i < N; i++) s[i] = i < n_valid ? (uint32_t) ((const T *) p)[i] : 0u;
        }
}

// K words = 4K bytes to dst; nbytes < 4K for a ragged group (byte stores)
template <int K>
__device__ __forceinline__ void store_words(uint8_t *dst, const uint32_t (&w)[K], int nbytes, int align)
{
        if (nbytes == 4 * K && align >= 4) {
                if (K % 4 == 0 && align >= 16) {
#pragma unroll
                        for (int i = 0; i < K / 4; i++) ug::st_stream((uint4 *) dst + i, make_uint4(w[4 * i], w[(4 * i + 1) % K], w[(4 * i + 2) % K], w[(4 * i + 3) % K]));
                } else if (K % 2 == 0 && align >= 8) {
#pragma unroll
                        for (int i = 0; i < K / 2; i++) ug::st_stream((uint2 *) dst + i, make_uint2(w[2 * i], w[(2 * i + 1) % K]));
                } else {
#pragma unroll
                        for (int i = 0; i < K; i++) ug::st_stream((uint32_t *) dst + i, w[i]);
                }
        } else {
#pragma unroll
                for (int i = 0; i < 4 * K; i++) {
                        if (i < nbytes) dst[i] = (uint8_t) (w[i / 4] >> (8 * (i % 4)));
                }
        }
}

// The group of lane `lane` of a wave whose lanes hold consecutive groups of one line (first group g0 at `first`): when every group of the
// wave is whole and the line allows the wide accesses, the 64 groups leave as one contiguous region (ug::WaveWords: every store
// instruction covers whole lines -- streamed lane by lane, K-word groups wrote up to 1.56 x their bytes, profiles/r03_write_by_row.txt);
// a wave with a ragged last group, or an unaligned line, stores lane by lane as before.  blockDim = (64, 4): one LDS region per wave.
template <int K>
__device__ __forceinline__ void store_group(uint8_t *first, int lane, int units, const uint32_t (&w)[K], int nbytes, int align)
{
        using WS = ug::WaveWords<K>;
        __shared__ uint32_t lds[WS::LDS_DWORDS ? 4 * WS::LDS_DWORDS : 1];
        const bool whole = nbytes == 4 * K && align >= 4 * WS::G;
        if (__all(whole)) {
                WS::store(first, w, lds + threadIdx.y * WS::LDS_DWORDS, lane, units, units); // the lanes past the line have returned
        } else {
                store_words<K>(first + (size_t) lane * (4 * K), w, nbytes, align);
        }
}

template <int NB>
__device__ __forceinline__ void bytes_to_words(const uint32_t (&b)[NB], uint32_t (&w)[NB / 4])
{
#pragma unroll
        for (int i = 0; i < NB / 4; i++) {
                w[i] = (b[4 * i] & 0xffu) | (b[4 * i + 1] & 0xffu) << 8 | (b[4 * i + 2] & 0xffu) << 16 | (b[4 * i + 3] & 0xffu) << 24;
        }
}

// planar R,G,B(,A) -> packed.  T = sample type of the planes; OUT = packed layout.
template <typename T, int OUT>
__global__ __launch_bounds__(256) void rgbp_to_packed_kernel(RgbpArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        const int x0 = 8 * gx;
        if (x0 >= a.width || y >= a.height) return;
        const int n = min(8, a.width - x0);
        uint32_t r[8], g[8], b[8], al[8];
        load_samples<T, 8>(a.in[0] + (size_t) y * a.ls[0], x0, n, a.in_align, r);
        load_samples<T, 8>(a.in[1] + (size_t) y * a.ls[1], x0, n, a.in_align, g);
        load_samples<T, 8>(a.in[2] + (size_t) y * a.ls[2], x0, n, a.in_align, b);
        if (OUT == O_RGBA_BYTES) load_samples<T, 8>(a.in[3] + (size_t) y * a.ls[3], x0, n, a.in_align, al);
        uint8_t *const row = a.out + (size_t) y * a.pitch;
        const int lane = threadIdx.x, g0 = gx - lane, units = min(64, (a.width + 7) / 8 - g0); // the wave's groups: g0 .. g0 + units - 1
        const int d = a.depth;
        if (OUT == O_RGB) { // gbrpXXle_to_rgb, from_planar.c:477-497; gbrap_to_rgb_rgba :335-354
                uint32_t bytes[24], w[6];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        bytes[3 * i] = r[i] >> (d - 8);
                        bytes[3 * i + 1] = g[i] >> (d - 8);
                        bytes[3 * i + 2] = b[i] >> (d - 8);
                }
                bytes_to_words<24>(bytes, w);
                store_group<6>(row + (size_t) g0 * 24, lane, units, w, 3 * n, a.out_align);
        } else if (OUT == O_RGBA_SHIFT) { // gbrpXXle_to_rgba, from_planar.c:499-529
                uint32_t w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        w[i] = a.alpha_mask | (r[i] >> (d - 8)) << a.rs | (g[i] >> (d - 8)) << a.gs | (b[i] >> (d - 8)) << a.bs;
                }
                store_group<8>(row + (size_t) g0 * 32, lane, units, w, 4 * n, a.out_align);
        } else if (OUT == O_RGBA_BYTES) { // gbrap_to_rgb_rgba with alpha plane, from_planar.c:335-354
                uint32_t w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) w[i] = r[i] | g[i] << 8 | b[i] << 16 | al[i] << 24;
                store_group<8>(row + (size_t) g0 * 32, lane, units, w, 4 * n, a.out_align);
        } else if (OUT == O_RG48) { // rgbpXXle_to_rg48_int, from_planar.c:159-178
                uint32_t w[12];
                uint32_t s[24];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        s[3 * i] = (r[i] << (16 - d)) & 0xffffu;
                        s[3 * i + 1] = (g[i] << (16 - d)) & 0xffffu;
                        s[3 * i + 2] = (b[i] << (16 - d)) & 0xffffu;
                }
#pragma unroll
                for (int i = 0; i < 12; i++) w[i] = s[2 * i] | s[2 * i + 1] << 16;
                store_group<12>(row + (size_t) g0 * 48, lane, units, w, 6 * n, a.out_align);
        } else if (OUT == O_R10K) { // gbrpXXle_to_r10k, from_planar.c:204-230
                uint32_t bytes[32], w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        bytes[4 * i] = r[i] >> (d - 8);
                        bytes[4 * i + 1] = ((r[i] >> (d - 10)) & 0x3u) << 6 | g[i] >> (d - 6);
                        bytes[4 * i + 2] = ((g[i] >> (d - 10)) & 0xfu) << 4 | b[i] >> (d - 4);
                        bytes[4 * i + 3] = ((b[i] >> (d - 10)) & 0x3fu) << 2 | 0x3u;
                }
                bytes_to_words<32>(bytes, w);
                store_group<8>(row + (size_t) g0 * 32, lane, units, w, 4 * n, a.out_align);
        } else { // O_R12L: gbrpXXle_to_r12l, from_planar.c:60-134 -- a little-endian stream of 12-bit r,g,b; the whole 36-byte group
                 // is written even when the line ends inside it (samples past the end read as 0 here, as stack garbage there)
                uint32_t v[24], bytes[36], w[9];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                        v[3 * i] = r[i] >> (d - 12);
                        v[3 * i + 1] = g[i] >> (d - 12);
                        v[3 * i + 2] = b[i] >> (d - 12);
                }
#pragma unroll
                for (int k = 0; k < 12; k++) { // value pair (even, odd) -> 3 bytes
                        const uint32_t e = v[2 * k], o = v[2 * k + 1];
                        bytes[3 * k] = e;
                        bytes[3 * k + 1] = (o & 0xfu) << 4 | e >> 8;
                        bytes[3 * k + 2] = o >> 4;
                }
                bytes_to_words<36>(bytes, w);
                store_group<9>(row + (size_t) g0 * 36, lane, units, w, 36, a.out_align);
        }
}

struct YuvArgs {
        const uint8_t *in[3];
        uint32_t ls[3];
        uint8_t *out;
        uint32_t pitch;
        int width, height, depth;
        int in_align, out_align;
};

// planar 4:2:2 -> UYVY / YUYV, from_planar.c:391-475: width / 2 pairs per line, samples >> (depth - 8)
template <typename T, bool YUYV>
__global__ __launch_bounds__(256) void yuv422p_to_packed_kernel(YuvArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        const int pairs = a.width / 2;
        if (4 * gx >= pairs || y >= a.height) return;
        const int np = min(4, pairs - 4 * gx);
        uint32_t ys[8], cb[4], cr[4], w[4];
        load_samples<T, 8>(a.in[0] + (size_t) y * a.ls[0], 8 * gx, 2 * np, a.in_align, ys);
        load_samples<T, 4>(a.in[1] + (size_t) y * a.ls[1], 4 * gx, np, a.in_align / 2, cb);
        load_samples<T, 4>(a.in[2] + (size_t) y * a.ls[2], 4 * gx, np, a.in_align / 2, cr);
        const int sh = a.depth - 8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
                const uint32_t y0 = (ys[2 * i] >> sh) & 0xffu, y1 = (ys[2 * i + 1] >> sh) & 0xffu, u = (cb[i] >> sh) & 0xffu, v = (cr[i] >> sh) & 0xffu;
                w[i] = YUYV ? y0 | u << 8 | y1 << 16 | v << 24 : u | y0 << 8 | v << 16 | y1 << 24;
        }
        store_words<4>(a.out + (size_t) y * a.pitch + (size_t) gx * 16, w, 4 * np, a.out_align);
}

// yuv444p_to_vuya, from_planar.c:565-581
__global__ __launch_bounds__(256) void yuv444p_to_vuya_kernel(YuvArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        if (8 * gx >= a.width || y >= a.height) return;
        const int n = min(8, a.width - 8 * gx);
        uint32_t ys[8], cb[8], cr[8], w[8];
        load_samples<uint8_t, 8>(a.in[0] + (size_t) y * a.ls[0], 8 * gx, n, a.in_align, ys);
        load_samples<uint8_t, 8>(a.in[1] + (size_t) y * a.ls[1], 8 * gx, n, a.in_align, cb);
        load_samples<uint8_t, 8>(a.in[2] + (size_t) y * a.ls[2], 8 * gx, n, a.in_align, cr);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = cr[i] | cb[i] << 8 | ys[i] << 16 | 0xff000000u;
        const int lane = threadIdx.x, g0 = gx - lane;
        store_group<8>(a.out + (size_t) y * a.pitch + (size_t) g0 * 32, lane, min(64, (a.width + 7) / 8 - g0), w, 4 * n, a.out_align);
}

// ---- packed -> planar -------------------------------------------------------------------------------------------------------------
struct ToArgs {
        const uint8_t *in;
        uint32_t in_ls;
        uint8_t *out[4];
        uint32_t ls[4];
        int width, height, depth;
        int idx[3]; // r,g,b plane numbers
        int in_align, out_align;
};

template <int NW>
__device__ __forceinline__ void load_words(const uint8_t *p, int nbytes, int align, uint32_t (&w)[NW])
{
        if (nbytes == 4 * NW && align >= 4) {
                if (NW % 4 == 0 && align >= 16) {
#pragma unroll
                        for (int i = 0; i < NW / 4; i++) {
                                const uint4 v = ((const uint4 *) p)[i];
                                w[4 * i] = v.x; w[(4 * i + 1) % NW] = v.y; w[(4 * i + 2) % NW] = v.z; w[(4 * i + 3) % NW] = v.w;
                        }
                } else {
#pragma unroll
                        for (int i = 0; i < NW; i++) w[i] = ((const uint32_t *) p)[i];
                }
        } else {
#pragma unroll
                for (int i = 0; i < NW; i++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                                if (4 * i + k < nbytes) v |= (uint32_t) p[4 * i + k] << (8 * k);
                        }
                        w[i] = v;
                }
        }
}

// N samples of T to a plane row; only the first n_valid are written
template <typename T, int N>
__device__ __forceinline__ void store_samples(uint8_t *row, int x0, int n_valid, int align, const uint32_t (&s)[N])
{
        constexpr int kBytes = N * (int) sizeof(T);
        uint8_t *p = row + (size_t) x0 * sizeof(T);
        if (n_valid >= N && align >= kBytes) {
                uint32_t w[kBytes / 4];
#pragma unroll
                for (int i = 0; i < kBytes / 4; i++) {
                        if (sizeof(T) == 1) w[i] = (s[4 * i] & 0xffu) | (s[(4 * i + 1) % N] & 0xffu) << 8 | (s[(4 * i + 2) % N] & 0xffu) << 16 | (s[(4 * i + 3) % N] & 0xffu) << 24;
                        else w[i] = (s[(2 * i) % N] & 0xffffu) | (s[(2 * i + 1) % N] & 0xffffu) << 16;
                }
                store_words<kBytes / 4>(p, w, kBytes, kBytes >= 16 ? 16 : (kBytes >= 8 ? 8 : 4));
        } else {
#pragma unroll
                for (int i = 0; i < N; i++) {
                        if (i < n_valid) ((T *) p)[i] = (T) s[i];
                }
        }
}

// r12l_to_gbrpXXle, to_planar.c:380-461: 36 bytes = 8 px of 12-bit r,g,b, little-endian bit stream -> three 16-bit planes, << (depth - 12).
// The reference decodes whole groups (it writes up to 7 samples past `width`); here only the samples inside the picture are written.
__global__ __launch_bounds__(256) void r12l_to_planar_kernel(ToArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        if (8 * gx >= a.width || y >= a.height) return;
        const int n = min(8, a.width - 8 * gx);
        uint32_t w[9];
        load_words<9>(a.in + (size_t) y * a.in_ls + (size_t) gx * 36, 36, a.in_align, w);
        uint32_t c[3][8];
#pragma unroll
        for (int k = 0; k < 24; k++) { // value k occupies stream bits 12k .. 12k+11
                const int bit = 12 * k, wi = bit / 32, sh = bit % 32;
                uint32_t v = w[wi] >> sh;
                if (sh > 20) v |= w[(wi + 1) % 9] << (32 - sh);
                c[k % 3][k / 3] = ((v & 0xfffu) << (a.depth - 12)) & 0xffffu;
        }
#pragma unroll
        for (int comp = 0; comp < 3; comp++) {
                const int pl = a.idx[comp];
                store_samples<uint16_t, 8>(a.out[pl] + (size_t) y * a.ls[pl], 8 * gx, n, a.out_align, c[comp]);
        }
}

// rgba_to_bgra, to_planar.c:304-319
__global__ __launch_bounds__(256) void rgba_to_bgra_kernel(ToArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        if (8 * gx >= a.width || y >= a.height) return;
        const int n = min(8, a.width - 8 * gx);
        uint32_t w[8];
        load_words<8>(a.in + (size_t) y * a.in_ls + (size_t) gx * 32, 4 * n, a.in_align, w);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (w[i] & 0xff00ff00u) | (w[i] & 0xffu) << 16 | ((w[i] >> 16) & 0xffu);
        const int lane = threadIdx.x, g0 = gx - lane;
        store_group<8>(a.out[0] + (size_t) y * a.ls[0] + (size_t) g0 * 32, lane, min(64, (a.width + 7) / 8 - g0), w, 4 * n, a.out_align);
}

// vuya_to_i444, to_planar.c:321-337
__global__ __launch_bounds__(256) void vuya_to_i444_kernel(ToArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int y = blockIdx.y * 4 + threadIdx.y;
        if (8 * gx >= a.width || y >= a.height) return;
        const int n = min(8, a.width - 8 * gx);
        uint32_t w[8], ys[8], u[8], v[8];
        load_words<8>(a.in + (size_t) y * a.in_ls + (size_t) gx * 32, 4 * n, a.in_align, w);
#pragma unroll
        for (int i = 0; i < 8; i++) {
                v[i] = w[i] & 0xffu;
                u[i] = (w[i] >> 8) & 0xffu;
                ys[i] = (w[i] >> 16) & 0xffu;
        }
        store_samples<uint8_t, 8>(a.out[0] + (size_t) y * a.ls[0], 8 * gx, n, a.out_align, ys);
        store_samples<uint8_t, 8>(a.out[1] + (size_t) y * a.ls[1], 8 * gx, n, a.out_align, u);
        store_samples<uint8_t, 8>(a.out[2] + (size_t) y * a.ls[2], 8 * gx, n, a.out_align, v);
}

// y216_to_p010le, to_planar.c:157-203: Y216 = Y0 Cb Y1 Cr in 16-bit words.  Luma of both lines of a pair, chroma of the even line only.
// grid.y walks line pairs; a lane moves 4 sample pairs (8 px) of both lines.
__global__ __launch_bounds__(256) void y216_to_p010_kernel(ToArgs a)
{
        const int gx = blockIdx.x * 64 + threadIdx.x;
        const int yp = blockIdx.y * 4 + threadIdx.y;
        const int pairs = (a.width + 1) / 2;
        if (4 * gx >= pairs || 2 * yp >= a.height) return;
        const int np = min(4, pairs - 4 * gx);
        const int ny = min(8, a.width - 8 * gx); // luma samples inside the picture
#pragma unroll
        for (int l = 0; l < 2; l++) {
                const int y = 2 * yp + l;
                if (y >= a.height) break;
                uint32_t w[8], ys[8], c[8];
                load_words<8>(a.in + (size_t) y * a.in_ls + (size_t) gx * 32, 8 * np, a.in_align, w);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                        ys[2 * i] = w[2 * i] & 0xffffu;
                        c[2 * i] = w[2 * i] >> 16;
                        ys[2 * i + 1] = w[2 * i + 1] & 0xffffu;
                        c[2 * i + 1] = w[2 * i + 1] >> 16;
                }
                store_samples<uint16_t, 8>(a.out[0] + (size_t) y * a.ls[0], 8 * gx, ny, a.out_align, ys);
                if (l == 0) store_samples<uint16_t, 8>(a.out[1] + (size_t) yp * a.ls[1], 8 * gx, 2 * np, a.out_align, c);
        }
}

int gcd_align(uintptr_t v, int cap)
{
        int a = cap;
        while (a > 1 && (v % (uintptr_t) a)) a /= 2;
        return a;
}

// largest power of two <= cap dividing every address a lane can form: base pointers, line sizes and the per-group step
int common_align(int cap, int step, const void *const *ptrs, const unsigned *ls, int n)
{
        int a = gcd_align((uintptr_t) step, cap);
        for (int i = 0; i < n; i++) {
                a = gcd_align((uintptr_t) ptrs[i], a);
                a = gcd_align((uintptr_t) ls[i], a);
        }
        return a;
}

dim3 grid_for(int groups, int rows)
{
        return dim3((unsigned) ((groups + 63) / 64), (unsigned) ((rows + 3) / 4), 1);
}

struct FromConv {
        const char *name;
        enum { RGBP, VUYA, UYVY420, I420, P422, V210 } family;
        int out;   // Out for RGBP; 1 = YUYV for P422
        int depth; // 0 = take from_planar_data.in_depth
        int r, g, b, a;
        bool ls0_for_all; // gbrap_to_rgb_rgba indexes every plane with in_linesize[0] (from_planar.c:345)
};

const FromConv kFrom[] = {
        { "gbrap_to_rgb", FromConv::RGBP, O_RGB, 8, 2, 0, 1, -1, true },
        { "gbrap_to_rgba", FromConv::RGBP, O_RGBA_BYTES, 8, 2, 0, 1, 3, true },
        { "gbrp10le_to_rgb", FromConv::RGBP, O_RGB, 10, 2, 0, 1, -1, false },
        { "gbrp10le_to_rgba", FromConv::RGBP, O_RGBA_SHIFT, 10, 2, 0, 1, -1, false },
        { "gbrp10le_to_rg48", FromConv::RGBP, O_RG48, 10, 2, 0, 1, -1, false },
        { "gbrp10le_to_r10k", FromConv::RGBP, O_R10K, 10, 2, 0, 1, -1, false },
        { "gbrp12le_to_rgb", FromConv::RGBP, O_RGB, 12, 2, 0, 1, -1, false },
        { "gbrp12le_to_rgba", FromConv::RGBP, O_RGBA_SHIFT, 12, 2, 0, 1, -1, false },
        { "gbrp12le_to_rg48", FromConv::RGBP, O_RG48, 12, 2, 0, 1, -1, false },
        { "gbrp12le_to_r10k", FromConv::RGBP, O_R10K, 12, 2, 0, 1, -1, false },
        { "gbrp12le_to_r12l", FromConv::RGBP, O_R12L, 12, 2, 0, 1, -1, false },
        { "gbrp16le_to_rgb", FromConv::RGBP, O_RGB, 16, 2, 0, 1, -1, false },
        { "gbrp16le_to_rgba", FromConv::RGBP, O_RGBA_SHIFT, 16, 2, 0, 1, -1, false },
        { "gbrp16le_to_rg48", FromConv::RGBP, O_RG48, 16, 2, 0, 1, -1, false },
        { "gbrp16le_to_r10k", FromConv::RGBP, O_R10K, 16, 2, 0, 1, -1, false },
        { "gbrp16le_to_r12l", FromConv::RGBP, O_R12L, 16, 2, 0, 1, -1, false },
        { "rgbpXX_to_rgb", FromConv::RGBP, O_RGB, 0, 0, 1, 2, -1, false }, // depth 8: gbrap_to_rgb_rgba(d, 0, 1, 2, -1)
        { "rgbpXXle_to_rg48", FromConv::RGBP, O_RG48, 0, 0, 1, 2, -1, false },
        { "rgbpXXle_to_r10k", FromConv::RGBP, O_R10K, 0, 0, 1, 2, -1, false },
        { "rgbpXXle_to_r12l", FromConv::RGBP, O_R12L, 0, 0, 1, 2, -1, false },
        { "yuv444p_to_vuya", FromConv::VUYA, 0, 8, 0, 0, 0, 0, false },
        { "yuv420p_to_uyvy", FromConv::UYVY420, 0, 8, 0, 0, 0, 0, false },
        { "yuv420_to_i420", FromConv::I420, 0, 8, 0, 0, 0, 0, false },
        { "yuv422p_to_uyvy", FromConv::P422, 0, 8, 0, 0, 0, 0, false },
        { "yuv422p_to_yuyv", FromConv::P422, 1, 8, 0, 0, 0, 0, false },
        { "yuv422pXX_to_uyvy", FromConv::P422, 0, 0, 0, 0, 0, 0, false },
        { "yuv422p10le_to_uyvy", FromConv::P422, 0, 10, 0, 0, 0, 0, false },
        { "yuv422p10le_to_v210", FromConv::V210, 0, 10, 0, 0, 0, 0, false },
};

template <typename T>
int launch_rgbp(int out, const RgbpArgs &a, hipStream_t st)
{
        const dim3 grid = grid_for((a.width + 7) / 8, a.height), block(64, 4, 1);
        switch (out) {
        case O_RGB: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_RGB>), grid, block, 0, st, a); break;
        case O_RGBA_SHIFT: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_RGBA_SHIFT>), grid, block, 0, st, a); break;
        case O_RGBA_BYTES: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_RGBA_BYTES>), grid, block, 0, st, a); break;
        case O_RG48: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_RG48>), grid, block, 0, st, a); break;
        case O_R10K: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_R10K>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((rgbp_to_packed_kernel<T, O_R12L>), grid, block, 0, st, a); break;
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

int from_rgbp(const FromConv &c, const ug_from_planar_data *d, hipStream_t st)
{
        const int depth = c.depth ? c.depth : d->in_depth;
        const int min_depth = c.out == O_R12L ? 12 : (c.out == O_R10K ? 10 : 8);
        if (depth < min_depth || depth > 16) {
                ug::set_last_error_msg("ug_hip_from_planar: in_depth out of range for this conversion");
                return UG_HIP_EINVAL;
        }
        const bool bytes8 = depth == 8;
        if (bytes8 && c.out != O_RGB && c.out != O_RGBA_BYTES) {
                ug::set_last_error_msg("ug_hip_from_planar: 8-bit planes only convert to RGB / RGBA");
                return UG_HIP_EINVAL;
        }
        RgbpArgs a = {};
        const int idx[4] = { c.r, c.g, c.b, c.a };
        const int nplanes = c.a >= 0 ? 4 : 3;
        // depth-8 rgbpXX_to_rgb goes through gbrap_to_rgb_rgba, which strides every plane by in_linesize[0]
        const bool ls0 = c.ls0_for_all || (bytes8 && c.depth == 0);
        const void *ptrs[4];
        unsigned ls[4];
        for (int i = 0; i < nplanes; i++) {
                a.in[i] = (const uint8_t *) d->in_data[idx[i]];
                a.ls[i] = ls0 ? d->in_linesize[0] : d->in_linesize[idx[i]];
                if (!a.in[i] || (!bytes8 && (((uintptr_t) a.in[i] | a.ls[i]) & 1))) {
                        ug::set_last_error_msg("ug_hip_from_planar: missing plane, or a 16-bit plane / line size at an odd address");
                        return UG_HIP_EINVAL;
                }
                ptrs[i] = a.in[i];
                ls[i] = a.ls[i];
        }
        a.out = (uint8_t *) d->out_data;
        a.pitch = d->out_pitch;
        a.width = d->width;
        a.height = d->height;
        a.depth = depth;
        if (c.out == O_RGBA_SHIFT) {
                a.rs = d->rgb_shift[0], a.gs = d->rgb_shift[1], a.bs = d->rgb_shift[2];
                if ((unsigned) a.rs > 24 || (unsigned) a.gs > 24 || (unsigned) a.bs > 24) {
                        ug::set_last_error_msg("ug_hip_from_planar: rgb_shift out of range");
                        return UG_HIP_EINVAL;
                }
                a.alpha_mask = 0xFFFFFFFFu ^ (0xFFu << a.rs) ^ (0xFFu << a.gs) ^ (0xFFu << a.bs);
        }
        a.in_align = common_align(16, bytes8 ? 8 : 16, ptrs, ls, nplanes);
        static const int kGroupBytes[] = { 24, 32, 32, 48, 32, 36 };
        const long need = c.out == O_R12L ? (d->width + 7) / 8 * 36L : (long) d->width * kGroupBytes[c.out] / 8;
        if ((long) d->out_pitch < need) {
                ug::set_last_error_msg("ug_hip_from_planar: out_pitch is smaller than a line of the output format");
                return UG_HIP_EINVAL;
        }
        const void *optr[1] = { a.out };
        const unsigned ols[1] = { a.pitch };
        a.out_align = common_align(16, kGroupBytes[c.out], optr, ols, 1);
        return bytes8 ? launch_rgbp<uint8_t>(c.out, a, st) : launch_rgbp<uint16_t>(c.out, a, st);
}

int from_yuv(const FromConv &c, const ug_from_planar_data *d, hipStream_t st)
{
        const int depth = c.depth ? c.depth : d->in_depth;
        if (depth < 8 || depth > 16) {
                ug::set_last_error_msg("ug_hip_from_planar: in_depth out of range");
                return UG_HIP_EINVAL;
        }
        YuvArgs a = {};
        const void *ptrs[3];
        unsigned ls[3];
        for (int i = 0; i < 3; i++) {
                a.in[i] = (const uint8_t *) d->in_data[i];
                a.ls[i] = d->in_linesize[i];
                if (!a.in[i] || (depth > 8 && (((uintptr_t) a.in[i] | a.ls[i]) & 1))) {
                        ug::set_last_error_msg("ug_hip_from_planar: missing plane, or a 16-bit plane / line size at an odd address");
                        return UG_HIP_EINVAL;
                }
                ptrs[i] = a.in[i];
                ls[i] = a.ls[i];
        }
        a.out = (uint8_t *) d->out_data;
        a.pitch = d->out_pitch;
        a.width = d->width;
        a.height = d->height;
        a.depth = depth;
        const void *optr[1] = { a.out };
        const unsigned ols[1] = { a.pitch };
        const dim3 block(64, 4, 1);
        if ((long) a.pitch < (c.family == FromConv::VUYA ? 4L * a.width : 4L * (a.width / 2))) {
                ug::set_last_error_msg("ug_hip_from_planar: out_pitch is smaller than a line of the output format");
                return UG_HIP_EINVAL;
        }
        if (c.family == FromConv::VUYA) {
                a.in_align = common_align(8, 8, ptrs, ls, 3);
                a.out_align = common_align(16, 32, optr, ols, 1);
                hipLaunchKernelGGL(yuv444p_to_vuya_kernel, grid_for((a.width + 7) / 8, a.height), block, 0, st, a);
        } else {
                // luma groups are 8 samples, chroma groups 4: the chroma planes need half the luma alignment (in_align / 2 in the kernel)
                const int sb = depth > 8 ? 2 : 1;
                int al = common_align(8 * sb, 8 * sb, ptrs, ls, 1);
                const int alc = common_align(4 * sb, 4 * sb, ptrs + 1, ls + 1, 2);
                if (2 * alc < al) al = 2 * alc;
                a.in_align = al;
                a.out_align = common_align(16, 16, optr, ols, 1);
                const dim3 grid = grid_for((a.width / 2 + 3) / 4, a.height);
                if (a.width / 2 == 0) return UG_HIP_SUCCESS;
                if (depth > 8) {
                        if (c.out) hipLaunchKernelGGL((yuv422p_to_packed_kernel<uint16_t, true>), grid, block, 0, st, a);
                        else hipLaunchKernelGGL((yuv422p_to_packed_kernel<uint16_t, false>), grid, block, 0, st, a);
                } else {
                        if (c.out) hipLaunchKernelGGL((yuv422p_to_packed_kernel<uint8_t, true>), grid, block, 0, st, a);
                        else hipLaunchKernelGGL((yuv422p_to_packed_kernel<uint8_t, false>), grid, block, 0, st, a);
                }
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

// yuv420_to_i420, from_planar.c:371-389: the three planes copied back to back, out_pitch ignored, both dimensions even
int from_i420(const ug_from_planar_data *d, hipStream_t st)
{
        if ((d->width | d->height) & 1) {
                ug::set_last_error_msg("ug_hip_from_planar: yuv420_to_i420 needs even width and height");
                return UG_HIP_EINVAL;
        }
        const size_t w = (size_t) d->width, h = (size_t) d->height;
        uint8_t *dst_y = (uint8_t *) d->out_data, *dst_u = dst_y + w * h, *dst_v = dst_u + (w / 2) * (h / 2);
        UG_HIP_TRY(hipMemcpy2DAsync(dst_y, w, d->in_data[0], d->in_linesize[0], w, h, hipMemcpyDeviceToDevice, st));
        UG_HIP_TRY(hipMemcpy2DAsync(dst_u, w / 2, d->in_data[1], d->in_linesize[1], w / 2, h / 2, hipMemcpyDeviceToDevice, st));
        UG_HIP_TRY(hipMemcpy2DAsync(dst_v, w / 2, d->in_data[2], d->in_linesize[2], w / 2, h / 2, hipMemcpyDeviceToDevice, st));
        return UG_HIP_SUCCESS;
}

struct ToConv {
        const char *name;
        enum { R12L, Y216, BGRA, I444, P010, NV12, I420 } family;
        int depth, r, g, b;
};

const ToConv kTo[] = {
        { "v210_to_p010le", ToConv::P010, 10, 0, 0, 0 },
        { "y216_to_p010le", ToConv::Y216, 16, 0, 0, 0 },
        { "uyvy_to_nv12", ToConv::NV12, 8, 0, 0, 0 },
        { "rgba_to_bgra", ToConv::BGRA, 8, 0, 0, 0 },
        { "vuya_to_i444", ToConv::I444, 8, 0, 0, 0 },
        { "uyvy_to_i420", ToConv::I420, 8, 0, 0, 0 },
        { "r12l_to_gbrp12le", ToConv::R12L, 12, 2, 0, 1 },
        { "r12l_to_gbrp16le", ToConv::R12L, 16, 2, 0, 1 },
        { "r12l_to_rgbp12le", ToConv::R12L, 12, 0, 1, 2 },
};

} // namespace

extern "C" {

int ug_hip_from_planar_supported(const char *func)
{
        if (!func) return 0;
        for (const FromConv &c : kFrom) {
                if (!strcmp(c.name, func)) return 1;
        }
        return 0;
}

int ug_hip_to_planar_supported(const char *func)
{
        if (!func) return 0;
        for (const ToConv &c : kTo) {
                if (!strcmp(c.name, func)) return 1;
        }
        return 0;
}

int ug_hip_from_planar(const char *func, const struct ug_from_planar_data *d, ug_hip_stream_t stream)
{
        const FromConv *c = nullptr;
        for (const FromConv &k : kFrom) {
                if (func && !strcmp(k.name, func)) c = &k;
        }
        if (!c || !d) {
                ug::set_last_error_msg("ug_hip_from_planar: unknown conversion name");
                return UG_HIP_EINVAL;
        }
        if (!ug::dims_ok(d->width, d->height) ||
            !ug::planes_ok(d->height, { d->out_pitch, d->in_linesize[0], d->in_linesize[1], d->in_linesize[2], d->in_linesize[3] })) {
                return ug::refuse_size("ug_hip_from_planar");
        }
        if (d->width <= 0 || d->height <= 0 || !d->out_data) {
                ug::set_last_error_msg("ug_hip_from_planar: bad geometry or null output");
                return UG_HIP_EINVAL;
        }
        hipStream_t st = (hipStream_t) stream;
        switch (c->family) {
        case FromConv::RGBP: return from_rgbp(*c, d, st);
        case FromConv::VUYA:
        case FromConv::P422: return from_yuv(*c, d, st);
        case FromConv::I420: return from_i420(d, st);
        case FromConv::UYVY420:
                return ug_hip_yuv420p_to_uyvy(d->in_data[0], (int) d->in_linesize[0], d->in_data[1], (int) d->in_linesize[1], d->in_data[2],
                                              (int) d->in_linesize[2], d->out_data, (int) d->out_pitch, d->width, d->height, stream);
        case FromConv::V210:
                return ug_hip_yuv422p10le_to_v210(d->in_data[0], (int) d->in_linesize[0], d->in_data[1], (int) d->in_linesize[1], d->in_data[2],
                                                  (int) d->in_linesize[2], d->out_data, (int) d->out_pitch, d->width, d->height, stream);
        }
        return UG_HIP_EINVAL;
}

int ug_hip_to_planar(const char *func, const struct ug_to_planar_data *d, ug_hip_stream_t stream)
{
        const ToConv *c = nullptr;
        for (const ToConv &k : kTo) {
                if (func && !strcmp(k.name, func)) c = &k;
        }
        if (!c || !d) {
                ug::set_last_error_msg("ug_hip_to_planar: unknown conversion name");
                return UG_HIP_EINVAL;
        }
        // the packed source's own line size (vc_get_linesize of the codec the conversion reads), not the widest one there is: a 16384 x 16384 UYVY
        // picture is 512 MiB, inside the documented limits
        long long src_ls = 0;
        switch (c->family) {
        case ToConv::P010: src_ls = ((long long) d->width + 47) / 48 * 128; break;          // v210
        case ToConv::NV12: case ToConv::I420: src_ls = ((long long) d->width + 1) / 2 * 4; break; // UYVY
        case ToConv::Y216: src_ls = ((long long) d->width + 1) / 2 * 8; break;
        case ToConv::BGRA: case ToConv::I444: src_ls = 4LL * d->width; break;                // RGBA, VUYA
        case ToConv::R12L: src_ls = ((long long) d->width + 7) / 8 * 36; break;
        }
        if (!ug::dims_ok(d->width, d->height) ||
            !ug::planes_ok(d->height, { d->out_linesize[0], d->out_linesize[1], d->out_linesize[2], d->out_linesize[3], src_ls })) {
                return ug::refuse_size("ug_hip_to_planar");
        }
        if (d->width <= 0 || d->height <= 0 || !d->in_data || !d->out_data[0]) {
                ug::set_last_error_msg("ug_hip_to_planar: bad geometry or null pointer");
                return UG_HIP_EINVAL;
        }
        hipStream_t st = (hipStream_t) stream;
        const int w = d->width, h = d->height;
        switch (c->family) { // the source line size is vc_get_linesize(width, codec), as in the reference functions
        case ToConv::P010:
                return ug_hip_v210_to_p010le(d->in_data, 0, d->out_data[0], (int) d->out_linesize[0], d->out_data[1], (int) d->out_linesize[1], w, h, stream);
        case ToConv::NV12:
                // the reference strides its input by 2 * width bytes (to_planar.c:213)
                return ug_hip_uyvy_to_nv12(d->in_data, 2 * w, d->out_data[0], (int) d->out_linesize[0], d->out_data[1], (int) d->out_linesize[1], w, h, stream);
        case ToConv::I420:
                return ug_hip_uyvy_to_i420(d->in_data, 0, d->out_data[0], (int) d->out_linesize[0], d->out_data[1], (int) d->out_linesize[1],
                                           d->out_data[2], (int) d->out_linesize[2], w, h, stream);
        default: break;
        }
        ToArgs a = {};
        a.in = (const uint8_t *) d->in_data;
        a.width = w;
        a.height = h;
        a.depth = c->depth;
        a.idx[0] = c->r, a.idx[1] = c->g, a.idx[2] = c->b;
        const int nout = c->family == ToConv::BGRA ? 1 : (c->family == ToConv::Y216 ? 2 : 3);
        const void *optr[4];
        unsigned ols[4];
        for (int i = 0; i < nout; i++) {
                a.out[i] = (uint8_t *) d->out_data[i];
                a.ls[i] = d->out_linesize[i];
                const bool wide = c->family == ToConv::R12L || c->family == ToConv::Y216;
                if (!a.out[i] || (wide && (((uintptr_t) a.out[i] | a.ls[i]) & 1))) {
                        ug::set_last_error_msg("ug_hip_to_planar: missing plane, or a 16-bit plane / line size at an odd address");
                        return UG_HIP_EINVAL;
                }
                optr[i] = a.out[i];
                ols[i] = a.ls[i];
        }
        const void *iptr[1] = { a.in };
        const dim3 block(64, 4, 1);
        if (c->family == ToConv::R12L) {
                a.in_ls = (unsigned) ((w + 7) / 8 * 36);
                const unsigned ils[1] = { a.in_ls };
                a.in_align = common_align(4, 36, iptr, ils, 1);
                a.out_align = common_align(16, 16, optr, ols, 3);
                hipLaunchKernelGGL(r12l_to_planar_kernel, grid_for((w + 7) / 8, h), block, 0, st, a);
        } else if (c->family == ToConv::BGRA) {
                a.in_ls = (unsigned) (4 * w);
                const unsigned ils[1] = { a.in_ls };
                a.in_align = common_align(16, 32, iptr, ils, 1);
                a.out_align = common_align(16, 32, optr, ols, 1);
                hipLaunchKernelGGL(rgba_to_bgra_kernel, grid_for((w + 7) / 8, h), block, 0, st, a);
        } else if (c->family == ToConv::I444) {
                a.in_ls = (unsigned) (4 * w);
                const unsigned ils[1] = { a.in_ls };
                a.in_align = common_align(16, 32, iptr, ils, 1);
                a.out_align = common_align(8, 8, optr, ols, 3);
                hipLaunchKernelGGL(vuya_to_i444_kernel, grid_for((w + 7) / 8, h), block, 0, st, a);
        } else { // Y216
                a.in_ls = (unsigned) ((w + 1) / 2 * 8);
                const unsigned ils[1] = { a.in_ls };
                a.in_align = common_align(16, 32, iptr, ils, 1);
                a.out_align = common_align(16, 16, optr, ols, 2);
                hipLaunchKernelGGL(y216_to_p010_kernel, grid_for(((w + 1) / 2 + 3) / 4, (h + 1) / 2), block, 0, st, a);
        }
        UG_HIP_LAUNCH_CHECK();
        return UG_HIP_SUCCESS;
}

} // extern "C"
