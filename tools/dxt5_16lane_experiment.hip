// dxt5_16lane_experiment.hip -- the alternative mapping SURVEY.md H2 asked to be evaluated for the headline kernel: ONE PIXEL PER
// LANE, 16 lanes (one DPP row) per 4x4 block, min / max / index words reduced across the row with DPP, instead of the product's one
// block per lane (ultragrid_amd/csrc/dxt_encode.hip).  UYVY -> DXT5-YCoCg, same arithmetic, same bits (checked against the product
// library on the same frames before timing).  Not part of the product; kept so that the number in DESIGN.md 4.1 can be reproduced:
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/dxt5_16lane_experiment tools/dxt5_16lane_experiment.hip \
//           -Lultragrid_amd -lug_mi355x -Wl,-rpath,'$ORIGIN/../ultragrid_amd'   &&   tools/dxt5_16lane_experiment
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/ug_mi355x.h"

namespace {
constexpr float kInv255 = 0.00392156862745f;
constexpr float kOffset = (float) (128.0 / 255.0);
constexpr float kInsetC = (float) ((8.0 / 255.0) / 16.0);
constexpr float kInsetY = (float) ((16.0 / 255.0) / 32.0);
__device__ __forceinline__ float clamp01(float v) { return fminf(1.0f, fmaxf(0.0f, v)); }
__device__ __forceinline__ uint32_t round_u32(float x) { return (uint32_t) rintf(x); } // UG_DXT_TIES_EVEN, the library default

// DPP row rotations: every lane of a 16-lane row sees lane (i - n) mod 16
template <int N>
__device__ __forceinline__ float row_ror(float v)
{
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
        return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x120 + N, 0xf, 0xf, false);
}
__device__ __forceinline__ float row_min(float v)
{
        v = fminf(v, row_ror<1>(v)); v = fminf(v, row_ror<2>(v)); v = fminf(v, row_ror<4>(v)); v = fminf(v, row_ror<8>(v));
        return v;
}
__device__ __forceinline__ float row_max(float v)
{
        v = fmaxf(v, row_ror<1>(v)); v = fmaxf(v, row_ror<2>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<8>(v));
        return v;
}
__device__ __forceinline__ uint32_t row_or(uint32_t v)
{
        v |= row_ror<1>(v); v |= row_ror<2>(v); v |= row_ror<4>(v); v |= row_ror<8>(v);
        return v;
}

// grid: x = ceil(blocks_per_row / 4) (a wave = 4 consecutive blocks of a block row), y = block rows / 4 (4 waves per workgroup), z = frame
__global__ __launch_bounds__(256) void dxt5_pixel_per_lane(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int bpr, int brows, uint32_t pitch,
                                                           size_t sfs, size_t dfs)
{
        const int lane = threadIdx.x & 63, p = lane & 15, row4 = p >> 2, col = p & 3;
        const int bx = blockIdx.x * 4 + (lane >> 4), by = blockIdx.y * 4 + threadIdx.y;
        if (by >= brows || bx >= bpr) return;
        src += (size_t) blockIdx.z * sfs;
        dst += (size_t) blockIdx.z * dfs;
        // the UYVY word of this pixel's pair
        const uint32_t w = *(const uint32_t *) (src + (size_t) (4 * by + row4) * pitch + (size_t) bx * 8 + (col >> 1) * 4);
        const float u = (float) (w & 0xff) * kInv255, v = (float) ((w >> 16) & 0xff) * kInv255;
        const float yy = (float) ((col & 1 ? w >> 24 : w >> 8) & 0xff) * kInv255;
        const float U = u - 0.5f, V = v - 0.5f;
        const float Yl = 1.1643f * (yy - 0.0625f);
        const float r = Yl + 1.7926f * V, g = (Yl - 0.2132f * U) - 0.5328f * V, b = Yl + 2.1124f * U;
        const float t = __builtin_fmaf(g, 2.0f, r);
        const float Y = (t + b) * 0.25f;
        const float Co = __builtin_fmaf(r - b, 0.5f, kOffset);
        const float Cg = __builtin_fmaf(__builtin_fmaf(g, 2.0f, -r) - b, 0.25f, kOffset);
        float mnY = row_min(Y), mxY = row_max(Y), mnCo = row_min(Co), mxCo = row_max(Co), mnCg = row_min(Cg), mxCg = row_max(Cg);
        // SelectYCoCgDiagonal: the SEQUENTIAL sum i = 0..15 (fp32 addition is not associative: a tree reduction would change bits).
        // The running sum travels one lane per step: after step i lane i holds ((..(0 + p0) + p1..) + pi).
        {
                const float midx = (mxCo + mnCo) * 0.5f, midy = (mxCg + mnCg) * 0.5f;
                const float prod = (Co - midx) * (Cg - midy);
                float acc = 0.0f + prod; // lane 0: 0 + p0
#pragma unroll
                for (int i = 1; i < 16; i++) {
                        const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x111, 0xf, 0xf, false)); // row_shr:1
                        if (p == i) acc = prev + prod;
                }
                const float cov = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane | 15), __builtin_bit_cast(int, acc)));
                if (cov < 0.0f) { const float s = mxCg; mxCg = mnCg; mnCg = s; }
        }
        uint32_t scale = 1;
        float fs = 1.0f, rfs = 1.0f;
        {
                const float m0 = fmaxf(fabsf(mnCo - kOffset), fabsf(mnCg - kOffset));
                const float m1 = fmaxf(fabsf(mxCo - kOffset), fabsf(mxCg - kOffset));
                const float m = fmaxf(m0, m1);
                if (m < (float) (64.0 / 255.0)) { scale = 2; fs = 2.0f; rfs = 0.5f; }
                if (m < (float) (32.0 / 255.0)) { scale = 4; fs = 4.0f; rfs = 0.25f; }
        }
        uint32_t w_end;
        float cmx[2], cmn[2];
        {
                const float q[2] = { 31.0f, 63.0f };
                const float mx_in[2] = { mxCo, mxCg }, mn_in[2] = { mnCo, mnCg };
                uint32_t imax[2], imin[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                        float a = (mx_in[k] - kOffset) * fs + kOffset;
                        float bb = (mn_in[k] - kOffset) * fs + kOffset;
                        const float inset = (a - bb) * 0.0625f - kInsetC;
                        bb = clamp01(bb + inset);
                        a = clamp01(a - inset);
                        imax[k] = round_u32(a * q[k]);
                        imin[k] = round_u32(bb * q[k]);
                }
                w_end = ((imax[0] << 11) | (imax[1] << 5) | (scale - 1)) | (((imin[0] << 11) | (imin[1] << 5) | (scale - 1)) << 16);
                imax[0] = (imax[0] << 3) | (imax[0] >> 2); imax[1] = (imax[1] << 2) | (imax[1] >> 4);
                imin[0] = (imin[0] << 3) | (imin[0] >> 2); imin[1] = (imin[1] << 2) | (imin[1] >> 4);
                const float inv255 = (float) (1.0 / 255.0);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                        cmx[k] = ((float) imax[k] * inv255 - kOffset) * rfs + kOffset;
                        cmn[k] = ((float) imin[k] * inv255 - kOffset) * rfs + kOffset;
                }
        }
        {
                const float inset = (mxY - mnY) * 0.03125f - kInsetY;
                mnY = clamp01(mnY + inset);
                mxY = clamp01(mxY - inset);
        }
        uint32_t w0 = (round_u32(mnY * 255.0f) << 8) | round_u32(mxY * 255.0f);
        // alpha index of THIS pixel: the reference's 7-compare count (one pixel per lane: nothing to amortise a search over)
        uint32_t aidx;
        {
                const float inv7 = (float) (1.0 / 7.0);
                const float mid = (mxY - mnY) / 14.0f;
                uint32_t c = Y <= mnY + mid ? 1u : 0u;
#pragma unroll
                for (int k = 2; k <= 7; k++) c += Y <= ((float) (8 - k) * mxY + (float) (k - 1) * mnY) * inv7 + mid ? 1u : 0u;
                aidx = (c + 1) & 7;
                aidx ^= aidx < 2 ? 1u : 0u;
        }
        // 48-bit field F: pixel p at bits [3p, 3p + 3): word0[31:16] = F[15:0], word1 = F[47:16]
        const unsigned long long field = (unsigned long long) aidx << (3 * p);
        const uint32_t flo = row_or((uint32_t) field), fhi = row_or((uint32_t) (field >> 32));
        w0 |= flo << 16;
        const uint32_t w1 = (flo >> 16) | (fhi << 16);
        uint32_t cidx;
        {
                const float q1 = (float) (1.0 / 3.0), q2 = (float) (2.0 / 3.0);
                const float wa = 1.0f - q1, wb = 1.0f - q2;
                const float cx[4] = { cmx[0], cmn[0], cmx[0] * wa + cmn[0] * q1, cmx[0] * wb + cmn[0] * q2 };
                const float cy[4] = { cmx[1], cmn[1], cmx[1] * wa + cmn[1] * q1, cmx[1] * wb + cmn[1] * q2 };
                float d[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        const float tx = Co - cx[k], ty = Cg - cy[k];
                        d[k] = tx * tx + ty * ty;
                }
                const uint32_t b0 = d[0] > d[3], b1 = d[1] > d[2], b2 = d[0] > d[2], b3 = d[1] > d[3], b4 = d[2] > d[3];
                cidx = (b0 & b4) | (((b1 & b2) | (b0 & b3)) << 1);
        }
        const uint32_t w_cidx = row_or(cidx << (2 * p));
        if (p == 0) *(uint4 *) (dst + ((size_t) by * bpr + bx) * 16) = make_uint4(w0, w1, w_end, w_cidx);
}
} // namespace

int main()
{
        const int w = 3840, h = 2160, frames = 16, iters = 200;
        const size_t in_frame = (size_t) w * h * 2, out_frame = (size_t) w * h;
        std::vector<uint8_t> host(in_frame * frames);
        srand(7);
        // legal-range smooth-ish content: a low-pass random field (typical video statistics matter for nothing here but the tie cases)
        for (size_t i = 0; i < host.size(); i++) host[i] = (uint8_t) (64 + (rand() % 128));
        uint8_t *src = nullptr, *a = nullptr, *b = nullptr;
        if (hipMalloc((void **) &src, host.size()) != hipSuccess || hipMalloc((void **) &a, out_frame * frames) != hipSuccess ||
            hipMalloc((void **) &b, out_frame * frames) != hipSuccess) {
                fprintf(stderr, "hipMalloc failed\n");
                return 1;
        }
        hipMemcpy(src, host.data(), host.size(), hipMemcpyHostToDevice);
        const int bpr = w / 4, brows = h / 4;
        const dim3 block(64, 4), grid((bpr + 3) / 4, (brows + 3) / 4, frames);
        auto product = [&]() { return ug_hip_dxt_encode_batch(UG_PF_UYVY, UG_DXT5_YCOCG, src, a, w, h, 0, frames, in_frame, out_frame, nullptr); };
        auto experiment = [&]() { hipLaunchKernelGGL(dxt5_pixel_per_lane, grid, block, 0, 0, src, b, bpr, brows, (uint32_t) (2 * w), in_frame, out_frame); };
        if (product() != UG_HIP_SUCCESS) {
                fprintf(stderr, "product encode failed: %s\n", ug_hip_last_error_string());
                return 1;
        }
        experiment();
        hipDeviceSynchronize();
        std::vector<uint8_t> ha(out_frame * frames), hb(out_frame * frames);
        hipMemcpy(ha.data(), a, ha.size(), hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, hb.size(), hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < ha.size(); i += 16) bad += memcmp(&ha[i], &hb[i], 16) != 0;
        printf("blocks that differ between the two mappings: %zu of %zu\n", bad, ha.size() / 16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms[2];
        for (int which = 0; which < 2; which++) {
                for (int i = 0; i < 10; i++) which ? experiment() : (void) product();
                hipEventRecord(e0, 0);
                for (int i = 0; i < iters; i++) which ? experiment() : (void) product();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[which], e0, e1);
                ms[which] /= iters;
        }
        const double px = (double) w * h * frames;
        printf("one block per lane (product):        %.4f ms per launch of %d x 4K, %.1f Gpx/s\n", ms[0], frames, px / ms[0] / 1e6);
        printf("one pixel per lane, 16 lanes/block:  %.4f ms per launch, %.1f Gpx/s  (%.2fx the product's time)\n", ms[1], px / ms[1] / 1e6, ms[1] / ms[0]);
        return bad != 0;
}
