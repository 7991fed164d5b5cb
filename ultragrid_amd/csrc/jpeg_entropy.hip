// jpeg_entropy.hip -- baseline JPEG entropy coding + JFIF container on gfx950, so that the JPEG path produces a
// decodable stream.  In UltraGrid this is the second half of gpujpeg_encoder_encode()
// (src/video_compress/gpujpeg.cpp:624, external libgpujpeg); the object shape below mirrors the call sites
// gpujpeg_encoder_create / _encode / _destroy (gpujpeg.cpp:353,624,639).
//
// Stream: SOI, APP0 (JFIF), DQT x2, SOF0 (8-bit, 3 components, 2x2 / 1x1 / 1x1 = 4:2:0), DHT x4 (T.81 Annex K.3
// tables), DRI, SOS (interleaved), entropy-coded segments of `restart_interval` MCUs separated by RSTm, EOI.
// Restart intervals make the scan data-parallel: every segment starts byte-aligned with DC predictors reset, so
// segments are coded independently (one lane per segment), then compacted by a prefix sum over segment sizes.
// The byte stream is identical to the test writer tests/jpeg_bitstream.py, which Pillow/libjpeg decodes.
#include <string.h>

#include <vector>

#include "ug_common.h"

namespace {

#include "jpeg_huffman_tables.h"

// zig-zag positions of the quantiser table entries in a DQT segment
const uint8_t kZigHost[64] = {
        0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

struct BitWriter {
        uint8_t *p;
        uint64_t acc; // n valid bits, right aligned
        int n;
        __device__ __forceinline__ void put(uint32_t code, int len)
        {
                acc = (acc << len) | code;
                n += len;
                while (n >= 8) {
                        const uint8_t b = (uint8_t) (acc >> (n - 8));
                        *p++ = b;
                        if (b == 0xFF) *p++ = 0; // byte stuffing (T.81 B.1.1.5)
                        n -= 8;
                }
        }
        __device__ __forceinline__ void flush()
        {
                if (n) put((1u << (8 - n)) - 1, 8 - n); // pad with 1-bits
        }
};

// one 8x8 block: DC difference + run/size coded AC (T.81 F.1.2)
__device__ __forceinline__ int encode_block(BitWriter &bw, const int16_t *__restrict__ zz, int pred, int comp)
{
        const uint4 *q = (const uint4 *) zz;
        int16_t c[64];
#pragma unroll
        for (int i = 0; i < 8; i++) {
                const uint4 v = q[i];
                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                        c[8 * i + 2 * k] = (int16_t) (w[k] & 0xffff);
                        c[8 * i + 2 * k + 1] = (int16_t) (w[k] >> 16);
                }
        }
        {
                const int diff = (int) c[0] - pred;
                const int a = diff < 0 ? -diff : diff;
                const int size = a ? 32 - __builtin_clz((unsigned) a) : 0;
                const uint32_t e = kDcTab[comp][size];
                bw.put(e & 0xffff, (int) (e >> 16));
                if (size) bw.put((uint32_t) (diff < 0 ? diff + (1 << size) - 1 : diff), size);
        }
        int run = 0;
        for (int k = 1; k < 64; k++) {
                const int v = c[k];
                if (v == 0) {
                        run++;
                        continue;
                }
                while (run > 15) {
                        const uint32_t z = kAcTab[comp][0xF0];
                        bw.put(z & 0xffff, (int) (z >> 16));
                        run -= 16;
                }
                const int a = v < 0 ? -v : v;
                const int size = 32 - __builtin_clz((unsigned) a);
                const uint32_t e = kAcTab[comp][(run << 4) | size];
                bw.put(e & 0xffff, (int) (e >> 16));
                bw.put((uint32_t) (v < 0 ? v + (1 << size) - 1 : v), size);
                run = 0;
        }
        if (run) {
                const uint32_t e = kAcTab[comp][0x00];
                bw.put(e & 0xffff, (int) (e >> 16));
        }
        return c[0];
}

__global__ __launch_bounds__(64) void entropy_segments_kernel(const int16_t *__restrict__ cy, const int16_t *__restrict__ cb,
                                                              const int16_t *__restrict__ cr, int mcu_w, int n_mcu, int ri,
                                                              int n_seg, uint8_t *__restrict__ scratch, int cap,
                                                              uint32_t *__restrict__ seg_len)
{
        const int seg = blockIdx.x * blockDim.x + threadIdx.x;
        if (seg >= n_seg) return;
        BitWriter bw = { scratch + (size_t) seg * cap, 0, 0 };
        uint8_t *const start = bw.p;
        int py = 0, pcb = 0, pcr = 0;
        const int m_end = min(n_mcu, (seg + 1) * ri);
        for (int m = seg * ri; m < m_end; m++) {
                const int my = m / mcu_w, mx = m - my * mcu_w;
#pragma unroll 1
                for (int b = 0; b < 4; b++) {
                        const long blk = (long) (2 * my + (b >> 1)) * (2 * mcu_w) + 2 * mx + (b & 1);
                        py = encode_block(bw, cy + 64 * blk, py, 0);
                }
                pcb = encode_block(bw, cb + 64L * m, pcb, 1);
                pcr = encode_block(bw, cr + 64L * m, pcr, 1);
        }
        bw.flush();
        seg_len[seg] = (uint32_t) (bw.p - start);
}

// exclusive prefix sum of (segment length + 2-byte marker), single workgroup; off[n_seg] = total stream length
__global__ __launch_bounds__(1024) void segment_offsets_kernel(const uint32_t *__restrict__ seg_len, int n_seg, uint32_t header_len,
                                                               uint32_t *__restrict__ off)
{
        __shared__ uint32_t part[1024];
        const int t = threadIdx.x;
        const int per = (n_seg + 1023) / 1024;
        const int lo = min(n_seg, t * per), hi = min(n_seg, lo + per);
        uint32_t s = 0;
        for (int i = lo; i < hi; i++) s += seg_len[i] + 2;
        part[t] = s;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
                const uint32_t v = t >= d ? part[t - d] : 0;
                __syncthreads();
                part[t] += v;
                __syncthreads();
        }
        uint32_t run = header_len + (t ? part[t - 1] : 0);
        for (int i = lo; i < hi; i++) {
                off[i] = run;
                run += seg_len[i] + 2;
        }
        if (t == 1023) off[n_seg] = header_len + part[1023];
}

// one wave per segment: copy its bytes to the final position and append RSTm (or EOI after the last one)
__global__ __launch_bounds__(256) void compact_kernel(const uint8_t *__restrict__ scratch, int cap, const uint32_t *__restrict__ seg_len,
                                                      const uint32_t *__restrict__ off, int n_seg, uint8_t *__restrict__ out)
{
        const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (seg >= n_seg) return;
        const uint8_t *s = scratch + (size_t) seg * cap;
        uint8_t *d = out + off[seg];
        const uint32_t n = seg_len[seg];
        for (uint32_t i = lane; i < n; i += 64) d[i] = s[i];
        if (lane == 0) {
                d[n] = 0xFF;
                d[n + 1] = seg == n_seg - 1 ? 0xD9 : (uint8_t) (0xD0 + (seg & 7));
        }
}

struct Encoder {
        int width, height, quality, ri, mcu_w, mcu_h, n_mcu, n_seg, cap, device;
        std::vector<uint8_t> header;
        // device workspace
        float *div;
        int16_t *cy, *cb, *cr;
        uint8_t *scratch;
        uint32_t *seg_len, *off;
        uint8_t *header_dev;
        uint32_t *total_host; // pinned
};

void put16(std::vector<uint8_t> &v, int x) { v.push_back((uint8_t) (x >> 8)); v.push_back((uint8_t) x); }

std::vector<uint8_t> build_header(int w, int h, const uint8_t *ql, const uint8_t *qc, int ri)
{
        std::vector<uint8_t> v = { 0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 };
        for (int t = 0; t < 2; t++) {
                v.insert(v.end(), { 0xFF, 0xDB, 0, 67, (uint8_t) t });
                for (int i = 0; i < 64; i++) v.push_back((t ? qc : ql)[kZigHost[i]]);
        }
        v.insert(v.end(), { 0xFF, 0xC0, 0, 17, 8 });
        put16(v, h); put16(v, w);
        v.insert(v.end(), { 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1 });
        const struct { int tc, th; const uint8_t *bits, *vals; int n; } dht[4] = {
                { 0, 0, kDcL_bits, kDcL_vals, (int) sizeof kDcL_vals }, { 1, 0, kAcL_bits, kAcL_vals, (int) sizeof kAcL_vals },
                { 0, 1, kDcC_bits, kDcC_vals, (int) sizeof kDcC_vals }, { 1, 1, kAcC_bits, kAcC_vals, (int) sizeof kAcC_vals } };
        for (const auto &d : dht) {
                v.insert(v.end(), { 0xFF, 0xC4 });
                put16(v, 19 + d.n);
                v.push_back((uint8_t) (d.tc << 4 | d.th));
                v.insert(v.end(), d.bits, d.bits + 16);
                v.insert(v.end(), d.vals, d.vals + d.n);
        }
        if (ri) {
                v.insert(v.end(), { 0xFF, 0xDD, 0, 4 });
                put16(v, ri);
        }
        v.insert(v.end(), { 0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0 });
        return v;
}

void destroy(Encoder *e)
{
        if (!e) return;
        for (void *p : { (void *) e->div, (void *) e->cy, (void *) e->cb, (void *) e->cr, (void *) e->scratch, (void *) e->seg_len,
                         (void *) e->off, (void *) e->header_dev }) {
                if (p) (void) hipFree(p);
        }
        if (e->total_host) (void) hipHostFree(e->total_host);
        delete e;
}

} // namespace

extern "C" {

typedef struct ug_hip_jpeg_encoder ug_hip_jpeg_encoder;

int ug_hip_jpeg_encoder_create(int width, int height, int quality, int restart_interval, ug_hip_jpeg_encoder **out)
{
        if (!out || width <= 0 || height <= 0 || width > 65535 || height > 65535 || restart_interval < 1 || restart_interval > 65535) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_create: bad arguments");
                return UG_HIP_EINVAL;
        }
        Encoder *e = new Encoder();
        e->width = width; e->height = height; e->quality = quality; e->ri = restart_interval;
        e->mcu_w = (width + 15) / 16; e->mcu_h = (height + 15) / 16; e->n_mcu = e->mcu_w * e->mcu_h;
        e->n_seg = (e->n_mcu + e->ri - 1) / e->ri;
        e->cap = e->ri * 6 * 448 + 16; // worst case per block: 64 coefficients x (16 + 11) bits = 216 B, doubled if every byte were stuffed
        uint8_t ql[64], qc[64];
        float div[128];
        ug_hip_jpeg_qtable(quality, 0, ql);
        ug_hip_jpeg_qtable(quality, 1, qc);
        ug_hip_jpeg_divisors(ql, div);
        ug_hip_jpeg_divisors(qc, div + 64);
        e->header = build_header(width, height, ql, qc, e->ri);
        hipError_t err = hipSuccess;
        auto alloc = [&](void **p, size_t n) { if (err == hipSuccess) err = hipMalloc(p, n); };
        alloc((void **) &e->div, sizeof div);
        alloc((void **) &e->cy, (size_t) 4 * e->n_mcu * 128);
        alloc((void **) &e->cb, (size_t) e->n_mcu * 128);
        alloc((void **) &e->cr, (size_t) e->n_mcu * 128);
        alloc((void **) &e->scratch, (size_t) e->n_seg * e->cap);
        alloc((void **) &e->seg_len, (size_t) e->n_seg * 4);
        alloc((void **) &e->off, (size_t) (e->n_seg + 1) * 4);
        alloc((void **) &e->header_dev, e->header.size());
        if (err == hipSuccess) err = hipHostMalloc((void **) &e->total_host, 64, hipHostMallocDefault);
        if (err == hipSuccess) err = hipMemcpy(e->div, div, sizeof div, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy(e->header_dev, e->header.data(), e->header.size(), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
                ug::set_last_error(err, "ug_hip_jpeg_encoder_create");
                destroy(e);
                return UG_HIP_ERUNTIME;
        }
        *out = (ug_hip_jpeg_encoder *) e;
        return UG_HIP_SUCCESS;
}

void ug_hip_jpeg_encoder_destroy(ug_hip_jpeg_encoder *enc) { destroy((Encoder *) enc); }

size_t ug_hip_jpeg_encoder_max_size(const ug_hip_jpeg_encoder *enc)
{
        const Encoder *e = (const Encoder *) enc;
        return e ? e->header.size() + (size_t) e->n_seg * (e->cap + 2) : 0;
}

int ug_hip_jpeg_encoder_encode(ug_hip_jpeg_encoder *enc, ug_pixfmt_t in, const void *src_dev, int src_pitch, void *out_dev,
                               size_t out_capacity, size_t *out_len, ug_hip_stream_t stream)
{
        Encoder *e = (Encoder *) enc;
        if (!e || !src_dev || !out_dev || !out_len || (15 & (uintptr_t) out_dev)) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: bad arguments");
                return UG_HIP_EINVAL;
        }
        if (in != UG_PF_UYVY) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: input must be UYVY (convert with ug_hip_pixfmt_convert)");
                return UG_HIP_EUNSUPP;
        }
        if (out_capacity < ug_hip_jpeg_encoder_max_size(enc)) {
                ug::set_last_error_msg("ug_hip_jpeg_encoder_encode: output buffer smaller than ug_hip_jpeg_encoder_max_size()");
                return UG_HIP_EINVAL;
        }
        hipStream_t st = (hipStream_t) stream;
        int rc = ug_hip_uyvy_to_jpeg420_coeffs(src_dev, src_pitch, e->width, e->height, e->div, e->cy, e->cb, e->cr, stream);
        if (rc != UG_HIP_SUCCESS) return rc;
        hipLaunchKernelGGL(entropy_segments_kernel, dim3((e->n_seg + 63) / 64), dim3(64), 0, st, e->cy, e->cb, e->cr, e->mcu_w, e->n_mcu,
                           e->ri, e->n_seg, e->scratch, e->cap, e->seg_len);
        hipLaunchKernelGGL(segment_offsets_kernel, dim3(1), dim3(1024), 0, st, e->seg_len, e->n_seg, (uint32_t) e->header.size(), e->off);
        UG_HIP_TRY(hipMemcpyAsync(out_dev, e->header_dev, e->header.size(), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(compact_kernel, dim3((e->n_seg + 3) / 4), dim3(256), 0, st, e->scratch, e->cap, e->seg_len, e->off, e->n_seg,
                           (uint8_t *) out_dev);
        UG_HIP_LAUNCH_CHECK();
        UG_HIP_TRY(hipMemcpyAsync(e->total_host, e->off + e->n_seg, 4, hipMemcpyDeviceToHost, st));
        UG_HIP_TRY(hipStreamSynchronize(st));
        *out_len = *e->total_host;
        return UG_HIP_SUCCESS;
}

} // extern "C"
