// probe: what exactly does gfx950's V_ASHR_PK_U8_I32 write?  (LLVM matches min(max(x >> s, 0), 255) pairs to it and then treats bits 31:16
// of the result as zero; the round-3 DXT5 decoder produced wrong bytes exactly where those bits would leak.)
// Build: hipcc --offload-arch=gfx950 -O3 -o ashr_pk_probe ashr_pk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int *in, uint32_t *out)
{
        const int a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
        uint32_t plain = 0xAAAAAAAAu, hi = 0x5555BBBBu, fresh;
        asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "+v"(plain) : "v"(a), "v"(b));
        asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 20 op_sel:[0,0,0,1]" : "+v"(hi) : "v"(a), "v"(b));
        asm volatile("v_mov_b32 %0, 0x12345678\n\tv_ashr_pk_u8_i32 %0, %1, %2, 20" : "=&v"(fresh) : "v"(a), "v"(b));
        out[4 * threadIdx.x] = plain;
        out[4 * threadIdx.x + 1] = hi;
        out[4 * threadIdx.x + 2] = fresh;
        const int ra = a >> 20, rb = b >> 20;
        out[4 * threadIdx.x + 3] = (uint32_t) (ra < 0 ? 0 : ra > 255 ? 255 : ra) | (uint32_t) (rb < 0 ? 0 : rb > 255 ? 255 : rb) << 8;
}
int main()
{
        int h[128];
        const int vals[16] = { 0, 1 << 20, (1 << 20) - 1, 255 << 20, 256 << 20, (255 << 20) + 0xFFFFF, -1, -(1 << 20), 0x7FFFFFFF, (int) 0x80000000, 100 << 20, (37 << 20) + 5, 1 << 30, -(1 << 30), 200 << 20, 17 };
        for (int i = 0; i < 64; i++) { h[2 * i] = vals[i % 16]; h[2 * i + 1] = vals[(i * 7 + 3) % 16]; }
        int *d; uint32_t *o;
        (void) hipMalloc(&d, sizeof h); (void) hipMalloc(&o, 64 * 16); (void) hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
        uint32_t r[256]; (void) hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int bad_lo = 0, zero_hi = 0, keep_hi = 0, opsel_ok = 0;
        for (int i = 0; i < 64; i++) {
                const uint32_t want = r[4 * i + 3];
                bad_lo += (r[4 * i] & 0xFFFF) != want;
                zero_hi += (r[4 * i] >> 16) == 0 && (r[4 * i + 2] >> 16) == 0;
                keep_hi += (r[4 * i] >> 16) == 0xAAAA && (r[4 * i + 2] >> 16) == 0x1234;
                opsel_ok += (r[4 * i + 1] >> 16) == want && (r[4 * i + 1] & 0xFFFF) == 0xBBBB;
                if (i < 16) printf("a=%08x b=%08x  plain(dst was AAAAAAAA)=%08x  op_sel hi(dst was 5555BBBB)=%08x  fresh(dst was 12345678)=%08x  want lo=%04x\n", h[2 * i], h[2 * i + 1], r[4 * i], r[4 * i + 1], r[4 * i + 2], want);
        }
        printf("V_ASHR_PK_U8_I32: low 16 bits wrong in %d of 64; bits 31:16 zero in %d, preserved in %d; op_sel:[0,0,0,1] writes the high half and keeps the low one in %d\n", bad_lo, zero_hi, keep_hi, opsel_ok);
        return 0;
}
