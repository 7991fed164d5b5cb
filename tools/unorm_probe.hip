// probe: what does a typed buffer load with NUM_FORMAT_UNORM return for each byte value?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint8_t *src, float *out)
{
        const uint64_t base = (uint64_t) src;
        i32x4 desc = { (int) (uint32_t) base, (int) ((uint32_t) (base >> 32) & 0xffffu), -1, 0x52FAC };
        uint32_t off = threadIdx.x * 4;
        f32x4 v;
        asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(desc) : "memory");
        out[4 * threadIdx.x] = v.x, out[4 * threadIdx.x + 1] = v.y, out[4 * threadIdx.x + 2] = v.z, out[4 * threadIdx.x + 3] = v.w;
}
int main()
{
        uint8_t h[256]; for (int i = 0; i < 256; i++) h[i] = i;
        uint8_t *d; float *o; hipMalloc(&d, 256); hipMalloc(&o, 1024); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
        float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int ne_mul = 0, ne_div = 0;
        for (int i = 0; i < 256; i++) {
                float m = (float) i * 0.00392156862745f, q = (float) i / 255.0f;
                if (memcmp(&r[i], &m, 4)) ne_mul++;
                if (memcmp(&r[i], &q, 4)) ne_div++;
        }
        printf("UNORM typed load: %d values differ from p*(1/255.f), %d differ from p/255.f; r[3]=%a mul=%a div=%a\n", ne_mul, ne_div, r[3], 3 * 0.00392156862745f, 3 / 255.0f);
        return 0;
}
