/* TEST INFRASTRUCTURE ONLY.  main() for the reference's test/codec_conversions_test.cpp (compiled unmodified): BASELINE.json configs[0]'s
 * harness.  Linked with libugref.so alone it runs the reference's CPU converters ("plumbing, no GPU"); linked with to_planar_gpu_shim.c
 * as well, the same two tests run against the MI355X implementation of uyvy_to_i420 / y216_to_p010le. */
#include <stdio.h>

int codec_conversion_test_testcard_uyvy_to_i420(void);
int codec_conversion_test_y216_to_p010le(void);
int to_planar_gpu_shim_calls(void) __attribute__((weak));

int main(void)
{
        int failed = 0;
#define RUN(f) { const int rc = f(); printf("%s: %s (%d)\n", #f, rc == 0 ? "PASSED" : "FAILED", rc); failed |= rc != 0; }
        RUN(codec_conversion_test_testcard_uyvy_to_i420)
        RUN(codec_conversion_test_y216_to_p010le)
        printf("conversions run on the GPU: %d\n", to_planar_gpu_shim_calls ? to_planar_gpu_shim_calls() : 0);
        return failed;
}
