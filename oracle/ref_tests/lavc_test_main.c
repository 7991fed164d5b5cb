/* TEST INFRASTRUCTURE ONLY.  main() for the reference's own unit tests of the lavc pixel-format converters (test/ff_codec_conversions_test.cpp,
 * compiled unmodified from the reference tree): run against oracle/_ref/libugref_lavc_hook.so, i.e. the reference's to_lavc_vid_conv.c /
 * from_lavc_vid_conv.c with their GPU hook enabled and this repository's hook functions behind it, the round trips of those tests go
 * through the MI355X; against libugref_lavc.so they run on the CPU (the control). */
#include <stdbool.h>
#include <stdio.h>
#include <string.h>

extern bool cuda_devices_explicit; /* what `--cuda-device` sets in host.cpp: the switch of the hook (to_lavc_vid_conv.c:1771-1783) */

int ff_codec_conversions_test_yuv444pXXle_from_to_r10k(void);
int ff_codec_conversions_test_yuv444pXXle_from_to_r12l(void);
int ff_codec_conversions_test_yuv444p16le_from_to_rg48(void);
int ff_codec_conversions_test_yuv444p16le_from_to_rg48_out_of_range(void);
int ff_codec_conversions_test_pX10_from_to_v210(void);

int main(int argc, char **argv)
{
        int failed = 0;
        cuda_devices_explicit = argc > 1 && strcmp(argv[1], "hook") == 0;
        printf("hook %s\n", cuda_devices_explicit ? "enabled" : "disabled");
#define RUN(f) { const int rc = f(); printf("%s: %s (%d)\n", #f, rc == 0 ? "PASSED" : "FAILED", rc); failed |= rc != 0; }
        RUN(ff_codec_conversions_test_yuv444pXXle_from_to_r10k)
        RUN(ff_codec_conversions_test_yuv444pXXle_from_to_r12l)
        RUN(ff_codec_conversions_test_yuv444p16le_from_to_rg48)
        RUN(ff_codec_conversions_test_yuv444p16le_from_to_rg48_out_of_range)
        RUN(ff_codec_conversions_test_pX10_from_to_v210)
        return failed;
}
