/**
 * @file ug_ref_unit_test.c
 * Runs the reference's OWN unit test of the JPEG path -- test/gpujpeg_test.cpp, compiled from where it lies under the reference tree,
 * unmodified -- against this repository's modules: compress_init("GPUJPEG:check"), compress_init("GPUJPEG"), a 1920x1080 RGB frame
 * through compress_frame / compress_pop, decompress_init_multi(JPEG -> RGB) with `--param decompress=gpujpeg`, reconfigure,
 * decompress_frame, max |difference| <= 1 over every byte (test/gpujpeg_test.cpp:68-106).
 * The only things supplied here are main() and the command-line parameter store of src/host.cpp, which the harness does not link.
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int gpujpeg_test_simple(void);

char *uv_argv[] = { "ug_ref_unit_test", NULL };

static struct { char key[64], val[192]; } params[16];
static int n_params;

const char *get_commandline_param(const char *key)
{
        for (int i = 0; i < n_params; i++) {
                if (strcmp(params[i].key, key) == 0) return params[i].val;
        }
        return NULL;
}
void set_commandline_param(const char *key, const char *val)
{
        if (n_params == 16) abort();
        snprintf(params[n_params].key, sizeof params[0].key, "%s", key);
        snprintf(params[n_params++].val, sizeof params[0].val, "%s", val);
}
void register_param(const char *param, const char *doc) { (void) param, (void) doc; }
bool tok_in_argv(char **argv, const char *tok) { (void) argv, (void) tok; return false; }

int main(void)
{
        const int rc = gpujpeg_test_simple(); // 0 passed, 1 skipped (no device), -1 failed
        printf("gpujpeg_test_simple: %s (%d)\n", rc == 0 ? "PASSED" : (rc == 1 ? "SKIPPED" : "FAILED"), rc);
        fflush(stdout);
        _Exit(rc == 0 ? 0 : (rc == 1 ? 77 : 1)); // the test leaves its modules to an atexit handler; the verdict is in
}
