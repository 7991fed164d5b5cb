"""The reference's own unit tests of the pixel-format path (test/codec_conversions_test.cpp, compiled unmodified from the reference tree by
`make -C oracle ref`), BASELINE.json configs[0]'s harness: on the reference's CPU converters (plumbing, no GPU), and with the two
to_planar.h functions they call -- uyvy_to_i420 via testcard_convert_buffer, y216_to_p010le -- replaced, under the reference's own names and
signature, by the MI355X implementation behind the C ABI (oracle/ref_tests/to_planar_gpu_shim.c).
(The JPEG and lavc unit tests of the reference run in tests/test_module_harness.py and tests/test_lavc_hook.py.)"""
import os
import subprocess

import pytest

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref")
NAMES = ["codec_conversion_test_testcard_uyvy_to_i420", "codec_conversion_test_y216_to_p010le"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ug_ref_codec_test_cpu")), reason="oracle/_ref/ug_ref_codec_test_cpu not built")
def test_codec_conversions_test_on_the_reference_cpu_path():
    r = subprocess.run([os.path.join(REF, "ug_ref_codec_test_cpu")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    for n in NAMES:
        assert f"{n}: PASSED" in r.stdout
    assert "conversions run on the GPU: 0" in r.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ug_ref_codec_test")), reason="oracle/_ref/ug_ref_codec_test not built")
@pytest.mark.gpu
def test_codec_conversions_test_on_the_gpu_implementation():
    r = subprocess.run([os.path.join(REF, "ug_ref_codec_test")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for n in NAMES:
        assert f"{n}: PASSED" in r.stdout
    assert "conversions run on the GPU: 17" in r.stdout      # 5 sizes of uyvy_to_i420 + 12 of y216_to_p010le, odd widths and heights among them


def test_misc_test_color_coeff_range_on_the_product_tables():
    """test/misc_test.c:46-87 (misc_test_color_coeff_range) restated on ug_hip_color_coeffs(CS_DFL, depth): the scaled coefficients map the
    extreme R,G,B inputs to within 1 LSB (at 8 bits) of the nominal limited range -- Y 16..235, Cb / Cr 16..240 (color_space.h:84-105)."""
    import ctypes as C

    from ultragrid_amd import lib
    names = "y_r y_g y_b cb_r cb_g cb_b cr_r cr_g cr_b y_scale r_cr g_cb g_cr b_cb".split()
    base = 14                                                     # COMP_BASE for a 32-bit comp_type_t (color_space.h:70-71)
    for d in (8, 10, 12, 16):
        got = (C.c_int * 14)()
        assert lib.load().ug_hip_color_coeffs(0, d, got) == 0     # CS_DFL
        c = dict(zip(names, got))
        d_max, max_diff = (1 << d) - 1, 1 << (d - 8)
        lo, hi_y, hi_c, mid = 1 << (d - 4), 235 << (d - 8), 240 << (d - 8), 1 << (d - 1)
        y = lambda r, g, b: (r * c["y_r"] + g * c["y_g"] + b * c["y_b"]) >> base
        cb = lambda r, g, b: (r * c["cb_r"] + g * c["cb_g"] + b * c["cb_b"]) >> base
        cr = lambda r, g, b: (r * c["cr_r"] + g * c["cr_g"] + b * c["cr_b"]) >> base
        assert abs(y(0, 0, 0) + lo - lo) <= max_diff and abs(y(d_max, d_max, d_max) + lo - hi_y) <= max_diff
        assert abs(cb(d_max, d_max, 0) + mid - lo) <= max_diff and abs(cb(0, 0, d_max) + mid - hi_c) <= max_diff
        assert abs(cr(0, d_max, d_max) + mid - lo) <= max_diff and abs(cr(d_max, 0, 0) + mid - hi_c) <= max_diff
