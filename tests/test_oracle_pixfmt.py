"""CPU: oracle/pixfmt_oracle.c vs the reference's own compiled C (oracle/_ref, when present)
and vs the committed fixtures generated from it (tests/golden/pixfmt_ref.npz)."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixfmt_ref.npz"))
PAIRS = sorted({tuple(k.split("_")[1:3]) for k in GOLD.files if k.startswith("in_")})


def test_color_coeffs_golden(po):
    # color_space.c:149-184; values also listed in SURVEY.md 8(c) [probe]
    for d in (0, 8, 10, 12, 16):
        assert po.color_coeffs(d) == GOLD[f"coeffs_d{d}"].tolist()
    assert po.color_coeffs(8)[:3] == [2992, 10063, 1016]
    assert po.color_coeffs(8)[9:] == [19077, 29371, -3494, -8733, 34610]
    assert po.color_coeffs(8, bt601=True)[:3] == [4207, 8260, 1604]
    assert po.color_coeffs(8, bt601=True)[10] == 26149


def test_color_coeff_range(po):
    """Restatement of test/misc_test.c:47-87 (misc_test_color_coeff_range): RGB extremes map to within
    1<<(d-8) of the nominal limited-range Y/Cb/Cr limits."""
    for d in (8, 10, 12, 16):
        c = po.color_coeffs(d)
        mx = (1 << d) - 1
        eps = 1 << (d - 8)
        y_r, y_g, y_b, cb_r, cb_g, cb_b, cr_r, cr_g, cr_b = c[:9]
        lo, hi_y, hi_c = 1 << (d - 4), 235 << (d - 8), 240 << (d - 8)
        half = 1 << (d - 1)
        y = lambda r, g, b: ((r * y_r + g * y_g + b * y_b) >> 14) + lo
        cb = lambda r, g, b: ((r * cb_r + g * cb_g + b * cb_b) >> 14) + half
        cr = lambda r, g, b: ((r * cr_r + g * cr_g + b * cr_b) >> 14) + half
        assert abs(y(0, 0, 0) - lo) <= eps and abs(y(mx, mx, mx) - hi_y) <= eps
        assert abs(cb(0, 0, mx) - hi_c) <= eps and abs(cb(mx, mx, 0) - lo) <= eps
        assert abs(cr(mx, 0, 0) - hi_c) <= eps and abs(cr(0, mx, mx) - lo) <= eps


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_restatement_vs_golden(po, pair):
    i, o = pair
    n = 0
    for k in GOLD.files:
        if not k.startswith(f"out_{i}_{o}_"):
            continue
        _, _, _, dims, rs, gs, bs = k.split("_")
        w, h = map(int, dims.split("x"))
        src = GOLD[f"in_{i}_{o}_{dims}"]
        got = po.convert_frame(i, o, src, w, h, (int(rs), int(gs), int(bs)))
        assert np.array_equal(got, GOLD[k]), k
        n += 1
    assert n >= 4


def test_i420_p010_vs_golden(po):
    for k in GOLD.files:
        if k.startswith("i420_in_"):
            dims = k.split("_")[2]
            w, h = map(int, dims.split("x"))
            y, u, v = po.uyvy_to_i420(GOLD[k], w, h)
            assert np.array_equal(y, GOLD[f"i420_y_{dims}"]) and np.array_equal(u, GOLD[f"i420_u_{dims}"]) and np.array_equal(v, GOLD[f"i420_v_{dims}"])
        if k.startswith("p010_in_"):
            dims = k.split("_")[2]
            w, h = map(int, dims.split("x"))
            y, uv = po.v210_to_p010le(GOLD[k], w, h)
            assert np.array_equal(y, GOLD[f"p010_y_{dims}"]) and np.array_equal(uv, GOLD[f"p010_uv_{dims}"])


def test_uyvy_to_i420_reference_test_pattern(po):
    """test/codec_conversions_test.cpp:28-84: constant pattern {'u','y','v','Y'}, sizes incl. odd ones."""
    for w, h in [(1, 2), (2, 1), (16, 1), (16, 16), (127, 255)]:
        src = np.tile(np.frombuffer(b"uyvY", np.uint8), ((w + 1) // 2) * h)
        y, u, v = po.uyvy_to_i420(src, w, h)
        assert (u == ord("u")).all() and (v == ord("v")).all()
        assert (y[:, 0::2] == ord("y")).all() and (y[:, 1::2] == ord("Y")).all()


def test_chroma_rounding_is_half_up(po):
    # (a+b+1)/2, to_planar.c:364-367 -- not pinned by the reference's constant-pattern test
    src = np.array([10, 1, 20, 2, 11, 3, 23, 4], np.uint8)  # two lines of one pair
    y, u, v = po.uyvy_to_i420(src, 2, 2)
    assert u[0, 0] == 11 and v[0, 0] == 22 and y.tolist() == [[1, 2], [3, 4]]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_restatement_vs_compiled_reference(po, pair):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    i, o = pair
    for (w, h) in [(48, 4), (50, 3), (1920, 2), (2, 2), (127, 5), (6, 1)]:
        for sh in [(0, 8, 16), (16, 8, 0), (8, 16, 24)]:
            src = synth.s1_random(i, w, h, salt=w + sh[0])
            got = po.convert_frame(i, o, src, w, h, sh)
            assert np.array_equal(got, po.ref_convert_frame(i, o, src, w, h, sh, scalar=True)), (w, h, sh)
            if (i, o) != ("RGBA", "RGB"):
                assert np.array_equal(got, po.ref_convert_frame(i, o, src, w, h, sh)), (w, h, sh)


def test_reference_rgba_to_rgb_ssse3_tail_bug(po):
    """Documents why RGBA->RGB is pinned to the reference's portable path: the SSSE3 tail of
    vc_copylineRGBAtoRGB (pixfmt_conv.c:832-838) never advances src, so the last 4-7 pixels of each
    line repeat one pixel on x86 builds.  Everything before the tail agrees."""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    w, h = 64, 2
    src = synth.s1_random("RGBA", w, h)
    sse = po.ref_convert_frame("RGBA", "RGB", src, w, h).reshape(h, w, 3)
    ours = po.convert_frame("RGBA", "RGB", src, w, h).reshape(h, w, 3)
    assert np.array_equal(sse[:, : w - 7], ours[:, : w - 7])
    assert not np.array_equal(sse, ours)
    assert (sse[:, -4:] == sse[:, -4:-3]).all()  # replicated pixel


def test_planar_vs_compiled_reference(po):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    for w, h in [(2, 2), (16, 16), (127, 255), (1, 2), (2, 1), (1920, 6)]:
        src = synth.s1_random("UYVY", w, h, salt=3)
        for a, b in zip(po.uyvy_to_i420(src, w, h), po.uyvy_to_i420(src, w, h, use_ref=True)):
            assert np.array_equal(a, b)
    for w, h in [(48, 2), (96, 4), (1920, 4)]:
        src = synth.s1_random("v210", w, h, salt=4)
        for a, b in zip(po.v210_to_p010le(src, w, h), po.v210_to_p010le(src, w, h, use_ref=True)):
            assert np.array_equal(a, b)


P010_RAGGED = [(1280, 720), (2048, 1080), (50, 7), (48, 7), (50, 6), (6, 1), (7, 5), (13, 9), (3, 5), (5, 8), (4, 7), (1, 6), (2, 5), (11, 6)]
P010_PADS = [(0, 0), (1, 0), (0, 2), (3, 3), (5, 5), (8, 6)]


def test_v210_to_p010le_ragged_geometry_vs_compiled_reference(po):
    """to_planar.c:80-94,139-150: odd last line, whole last group written past `width` on interior line pairs, the margin of the
    last one or two lines copied from two lines above.  Whole planes compared INCLUDING the line padding (pad < roundup6(w) - w:
    the reference's line tails overlap the following lines, the order of its writes decides the result)."""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    for w, h in P010_RAGGED:
        src = synth.s1_random("v210", w, h, salt=w + h)
        for yp, up in P010_PADS:
            got = po.v210_to_p010le(src, w, h, False, yp, up, 0x5A5A)
            want = po.v210_to_p010le(src, w, h, True, yp, up, 0x5A5A)
            for k, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), (w, h, yp, up, k)
