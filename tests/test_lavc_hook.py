"""The drop-in proof for SURVEY.md 8(f) N3: the reference's own libavcodec/to_lavc_vid_conv.c and from_lavc_vid_conv.c, compiled with their
GPU hook enabled (-DHAVE_LAVC_CUDA_CONV) and linked against ultragrid_amd/module/lavc_conv_mi355x.cpp instead of the reference's stubs
(oracle/_ref/libugref_lavc_hook.so, `make -C oracle ref_lavc_hook`).  Driving the reference's public API -- to_lavc_vid_conv_init() /
to_lavc_vid_conv(), get_av_to_uv_conversion() / av_to_uv_convert() -- then runs the conversion on the GPU; the result must equal what the
same API gives in the plain CPU build (oracle/_ref/libugref_lavc.so)."""
import ctypes as C
import os

import numpy as np
import pytest

import test_lavc_conv as T

HOOK = os.path.join(T.HERE, "..", "oracle", "_ref", "libugref_lavc_hook.so")


def hook_lib():
    if not os.path.exists(HOOK):
        pytest.skip("oracle/_ref/libugref_lavc_hook.so not built")
    h = C.CDLL(HOOK)
    for name in ("ug_stub_frame_new", "ug_stub_pixfmt_by_name", "ug_stub_plane_rows", "av_frame_free", "get_codec_from_name", "vc_get_linesize",
                 "to_lavc_vid_conv_init", "to_lavc_vid_conv", "to_lavc_vid_conv_destroy", "get_av_to_uv_conversion", "av_to_uv_convert",
                 "av_to_uv_conversion_destroy"):
        src = getattr(T.ref(), name)
        dst = getattr(h, name)
        dst.restype, dst.argtypes = src.restype, src.argtypes
    C.c_bool.in_dll(h, "cuda_devices_explicit").value = True  # what `--cuda-device` sets (host.cpp); enables the hook (to_lavc_vid_conv.c:1771-1783)
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("uv,av", [("UYVY", "yuv444p"), ("UYVY", "yuv422p"), ("v210", "yuv422p10le"), ("v210", "yuv420p10le"), ("RGB", "yuv444p"),
                                   ("R10k", "yuv420p10le"), ("R12L", "yuv444p12le")])
def test_reference_to_lavc_runs_on_the_gpu(hip, capfd, monkeypatch, uv, av):
    monkeypatch.setenv("UG_MI355X_VERBOSE", "1")
    h, r = hook_lib(), T.ref()
    for (w, h_) in [(48, 8), (96, 6)]:
        ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        src = np.random.default_rng(w).integers(0, 256, ls * h_ + 64).astype(np.uint8)
        want = T.ref_uv_to_av(uv, av, src, w, h_)
        st = h.to_lavc_vid_conv_init(h.get_codec_from_name(uv.encode()), w, h_, h.ug_stub_pixfmt_by_name(av.encode()), 1)
        assert st
        fr = h.to_lavc_vid_conv(st, src.ctypes.data)
        assert fr, "the hook returned no frame"
        got = [p.copy() for p in T.plane_arrays(h, fr.contents, h_)]
        stp = C.c_void_p(st)
        h.to_lavc_vid_conv_destroy(C.byref(stp))
        for k, (g, wnt) in enumerate(zip(got, want)):
            assert np.array_equal(g, wnt), (uv, av, w, h_, k)
    assert f"to_lavc {uv} -> {av} on the device" in capfd.readouterr().err


@pytest.mark.gpu
@pytest.mark.parametrize("uv", ["UYVY", "v210", "RGB", "RGBA"])
def test_reference_from_lavc_runs_on_the_gpu(hip, capfd, monkeypatch, uv):
    """from_lavc_cuda_supp_formats lists yuv422p (from_lavc_vid_conv_cuda.h:50-52): those frames take the hook"""
    monkeypatch.setenv("UG_MI355X_VERBOSE", "1")
    h, r = hook_lib(), T.ref()
    av = "yuv422p"
    for (w, h_), cs, rng_ in [((48, 8), 1, 1), ((96, 6), 5, 2)]:
        frp = T.make_frame(r, av, w, h_, 7, cs, rng_)
        pitch = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        want = T.ref_av_to_uv(frp, av, uv, w, h_, pitch, (0, 8, 16))
        conv = h.get_av_to_uv_conversion(h.ug_stub_pixfmt_by_name(av.encode()), h.get_codec_from_name(uv.encode()))
        assert conv
        dst = np.zeros(pitch * h_ + 64, np.uint8)
        h.av_to_uv_convert(conv, dst.ctypes.data, frp, pitch, (C.c_int * 3)(0, 8, 16))
        cp = C.c_void_p(conv)
        h.av_to_uv_conversion_destroy(C.byref(cp))
        r.av_frame_free(C.byref(frp))
        assert np.array_equal(dst[: pitch * h_].reshape(h_, pitch), want), (uv, w, h_, cs, rng_)
    assert f"from_lavc {av} -> {uv} on the device" in capfd.readouterr().err


@pytest.mark.gpu
def test_hook_declines_geometry_the_device_row_cannot_take(hip, capfd, monkeypatch):
    """ADVICE r1 / VERDICT r2 #3: v210 -> p010le runs on the device for every geometry the reference's CPU function converts -- 1280x720
    (1280 % 6 == 2), odd heights -- and equals the plain CPU build.  Where the device row refuses (width % 6 != 0 with fewer than 5 lines:
    the reference reads in front of its planes) the hook's init must DECLINE (NULL), so that the reference sets up its own CPU
    conversion (to_lavc_vid_conv.c:1901-1906) and every frame still arrives, instead of every to_lavc_vid_conv_cuda() call returning
    NULL and all frames being lost."""
    monkeypatch.setenv("UG_MI355X_VERBOSE", "1")
    h, r = hook_lib(), T.ref()
    uv, av = "v210", "p010le"
    if not hip.L.load().ug_hip_uv_to_av_supported(uv.encode(), av.encode()):
        pytest.skip("row not in the table")
    for (w, h_), on_device in (((1280, 720), True), ((1920, 8), True), ((50, 6), True), ((48, 7), True), ((50, 7), True), ((48, 1), True), ((50, 4), False)):
        ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        src = (np.random.default_rng(w).integers(0, 256, ls * h_ + 64).astype(np.uint8).view(np.uint32) & 0x3FFFFFFF).view(np.uint8)
        want = T.ref_uv_to_av(uv, av, src, w, h_)
        st = h.to_lavc_vid_conv_init(h.get_codec_from_name(uv.encode()), w, h_, h.ug_stub_pixfmt_by_name(av.encode()), 1)
        assert st
        fr = h.to_lavc_vid_conv(st, src.ctypes.data)
        assert fr, "no frame: the hook initialised for a geometry it cannot convert"
        got = [p.copy() for p in T.plane_arrays(h, fr.contents, h_)]
        stp = C.c_void_p(st)
        h.to_lavc_vid_conv_destroy(C.byref(stp))
        err = capfd.readouterr().err
        assert (f"to_lavc {uv} -> {av} on the device" in err) == on_device, (w, h_, err)
        if not on_device:
            assert "left to the CPU path" in err
            continue  # the frame arrived from the reference's own CPU code; at this geometry it copies from in front of its planes
        for k, (g, wnt) in enumerate(zip(got, want)):
            assert np.array_equal(g, wnt), (w, h_, k)


@pytest.mark.gpu
def test_from_lavc_hook_with_a_display_pitch_touches_only_the_lines(hip):
    """with pitch > linesize the caller's buffer may end right after the last LINE (pitch * (height - 1) + linesize bytes);
    the hook must not read or write the pitch - linesize bytes behind it."""
    h, r = hook_lib(), T.ref()
    av, uv, w, h_ = "yuv422p", "UYVY", 48, 8
    frp = T.make_frame(r, av, w, h_, 7, 1, 1)
    ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
    pitch = ls + 160
    want = T.ref_av_to_uv(frp, av, uv, w, h_, ls, (0, 8, 16))
    conv = h.get_av_to_uv_conversion(h.ug_stub_pixfmt_by_name(av.encode()), h.get_codec_from_name(uv.encode()))
    assert conv
    exact = pitch * (h_ - 1) + ls
    dst = np.full(exact + 4096, 0xA5, np.uint8)   # guard zone behind the exact-size buffer
    h.av_to_uv_convert(conv, dst.ctypes.data, frp, pitch, (C.c_int * 3)(0, 8, 16))
    cp = C.c_void_p(conv)
    h.av_to_uv_conversion_destroy(C.byref(cp))
    r.av_frame_free(C.byref(frp))
    assert (dst[exact:] == 0xA5).all(), "bytes behind the last line were written"
    lines = np.stack([dst[y * pitch: y * pitch + ls] for y in range(h_)])
    assert np.array_equal(lines, want.reshape(h_, ls))
    gaps = np.stack([dst[y * pitch + ls: (y + 1) * pitch] for y in range(h_ - 1)])
    assert (gaps == 0xA5).all()


REF_LAVC_TEST = os.path.join(T.HERE, "..", "oracle", "_ref", "ug_ref_lavc_test")
REF_LAVC_NAMES = ["yuv444pXXle_from_to_r10k", "yuv444pXXle_from_to_r12l", "yuv444p16le_from_to_rg48", "yuv444p16le_from_to_rg48_out_of_range", "pX10_from_to_v210"]


@pytest.mark.skipif(not os.path.exists(REF_LAVC_TEST + "_cpu"), reason="oracle/_ref/ug_ref_lavc_test_cpu not built")
def test_reference_lavc_unit_tests_on_the_cpu_build():
    """Control: the reference's own test/ff_codec_conversions_test.cpp (compiled unmodified over the FFmpeg stand-in) passes on the reference's
    own CPU converters -- so the stand-in and the way the file is built do not bend it."""
    import subprocess
    r = subprocess.run([REF_LAVC_TEST + "_cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for n in REF_LAVC_NAMES:
        assert f"ff_codec_conversions_test_{n}: PASSED" in r.stdout


@pytest.mark.skipif(not os.path.exists(REF_LAVC_TEST), reason="oracle/_ref/ug_ref_lavc_test not built")
@pytest.mark.gpu
def test_reference_lavc_unit_tests_through_the_gpu_hook():
    """The same five reference tests against the hook build: to_lavc_vid_conv() / av_to_uv_convert() of the reference with this repository's
    hook functions behind them, i.e. the R10k / R12L / RG48 / v210 <-> planar round trips of those tests run on the MI355X."""
    import subprocess
    r = subprocess.run([REF_LAVC_TEST, "hook"], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for n in REF_LAVC_NAMES:
        assert f"ff_codec_conversions_test_{n}: PASSED" in r.stdout
    # the reference's own message every time the hook has taken a conversion (to_lavc_vid_conv.c:1901-1906): all 22 to_lavc_vid_conv_init calls of
    # the five tests.  The way back stays on the CPU there: the reference offers its from_lavc hook AV_PIX_FMT_YUV422P only
    # (from_lavc_vid_conv_cuda.h:58-60), which these tests do not use -- that direction is covered by the tests above.
    assert out.count("[to_lavc_vid_conv] Using CUDA FFmpeg conversions") == 22 and "hook enabled" in out
