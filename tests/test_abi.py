"""CPU: the C-ABI library loads and exports every symbol include/ug_mi355x.h declares (no compute)."""
import ctypes as C
import os
import re

from ultragrid_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "ug_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(ug_hip_[a-z0-9_]+)\s*\(", hdr))


def test_header_and_binding_agree():
    assert _declared() == set(lib.SYMBOLS), _declared() ^ set(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    l = lib.load()  # raises if the .so or a symbol is missing
    for name in _declared():
        assert getattr(l, name) is not None
    assert l.ug_hip_abi_version() == 5


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}


def test_exports_equal_the_header():
    """The converse of the test above (VERDICT r5 "What's weak" #4): the library exports the header's functions and NOTHING else -- no C++
    helper, no anonymous-namespace function that an extern "C" block turned into a plain C name, no data.  UltraGrid dlopens its plugins
    RTLD_GLOBAL into a host linked -rdynamic (src/lib_common.cpp:186-223): any other global name could interpose or be interposed."""
    for so in ("libug_mi355x.so", "libug_mi355x_alphalinear.so"):
        path = os.path.join(ROOT, "ultragrid_amd", so)
        assert os.path.exists(path), f"{path} not built"
        exp = _exported(path)
        assert exp - _declared() == set(), (so, sorted(exp - _declared()))
        assert _declared() - exp == set(), (so, sorted(_declared() - exp))


def test_export_map_is_in_step_with_the_header():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_export_map", os.path.join(ROOT, "ultragrid_amd", "csrc", "gen_export_map.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert set(gen.header_functions()) == _declared()
    committed = open(os.path.join(ROOT, "ultragrid_amd", "csrc", "libug_mi355x.map")).read()
    assert committed == gen.render(), "include/ug_mi355x.h changed: run python3 ultragrid_amd/csrc/gen_export_map.py"


def test_no_torch_or_cxx_types_in_the_abi():
    hdr = open(os.path.join(ROOT, "include", "ug_mi355x.h")).read()
    assert "std::" not in hdr and "torch" not in hdr and "at::" not in hdr
    assert 'extern "C"' in hdr


def test_error_paths_without_a_gpu():
    l = lib.load()
    # argument validation happens before any device call (cuda_dxt.cu:745 semantics: -1)
    assert l.ug_hip_rgb_to_dxt1(16, 16, 18, 4, None) == lib.EINVAL                              # width % 4: the cuda_dxt.h-shaped entry points keep
    assert l.ug_hip_yuv_to_dxt6(16, 16, 16, 6, None) == lib.EINVAL                              # height % 4: that interface's limit (cuda_dxt.cu:745)
    assert l.ug_hip_dxt_encode(lib.PF_UYVY, lib.DXT1, 16, 16, 17, 4, 0, None) == lib.EINVAL      # 4:2:2: even width (any other size is taken, dxt_util.h:59-67)
    assert l.ug_hip_dxt_encode(lib.PF_RGBA, lib.DXT1, 16, 16, 18, 4, 74, None) == lib.EINVAL     # RGBA pitch % 4
    assert l.ug_hip_dxt_decode(lib.DXT1, lib.PF_UYVY, 16, 16, 17, 4, 0, 0, 8, 16, None) == lib.EINVAL  # UYVY output: even width
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, 8, 16, 16, 4, 0, None) == lib.EINVAL        # src alignment
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, None, 16, 16, 4, 0, None) == lib.EINVAL     # NULL
    assert l.ug_hip_dxt_encode(lib.PF_RG48, lib.DXT1, 16, 16, 16, 4, 0, None) == lib.EUNSUPP
    assert l.ug_hip_dxt_encode(lib.PF_V210, lib.DXT1, 16, 16, 16, 4, 40, None) == lib.EINVAL     # v210 pitch % 16
    assert l.ug_hip_pixfmt_convert(lib.PF_RGB, lib.PF_V210, 16, 16, 8, 8, 0, 0, 0, 8, 16, None) == lib.EUNSUPP
    assert b"unsupported" in l.ug_hip_last_error_string()
    assert l.ug_hip_pixfmt_supported(lib.PF_V210, lib.PF_UYVY) == 1 and l.ug_hip_pixfmt_supported(lib.PF_RGB, lib.PF_V210) == 0
    assert l.ug_hip_dxt_size(lib.DXT1, 1920, 1080) == 1036800 and l.ug_hip_dxt_size(lib.DXT5_YCOCG, 3840, -2160) == 8294400
    assert l.ug_hip_dxt_size(lib.DXT5_YCOCG, 1366, 766) == 1368 * 768 and l.ug_hip_dxt_size(lib.DXT1, 5, -5) == 32   # dxt_get_size: whole blocks
    # linesizes == vc_get_linesize (video_codec.c:507-521; SURVEY.md 8 geometry table)
    assert l.ug_hip_linesize(lib.PF_V210, 1920) == 5120 and l.ug_hip_linesize(lib.PF_V210, 7680) == 20480
    assert l.ug_hip_linesize(lib.PF_UYVY, 3841) == 7684 and l.ug_hip_linesize(lib.PF_RGB, 1920) == 5760
    n = C.c_int(0)
    rc = l.ug_hip_device_count(C.byref(n))
    assert rc in (lib.SUCCESS, lib.ERUNTIME)  # no GPU here: error code + message, never a crash


def test_jpeg_tables_match_oracle(po):
    l = lib.load()
    import numpy as np
    for q in (1, 35, 50, 75, 100):
        for comp in (0, 1):
            t = (C.c_uint8 * 64)()
            d = (C.c_float * 64)()
            l.ug_hip_jpeg_qtable(q, comp, t)
            l.ug_hip_jpeg_divisors(t, d)
            assert np.array_equal(np.frombuffer(t, np.uint8), po.jpeg_qtable(q, comp))
            assert np.array_equal(np.frombuffer(d, np.float32).view(np.uint32), po.jpeg_divisors(po.jpeg_qtable(q, comp)).view(np.uint32))


def test_product_does_not_reference_the_oracle():
    """The product path must not import / link anything under oracle/."""
    pkg = os.path.join(ROOT, "ultragrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and '"oracle.h"' not in txt, os.path.join(dirpath, f)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Argument ranges: the header promises "0 or a negative UG_HIP_E* code" for every entry point, so sizes whose byte
# counts leave int / size_t range -- or are simply not a picture -- must be REFUSED, before any device call.  Bound: width, |height| <= 65536
# (UltraGrid's largest mode is 8K) and every frame / plane <= INT_MAX bytes.  Reference contract: cuda_dxt.cu:745-746 (-1 for a bad size).
# ---------------------------------------------------------------------------------------------------------------------------------------
_P = 0x7F0000001000  # a non-NULL, 4 KiB-aligned address that is never dereferenced: every case below must be refused on its arguments alone
_BIG = 2 ** 30
ABSURD = [(_BIG, _BIG), (2 ** 31 - 4, 4), (65540, 65540), (16, 2 ** 31 - 4), (16, -2 ** 31), (0, 16), (-16, 16), (16, 0), (65536, 65536), (49152, 65536)]


class _FromPlanar(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p), ("out_pitch", C.c_uint), ("in_data", C.c_void_p * 4), ("in_linesize", C.c_uint * 4),
                ("in_depth", C.c_int), ("log2_chroma_h", C.c_int), ("rgb_shift", C.c_int * 3)]


class _ToPlanar(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p * 4), ("out_linesize", C.c_uint * 4), ("in_data", C.c_void_p)]


class _AvFrame(C.Structure):
    _fields_ = [("data", C.c_void_p * 4), ("linesize", C.c_int * 4), ("width", C.c_int), ("height", C.c_int), ("colorspace", C.c_int), ("color_range", C.c_int)]


def _pitch(w, bpp):
    return max(0, min(w, 2 ** 31 - 1) * bpp) & 0x7FFFFFF0


def _entry_points(l):
    """name -> f(w, h): one well-formed call of every geometry-taking entry point of the header (device pointers fake, never dereferenced)"""
    def from_planar(w, h):
        d = _FromPlanar(width=w, height=h, out_data=_P, out_pitch=_pitch(w, 2) & 0xFFFFFFFF, in_depth=8)
        for i in range(3):
            d.in_data[i], d.in_linesize[i] = _P, max(0, min(w, 2 ** 31 - 1))
        return l.ug_hip_from_planar(b"yuv422p_to_uyvy", C.byref(d), None)

    def to_planar(w, h):
        d = _ToPlanar(width=w, height=h, in_data=_P)
        for i in range(3):
            d.out_data[i], d.out_linesize[i] = _P, max(0, min(w, 2 ** 31 - 1))
        return l.ug_hip_to_planar(b"uyvy_to_i420", C.byref(d), None)

    def av(w, h):
        f = _AvFrame(width=w, height=h, colorspace=1, color_range=1)
        for i in range(3):
            f.data[i], f.linesize[i] = _P, max(0, min(w, 2 ** 31 - 1))
        return f

    shifts = (C.c_int * 3)(0, 8, 16)
    ms = C.c_float(0)
    enc = C.c_void_p()
    return {
        "ug_hip_dxt_encode": lambda w, h: l.ug_hip_dxt_encode(lib.PF_UYVY, lib.DXT5_YCOCG, _P, _P, w, h, 0, None),
        "ug_hip_dxt_encode_batch": lambda w, h: l.ug_hip_dxt_encode_batch(lib.PF_RGB, lib.DXT1, _P, _P, w, h, 0, 2, 0, 0, None),
        "ug_hip_dxt_encode_batch_ex": lambda w, h: l.ug_hip_dxt_encode_batch_ex(lib.PF_V210, lib.DXT5_YCOCG, _P, _P, w, h, 0, 2, 0, 0, 0, None),
        "ug_hip_time_dxt_encode": lambda w, h: l.ug_hip_time_dxt_encode(lib.PF_UYVY, lib.DXT5_YCOCG, _P, _P, w, h, 0, 1, 0, 0, 1, None, C.byref(ms)),
        "ug_hip_rgb_to_dxt1": lambda w, h: l.ug_hip_rgb_to_dxt1(_P, _P, w, h, None),
        "ug_hip_yuv_to_dxt1": lambda w, h: l.ug_hip_yuv_to_dxt1(_P, _P, w, h, None),
        "ug_hip_rgb_to_dxt6": lambda w, h: l.ug_hip_rgb_to_dxt6(_P, _P, w, h, None),
        "ug_hip_yuv_to_dxt6": lambda w, h: l.ug_hip_yuv_to_dxt6(_P, _P, w, h, None),
        "ug_hip_dxt_decode": lambda w, h: l.ug_hip_dxt_decode(lib.DXT5_YCOCG, lib.PF_RGBA, _P, _P, w, h, 0, 0, 8, 16, None),
        "ug_hip_dxt_decode_ex": lambda w, h: l.ug_hip_dxt_decode_ex(lib.DXT1, lib.PF_UYVY, _P, _P, w, h, 0, 0, 8, 16, 0, None),
        "ug_hip_pixfmt_convert": lambda w, h: l.ug_hip_pixfmt_convert(lib.PF_UYVY, lib.PF_RGB, _P, _P, w, h, 0, 0, 0, 8, 16, None),
        "ug_hip_pixfmt_convert(ext)": lambda w, h: l.ug_hip_pixfmt_convert(lib.PF_RG48, lib.PF_RGB, _P, _P, w, h, 0, 0, 0, 8, 16, None),
        "ug_hip_pixfmt_convert_batch": lambda w, h: l.ug_hip_pixfmt_convert_batch(lib.PF_V210, lib.PF_UYVY, _P, _P, w, h, 0, 0, 0, 8, 16, 2, 0, 0, None),
        "ug_hip_pixfmt_line_func": lambda w, h: l.ug_hip_pixfmt_line_func(b"vc_copylineUYVYtoGrayscale", _P, _P, w, h, _pitch(w, 2), _pitch(w, 1), max(0, min(w, 2 ** 30)), 0, 8, 16, None),
        "ug_hip_deinterlace_blend": lambda w, h: l.ug_hip_deinterlace_blend(_P, max(0, w) * 2, h, None),
        "ug_hip_deinterlace_blend_batch": lambda w, h: l.ug_hip_deinterlace_blend_batch(_P, max(0, w) * 2, h, 2, max(0, w) * 2 * max(0, h), None),
        "ug_hip_uyvy_to_i420": lambda w, h: l.ug_hip_uyvy_to_i420(_P, _pitch(w, 2), _P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 1), w, h, None),
        "ug_hip_v210_to_p010le": lambda w, h: l.ug_hip_v210_to_p010le(_P, _pitch(w, 3), _P, _pitch(w, 2), _P, _pitch(w, 2), w, h, None),
        "ug_hip_yuv420p_to_uyvy": lambda w, h: l.ug_hip_yuv420p_to_uyvy(_P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 2), w, h, None),
        "ug_hip_yuv422p_to_uyvy": lambda w, h: l.ug_hip_yuv422p_to_uyvy(_P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 2), w, h, None),
        "ug_hip_yuv422p10le_to_v210": lambda w, h: l.ug_hip_yuv422p10le_to_v210(_P, _pitch(w, 2), _P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 3), w, h, None),
        "ug_hip_uyvy_to_i422": lambda w, h: l.ug_hip_uyvy_to_i422(_P, _pitch(w, 2), _P, _pitch(w, 1), _P, _pitch(w, 1), _P, _pitch(w, 1), w, h, None),
        "ug_hip_uyvy_to_nv12": lambda w, h: l.ug_hip_uyvy_to_nv12(_P, _pitch(w, 2), _P, _pitch(w, 1), _P, _pitch(w, 1), w, h, None),
        "ug_hip_from_planar": from_planar,
        "ug_hip_to_planar": to_planar,
        "ug_hip_uv_to_av": lambda w, h: l.ug_hip_uv_to_av(b"UYVY", b"yuv422p", _P, C.byref(av(w, h)), None),
        "ug_hip_av_to_uv": lambda w, h: l.ug_hip_av_to_uv(b"yuv422p", b"UYVY", _P, _pitch(w, 2), C.byref(av(w, h)), shifts, None),
        "ug_hip_jpeg_fdct_quant_plane": lambda w, h: l.ug_hip_jpeg_fdct_quant_plane(_P, _pitch(w, 1), w, h, (max(w, 0) + 7) // 8, (max(h, 0) + 7) // 8, _P, _P, None, None),
        "ug_hip_uyvy_to_jpeg420_coeffs": lambda w, h: l.ug_hip_uyvy_to_jpeg420_coeffs(_P, 0, w, h, _P, _P, _P, _P, None),
        "ug_hip_uyvy_to_jpeg422_coeffs": lambda w, h: l.ug_hip_uyvy_to_jpeg422_coeffs(_P, 0, w, h, _P, _P, _P, _P, None),
        "ug_hip_uyvy_to_jpeg42x_coeffs_batch": lambda w, h: l.ug_hip_uyvy_to_jpeg42x_coeffs_batch(420, _P, 0, w, h, _P, _P, _P, _P, 2, 0, 0, 0, None),
        "ug_hip_jpeg_encoder_create": lambda w, h: l.ug_hip_jpeg_encoder_create(w, h, 75, 4, C.byref(enc)),
        "ug_hip_jpeg_encoder_create_sub": lambda w, h: l.ug_hip_jpeg_encoder_create_sub(w, h, 75, 4, 422, C.byref(enc)),
        "ug_hip_jpeg_encoder_create_ex": lambda w, h: l.ug_hip_jpeg_encoder_create_ex(w, h, 75, 4, 444, lib.JPEG_CS_YCBCR_BT601_256LVLS, lib.JPEG_NONINTERLEAVED, C.byref(enc)),
        "ug_hip_jpeg_colour_convert": lambda w, h: l.ug_hip_jpeg_colour_convert(lib.PF_RGB, lib.JPEG_CS_RGB, lib.JPEG_CS_YCBCR_BT709, _P, 0, _P, 0, w, h, None),
    }


def test_absurd_geometry_is_refused():
    """every geometry-taking entry point of the header x every absurd size: UG_HIP_EINVAL / EUNSUPP -- never success, never a runtime error (that
    would mean a device call was attempted with it)"""
    l = lib.load()
    calls = _entry_points(l)
    hdr = open(os.path.join(ROOT, "include", "ug_mi355x.h")).read()
    hdr = re.sub(r"\s+", " ", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))
    with_geometry = {m.group(1) for m in re.finditer(r"\b(ug_hip_\w+)\s*\(([^)]*)\)\s*;", hdr)
                     if re.search(r"\bint (width|height|size_x|size_y|lines|pix_count)\b|ug_av_frame|planar_data", m.group(2))}
    with_geometry -= {"ug_hip_dxt_size", "ug_hip_linesize", "ug_hip_jpeg_read_info", "ug_hip_jpeg_decoder_plane", "ug_hip_jpeg_decoder_decode_sized"}
    #                  (sizes: below)                           (outputs, not inputs)                               (expected size only compared with the header's)
    with_geometry -= {"ug_hip_yuv422_to_yuv444"}                            # (a pixel count, not a picture: below)
    assert with_geometry == {n.split("(")[0] for n in calls}, with_geometry ^ {n.split("(")[0] for n in calls}
    bad = []
    for name, f in calls.items():
        for w, h in ABSURD:
            if "deinterlace" in name and h == 0:
                continue                                                    # vc_deinterlace leaves fewer than 5 lines alone (video_codec.c:597-664): 0 lines = nothing to do
            rc = f(w, h)
            if rc not in (lib.EINVAL, lib.EUNSUPP):
                bad.append((name, w, h, rc))
    assert not bad, bad
    for n in (-4, 2 ** 31 - 4, 2 ** 30):
        assert l.ug_hip_yuv422_to_yuv444(_P, _P, n, None) == lib.EINVAL, n
    # the 2-D copies: 0 < width_bytes <= both pitches, 0 < rows <= 65536
    for dp, sp, wb, rows in ((64, 64, 65, 4), (64, 32, 64, 4), (64, 64, 0, 4), (64, 64, 64, 0), (64, 64, 64, 65537), (2 ** 31, 64, 64, 4)):
        assert l.ug_hip_memcpy_2d_async(_P, dp, _P, sp, wb, rows, lib.MEMCPY_D2H if hasattr(lib, "MEMCPY_D2H") else 1, None) == lib.EINVAL, (dp, sp, wb, rows)
        assert l.ug_hip_download_2d_ordered_ex(0, _P, dp, _P, sp, wb, rows, None, 0) == lib.EINVAL, (dp, sp, wb, rows)
    assert l.ug_hip_download_2d_ordered_ex(0, _P, 64, _P, 64, 64, 4, None, 4) == lib.EINVAL   # unknown flag
    # the two size helpers: a wrapped int is not an answer
    for fmt in (lib.PF_V210, lib.PF_RGBA, lib.PF_Y416, lib.PF_R12L, lib.PF_UYVY):
        for w in (_BIG, 2 ** 31 - 1, 65537, 0, -1, -2 ** 31):
            assert l.ug_hip_linesize(fmt, w) == lib.EINVAL, (fmt, w, l.ug_hip_linesize(fmt, w))
        assert l.ug_hip_linesize(fmt, 65536) > 0
    for w, h in ABSURD[:8]:
        assert l.ug_hip_dxt_size(lib.DXT5_YCOCG, w, h) == 0, (w, h)
    assert l.ug_hip_dxt_size(lib.DXT5_YCOCG, 65536, 65536) == 2 ** 32      # (a size_t: representable, said as it is)


def test_sane_geometry_passes_validation_without_a_gpu():
    """the control of the test above: the same calls with a sane size get PAST validation -- on a box without a GPU that shows as UG_HIP_ERUNTIME
    (the launch fails), so the refusals above are about the sizes and nothing else.  Not run where a GPU is present (the pointers are fake)."""
    l = lib.load()
    n = C.c_int(0)
    if l.ug_hip_device_count(C.byref(n)) == lib.SUCCESS and n.value > 0:
        import pytest
        pytest.skip("a GPU is present: fake device pointers must not be launched on")
    for name, f in _entry_points(l).items():
        rc = f(96, 32)
        assert rc == lib.ERUNTIME, (name, rc, l.ug_hip_last_error_string())
    assert l.ug_hip_yuv422_to_yuv444(_P, _P, 96 * 32, None) == lib.ERUNTIME
