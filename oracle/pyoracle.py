"""ctypes front-end to the CPU oracles -- TEST INFRASTRUCTURE ONLY.

Two libraries:
  * ``oracle/liboracle_ug.so``  -- our plain-C restatements (oracle/*.c).
  * ``oracle/_ref/libugref.so`` -- the reference's OWN pixfmt C compiled from
    /root/reference by ``make -C oracle ref`` (present in the build container and, as a
    prebuilt file, on the GPU box).  Used to pin the restatement and as the
    ``cpu_baseline.kind == "reference"`` timing leg.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this module.
The product package (ultragrid_amd) must never import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_ug.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libugref.so")
# same sources built without -msse4.1 (portable scalar paths; see oracle/Makefile)
_REF_SCALAR_PATH = os.path.join(_HERE, "_ref", "libugref_scalar.so")

# our format ids (oracle.h)
IN_RGB, IN_RGBA, IN_YUV444, IN_UYVY, IN_UYVY_RAW, IN_V210 = range(6)
OUT_DXT1, OUT_DXT1_YUV, OUT_DXT5YCOCG = 1, 2, 6
OPF = dict(RGBA=1, UYVY=2, YUYV=3, RGB=4, BGR=5, v210=6, RG48=7, I420=8)
# reference codec_t values (src/types.h:62-112)
REF_CODEC = dict(RGBA=1, UYVY=2, YUYV=3, v210=7, DXT1=9, DXT5=11, RGB=12, BGR=20, RG48=27, I420=29)

_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> None:
    """Compile oracle/*.c (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(_HERE) if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/src") and (force or not have_ref() or not os.path.exists(os.path.join(_HERE, "_ref", "glsl_ref"))
                                                 or not os.path.exists(os.path.join(_HERE, "_ref", "libugref_lavc.so"))
                                                 or not os.path.exists(os.path.join(_HERE, "_ref", "libugref_lavc_hook.so"))
                                                 or not os.path.exists(os.path.join(_HERE, "_ref", "ug_ref_lavc_test"))
                                                 or not os.path.exists(os.path.join(_HERE, "_ref", "ug_ref_codec_test"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_refs: dict = {}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_dxt_encode.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long]
        _lib.oracle_dxt_encode.restype = C.c_int
        _lib.oracle_dxt_encode_mt.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int]
        _lib.oracle_dxt_encode_mt.restype = C.c_int
        _lib.oracle_yuv422_to_yuv444.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        _lib.oracle_yuv422_to_yuv444.restype = None
        for n in ("oracle_dxt5ycocg_decode_rgb", "oracle_dxt1_decode_rgb"):
            getattr(_lib, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            getattr(_lib, n).restype = None
        _lib.oracle_dxt_decode.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int]
        _lib.oracle_dxt_decode.restype = C.c_int
        _lib.oracle_linesize.argtypes = [C.c_int, C.c_int]
        _lib.oracle_size.argtypes = [C.c_int, C.c_int]
        _lib.oracle_convert_line.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 3
        _lib.oracle_convert_frame.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_int] * 3
        _lib.oracle_uyvy_to_i420.argtypes = [C.c_void_p, C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int]
        _lib.oracle_uyvy_to_i420.restype = None
        _lib.oracle_v210_to_p010le.argtypes = [C.c_void_p, C.c_int] * 2 + [C.c_void_p, C.c_int, C.c_int]
        _lib.oracle_v210_to_p010le.restype = None
        _lib.oracle_color_coeffs.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
        _lib.oracle_jpeg_qtable.argtypes = [C.c_int, C.c_int, C.c_void_p]
        _lib.oracle_jpeg_qtable.restype = None
        _lib.oracle_jpeg_divisors.argtypes = [C.c_void_p, C.c_void_p]
        _lib.oracle_jpeg_divisors.restype = None
        _lib.oracle_jpeg_fdct_quant_plane.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3
        _lib.oracle_jpeg_fdct_quant_plane.restype = None
        _lib.oracle_dxt5ycocg_encode_block.argtypes = [C.c_void_p, C.c_void_p]
        _lib.oracle_dxt1_encode_block.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


def have_ref() -> bool:
    return os.path.exists(_REF_PATH) and os.path.exists(_REF_SCALAR_PATH) and os.path.exists(os.path.join(_HERE, "_ref", "dxt62tga"))


class _ToPlanar(C.Structure):  # src/to_planar.h:53-59
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p * 4),
                ("out_linesize", C.c_uint * 4), ("in_data", C.c_void_p)]


def ref(scalar: bool = False) -> C.CDLL:
    """The compiled reference (raises if it was not built).  scalar=True: the build
    without -msse4.1 (the reference's portable code paths)."""
    path = _REF_SCALAR_PATH if scalar else _REF_PATH
    _ref = _refs.get(path)
    if _ref is None:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        _ref = _refs[path] = C.CDLL(path)
        _ref.get_decoder_from_to.argtypes = [C.c_int, C.c_int]
        _ref.get_decoder_from_to.restype = C.c_void_p
        _ref.vc_get_linesize.argtypes = [C.c_uint, C.c_int]
        _ref.vc_get_size.argtypes = [C.c_uint, C.c_int]
        _ref.get_color_coeffs.argtypes = [C.c_int, C.c_int]
        _ref.get_color_coeffs.restype = C.c_void_p
        for n in ("uyvy_to_i420", "v210_to_p010le"):
            getattr(_ref, n).argtypes = [_ToPlanar]
            getattr(_ref, n).restype = None
    return _ref


_DECODER_T = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
MAX_PADDING = 64  # src/video_codec.h:61


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------------
# DXT
# ----------------------------------------------------------------------------------
def set_ties(ties: str) -> None:
    """GLSL's implementation-defined choices, all at once: "even" (default; Mesa llvmpipe's, pinned to the executed reference shaders:
    round() ties to even, dot(vec3) from the last component, unorm8 writes tie to even) or "away" (the reference's CUDA text: roundf,
    left-to-right dot, floor(x*255+0.5)).  Same meaning as UG_DXT_TIES_EVEN / UG_DXT_TIES_AWAY of the product library."""
    lib().oracle_set_ties({"even": 0, "away": 1}[ties])


def dxt_size(out_fmt: int, w: int, h: int) -> int:
    """dxt_get_size (dxt_compress/dxt_util.h:59-67): both dimensions rounded up to whole 4x4 blocks."""
    n = ((w + 3) // 4 * 4) * ((abs(h) + 3) // 4 * 4)
    return n // 2 if out_fmt == OUT_DXT1 else n


def dxt_encode(in_fmt: int, out_fmt: int, src: np.ndarray, w: int, h: int, pitch: int | None = None,
               threads: int = 1, ties: str = "even") -> np.ndarray:
    """h < 0 => bottom-up source (cuda_dxt.cu:652-655).  threads != 1: OpenMP row bands (0 = all cores)."""
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    if pitch is None:
        pitch = {IN_RGB: 3 * w, IN_RGBA: 4 * w, IN_YUV444: 3 * w, IN_UYVY: 2 * w, IN_UYVY_RAW: 2 * w,
                 IN_V210: (w + 47) // 48 * 128}[in_fmt]
    n = dxt_size(out_fmt, w, h)
    out = np.zeros(n, dtype=np.uint8)
    set_ties(ties)
    if threads == 1:
        rc = lib().oracle_dxt_encode(in_fmt, out_fmt, _ptr(src), _ptr(out), w, h, pitch)
    else:
        rc = lib().oracle_dxt_encode_mt(in_fmt, out_fmt, _ptr(src), _ptr(out), w, h, pitch, threads)
    if rc:
        raise ValueError(f"oracle_dxt_encode rc={rc}")
    return out


def yuv422_to_yuv444(src: np.ndarray, pix: int) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    out = np.zeros(pix * 3, np.uint8)
    lib().oracle_yuv422_to_yuv444(_ptr(src), _ptr(out), pix)
    return out


def dxt_decode_rgb(out_fmt: int, blocks: np.ndarray, w: int, h: int) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).ravel()
    out = np.zeros(w * h * 3, np.uint8)
    fn = lib().oracle_dxt5ycocg_decode_rgb if out_fmt == OUT_DXT5YCOCG else lib().oracle_dxt1_decode_rgb
    fn(_ptr(blocks), _ptr(out), w, h)
    return out.reshape(h, w, 3)


def dxt_decode(in_fmt: int, out_fmt: str, blocks: np.ndarray, w: int, h: int, shifts=(0, 8, 16), ties: str = "even") -> np.ndarray:
    """Frame decode to RGB / BGR / RGBA / UYVY (oracle/dxt_decode_oracle.c)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).ravel()
    pitch = linesize(w, out_fmt)
    out = np.zeros(pitch * h, np.uint8)
    set_ties(ties)
    rc = lib().oracle_dxt_decode(in_fmt, OPF[out_fmt], _ptr(blocks), _ptr(out), w, h, pitch, *shifts)
    if rc:
        raise ValueError(f"oracle_dxt_decode rc={rc}")
    return out


def deinterlace_blend(frame: np.ndarray, linesize: int, lines: int) -> np.ndarray:
    """vc_deinterlace (video_codec.c:597-664, SSE2 bodies) on a copy of `frame`."""
    out = np.ascontiguousarray(frame, dtype=np.uint8).ravel().copy()
    assert out.size >= linesize * lines
    assert linesize >= 16 or lines < 5, "the restatement holds from one 16-byte column up (oracle/pixfmt_oracle.c)"
    lib().oracle_deinterlace_blend.restype = None
    lib().oracle_deinterlace_blend.argtypes = [C.c_void_p, C.c_long, C.c_int]
    lib().oracle_deinterlace_blend(_ptr(out), linesize, lines)
    return out


def ref_deinterlace(frame: np.ndarray, linesize: int, lines: int) -> np.ndarray:
    """The compiled reference's own vc_deinterlace (oracle/_ref/libugref.so) on a copy of `frame`."""
    out = np.ascontiguousarray(frame, dtype=np.uint8).ravel().copy()
    r = ref()
    r.vc_deinterlace.restype = None
    r.vc_deinterlace.argtypes = [C.c_void_p, C.c_long, C.c_int]
    r.vc_deinterlace(_ptr(out), linesize, lines)
    return out


GLSL_REF = os.path.join(_HERE, "_ref", "glsl_ref")


def have_glsl_ref() -> bool:
    return os.path.exists(GLSL_REF) and os.path.isdir("/root/reference/dxt_compress")


def ref_glsl_dxt_encode(mode: str, fmt: str, src: np.ndarray, w: int, h: int, gl_row_stride: bool = False) -> np.ndarray:
    """The reference's own GLSL encoder (dxt_compress/compress_*_fp.glsl) executed by Mesa llvmpipe through oracle/_ref/glsl_ref.
    mode: dxt5 | dxt1 | dxt1yuv ; fmt: rgb | rgba | yuv444 | uyvy.
    gl_row_stride (rgb only): lay the lines out at the stride GL READS them with -- the reference never calls glPixelStorei, so its
    glTexSubImage2D(GL_RGB) takes lines at GL's default 4-byte unpack alignment, (3 w + 3) & ~3, whatever the caller packed
    (dxt_encoder.c:562-575).  For 3 w % 4 == 0 the two layouts are the same."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in.raw"), os.path.join(d, "out.bin")
        src = np.ascontiguousarray(src, np.uint8)
        if gl_row_stride and fmt == "rgb" and (3 * w) % 4:
            lines = np.zeros((h, (3 * w + 3) // 4 * 4), np.uint8)
            lines[:, :3 * w] = src.reshape(h, 3 * w)
            src = lines
        src.tofile(a)
        r = subprocess.run([GLSL_REF, "/root/reference", mode, fmt, str(w), str(h), a, b], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return np.fromfile(b, np.uint8)


DXT62TGA = os.path.join(_HERE, "_ref", "dxt62tga")


def ref_dxt62tga(blocks: np.ndarray, w: int, h: int) -> np.ndarray:
    """Run the reference's own stand-alone DXT5-YCoCg decoder (cuda_dxt/dxt62tga.c, compiled to oracle/_ref/dxt62tga)
    and return its image as RGB rows top-down."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "in.yog"), os.path.join(d, "out.tga")
        np.ascontiguousarray(blocks, np.uint8).tofile(src)
        subprocess.check_call([DXT62TGA, str(w), str(h), src, out], stdout=subprocess.DEVNULL)
        raw = np.fromfile(out, np.uint8)
    hdr, data = raw[:18], raw[18:18 + w * h * 3].reshape(h, w, 3)
    if not (hdr[17] & 0x20):      # TGA origin bit: 0 = bottom-left
        data = data[::-1]
    return np.ascontiguousarray(data[..., ::-1])  # BGR -> RGB


# ----------------------------------------------------------------------------------
# pixfmt: restatement and compiled reference behind the same call shape
# ----------------------------------------------------------------------------------
def linesize(width: int, fmt: str) -> int:
    return lib().oracle_linesize(width, OPF[fmt])


def convert_frame(in_fmt: str, out_fmt: str, src: np.ndarray, w: int, h: int, shifts=(0, 8, 16)) -> np.ndarray:
    """Restatement (oracle/pixfmt_oracle.c); line loop of testcard_common.c:121-129."""
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    sls, dls = linesize(w, in_fmt), linesize(w, out_fmt)
    assert src.size >= sls * h
    src = np.concatenate([src, np.zeros(MAX_PADDING, np.uint8)])
    out = np.zeros(dls * h + MAX_PADDING, np.uint8)
    rc = lib().oracle_convert_frame(OPF[in_fmt], OPF[out_fmt], _ptr(out), _ptr(src), w, h, *shifts)
    if rc:
        raise ValueError(f"no restated decoder {in_fmt}->{out_fmt}")
    return out[: dls * h]


def ref_convert_frame(in_fmt: str, out_fmt: str, src: np.ndarray, w: int, h: int, shifts=(0, 8, 16),
                      scalar: bool = False) -> np.ndarray:
    """The compiled reference: get_decoder_from_to() per line (pixfmt_conv.c:3110)."""
    r = ref(scalar)
    fn = r.get_decoder_from_to(REF_CODEC[in_fmt], REF_CODEC[out_fmt])
    if not fn:
        raise ValueError(f"reference has no decoder {in_fmt}->{out_fmt}")
    dec = _DECODER_T(fn)
    sls, dls = r.vc_get_linesize(w, REF_CODEC[in_fmt]), r.vc_get_linesize(w, REF_CODEC[out_fmt])
    dsz = r.vc_get_size(w, REF_CODEC[out_fmt])
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    assert src.size >= sls * h
    src = np.concatenate([src, np.zeros(MAX_PADDING, np.uint8)])
    out = np.zeros(dls * h + MAX_PADDING, np.uint8)
    sp, dp = src.ctypes.data, out.ctypes.data
    for y in range(h):
        dec(dp + y * dls, sp + y * sls, dsz, *shifts)
    return out[: dls * h]


def uyvy_to_i420(src: np.ndarray, w: int, h: int, use_ref: bool = False):
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    cw, ch = (w + 1) // 2, (h + 1) // 2
    y = np.zeros((h, w), np.uint8)
    u = np.zeros((ch, cw), np.uint8)
    v = np.zeros((ch, cw), np.uint8)
    if use_ref:
        d = _ToPlanar()
        d.width, d.height = w, h
        d.out_data[0], d.out_data[1], d.out_data[2] = y.ctypes.data, u.ctypes.data, v.ctypes.data
        d.out_linesize[0], d.out_linesize[1], d.out_linesize[2] = w, cw, cw
        d.in_data = src.ctypes.data
        ref().uyvy_to_i420(d)
    else:
        lib().oracle_uyvy_to_i420(_ptr(y), w, _ptr(u), cw, _ptr(v), cw, _ptr(src), w, h)
    return y, u, v


def uyvy_to_i422(src: np.ndarray, w: int, h: int, use_ref: bool = False):
    """uyvy_to_i422 (video_codec.c:949-969): Y, Cb, Cr planes, chroma (w+1)/2 wide, samples copied as they are.  Lines are
    vc_get_linesize(UYVY) apart; for odd widths the reference function itself assumes 2*w+1 bytes per line, which contradicts
    its own linesize, so use_ref is only meaningful for even widths."""
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    cw = (w + 1) // 2
    if use_ref:
        assert w % 2 == 0
        out = np.zeros(w * h + 2 * cw * h, np.uint8)
        fn = ref().uyvy_to_i422
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        fn(w, h, src.ctypes.data, out.ctypes.data)
        return out[: w * h].reshape(h, w), out[w * h: w * h + cw * h].reshape(h, cw), out[w * h + cw * h:].reshape(h, cw)
    ls = 4 * cw
    rows = src[: ls * h].reshape(h, ls)
    return (np.ascontiguousarray(rows[:, 1::2][:, :w]), np.ascontiguousarray(rows[:, 0::4]), np.ascontiguousarray(rows[:, 2::4]))


def uyvy_to_nv12(src: np.ndarray, w: int, h: int, src_pitch: int = 0, use_ref: bool = False, scalar: bool = False):
    """uyvy_to_nv12 (to_planar.c:207-302).  Default-build arithmetic: (a+b+1)>>1 for the first 16*(w/16) pixels (SSE avg), (a+b)/2 for
    the tail; scalar=True: the build without SSE3 (all truncating).  The reference strides its input by 2*width bytes; pass
    src_pitch=2*w to compare odd widths against it."""
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    cw, ch = (w + 1) // 2, (h + 1) // 2
    ls = src_pitch or linesize(w, "UYVY")
    if use_ref:
        assert ls == 2 * w
        y = np.zeros((h, w), np.uint8)
        c = np.zeros((ch, 2 * cw + 16), np.uint8)
        d = _ToPlanar()
        d.width, d.height = w, h
        d.out_data[0], d.out_data[1] = y.ctypes.data, c.ctypes.data
        d.out_linesize[0], d.out_linesize[1] = w, c.strides[0]
        pad = np.concatenate([src, np.zeros(MAX_PADDING, np.uint8)])
        d.in_data = pad.ctypes.data
        fn = ref(scalar).uyvy_to_nv12
        fn.restype, fn.argtypes = None, [_ToPlanar]
        fn(d)
        return y, np.ascontiguousarray(c[:, : 2 * cw])
    need = ls * (h - 1) + 4 * cw
    buf = np.concatenate([src, np.zeros(max(0, ls * h - src.size) + 8, np.uint8)])
    rows = np.stack([buf[r * ls: r * ls + 4 * cw] for r in range(h)]).astype(np.uint16)
    top = rows[0::2]
    bot = rows[1::2] if h % 2 == 0 else np.concatenate([rows[1::2], rows[-1:]])
    vec = 0 if scalar else 16 * (w // 16)
    rnd = (np.arange(cw) * 2 < vec).astype(np.uint16)
    c = np.zeros((ch, 2 * cw), np.uint8)
    c[:, 0::2] = ((top[:, 0::4] + bot[:, 0::4] + rnd) >> 1).astype(np.uint8)
    c[:, 1::2] = ((top[:, 2::4] + bot[:, 2::4] + rnd) >> 1).astype(np.uint8)
    y = np.zeros((h, 2 * cw), np.uint8)
    y[:, 0::2] = rows[:, 1::4]
    y[:, 1::2] = rows[:, 3::4]
    return np.ascontiguousarray(y[:, :w]), c


class _FromPlanar(C.Structure):
    """struct from_planar_data (from_planar.h:58-70), passed by value"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p), ("out_pitch", C.c_uint),
                ("in_data", C.c_void_p * 4), ("in_linesize", C.c_uint * 4), ("in_depth", C.c_int), ("log2_chroma_h", C.c_int),
                ("rgb_shift", C.c_int * 3)]


def _ref_from_planar(name: str, planes, w: int, h: int, out: np.ndarray, out_pitch: int, depth: int = 8, scalar: bool = False):
    d = _FromPlanar()
    d.width, d.height = w, h
    d.out_data, d.out_pitch = out.ctypes.data, out_pitch
    for i, pl in enumerate(planes):
        d.in_data[i] = pl.ctypes.data
        d.in_linesize[i] = pl.strides[0]
    d.in_depth = depth
    fn = getattr(ref(scalar), name)
    fn.restype = None
    fn.argtypes = [_FromPlanar]
    fn(d)


def planar_to_uyvy(y: np.ndarray, u: np.ndarray, v: np.ndarray, w: int, h: int, chroma: int = 420, use_ref: bool = False) -> np.ndarray:
    """yuv420p_to_uyvy (from_planar.c:583-683) / yuv422p_to_uyvy (:391-423).  8-bit planes, chroma planes (w+1)/2 wide."""
    ls = linesize(w, "UYVY")
    y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
    if use_ref:
        out = np.zeros(ls * h + MAX_PADDING, np.uint8)
        # the SSE3 loop of yuv420p_to_uyvy runs `x < width - 15` on an unsigned width: widths below 15 wrap around and crash the
        # -msse4.1 build, so those go to the build without it (same scalar statements)
        _ref_from_planar("yuv420p_to_uyvy" if chroma == 420 else "yuv422p_to_uyvy", (y, u, v), w, h, out, ls, scalar=w < 16)
        return out[: ls * h]
    out = np.zeros((h, ls), np.uint8)
    rows = np.arange(h)
    crow = rows // 2 if chroma == 420 else rows
    npair = (w + 1) // 2 if chroma == 420 else w // 2
    out[:, 0:4 * npair:4] = u[crow, :npair]
    out[:, 2:4 * npair:4] = v[crow, :npair]
    out[:, 1:4 * (w // 2):4] = y[:, 0:2 * (w // 2):2]
    out[:, 3:4 * (w // 2):4] = y[:, 1:2 * (w // 2):2]
    if chroma == 420 and w % 2:          # odd width: Cb Y Cr 0 (from_planar.c:669-680)
        out[:, 4 * (w // 2) + 1] = y[:, w - 1]
    return out.ravel()


def yuv422p10le_to_v210(y: np.ndarray, u: np.ndarray, v: np.ndarray, w: int, h: int, use_ref: bool = False) -> np.ndarray:
    """yuv422p10le_to_v210 (from_planar.c:296-333): width / 6 groups per line, samples OR-ed in unmasked."""
    ls = linesize(w, "v210")
    y, u, v = (np.ascontiguousarray(a, dtype=np.uint16) for a in (y, u, v))
    if use_ref:
        out = np.zeros(ls * h + MAX_PADDING, np.uint8)
        _ref_from_planar("yuv422p10le_to_v210", (y, u, v), w, h, out, ls, depth=10)
        return out[: ls * h]
    g = w // 6
    out = np.zeros((h, ls // 4), np.uint32)
    Y = y[:, : 6 * g].astype(np.uint32).reshape(h, g, 6)
    U = u[:, : 3 * g].astype(np.uint32).reshape(h, g, 3)
    V = v[:, : 3 * g].astype(np.uint32).reshape(h, g, 3)
    words = np.stack([U[..., 0] | Y[..., 0] << 10 | V[..., 0] << 20, Y[..., 1] | U[..., 1] << 10 | Y[..., 2] << 20,
                      V[..., 1] | Y[..., 3] << 10 | U[..., 2] << 20, Y[..., 4] | V[..., 2] << 10 | Y[..., 5] << 20], -1)
    out[:, : 4 * g] = words.reshape(h, 4 * g)
    return out.view(np.uint8).ravel()


def v210_to_p010le(src: np.ndarray, w: int, h: int, use_ref: bool = False, y_pad: int = 0, uv_pad: int = 0, fill: int = 0):
    """to_planar.c:64-155; planes of h x (w + y_pad) and ceil(h / 2) x (w + uv_pad) samples (pads = line padding in samples,
    returned too: the reference writes the whole last group of a line there).  With width % 6 != 0 and a pad < roundup6(w) - w the
    reference's line tails overlap the following lines; spare lines behind each plane take the last tail."""
    src = np.ascontiguousarray(src, dtype=np.uint8).ravel()
    ch = (h + 1) // 2
    ybuf = np.full((h + 1 + 6 // (w + y_pad), w + y_pad), fill, np.uint16)
    uvbuf = np.full((ch + 1 + 6 // (w + uv_pad), w + uv_pad), fill, np.uint16)
    if use_ref:
        d = _ToPlanar()
        d.width, d.height = w, h
        d.out_data[0], d.out_data[1] = ybuf.ctypes.data, uvbuf.ctypes.data
        d.out_linesize[0], d.out_linesize[1] = 2 * (w + y_pad), 2 * (w + uv_pad)
        d.in_data = src.ctypes.data
        ref().v210_to_p010le(d)
    else:
        lib().oracle_v210_to_p010le(_ptr(ybuf), 2 * (w + y_pad), _ptr(uvbuf), 2 * (w + uv_pad), _ptr(src), w, h)
    return ybuf[:h], uvbuf[:ch]


def color_coeffs(depth: int, bt601: bool = False) -> list[int]:
    o = (C.c_int * 14)()
    if lib().oracle_color_coeffs(int(bt601), depth, o):
        raise ValueError("bad depth")
    return list(o)


class _RefCoeffs(C.Structure):  # src/color_space.h:135-148
    _fields_ = [(n, C.c_short) for n in
                "y_r y_g y_b cb_r cb_g cb_b cr_r cr_g cr_b y_scale r_cr g_cb g_cr".split()] + [("b_cb", C.c_int)]


def ref_color_coeffs(depth: int) -> list[int]:
    p = ref().get_color_coeffs(0, depth)  # CS_DFL
    s = _RefCoeffs.from_address(p)
    return [getattr(s, n) for n, _ in _RefCoeffs._fields_]


# ----------------------------------------------------------------------------------
# JPEG FDCT + quant
# ----------------------------------------------------------------------------------
def jpeg_qtable(quality: int, comp: int) -> np.ndarray:
    t = np.zeros(64, np.uint8)
    lib().oracle_jpeg_qtable(quality, comp, _ptr(t))
    return t


def jpeg_divisors(qtable: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(qtable, np.uint8)
    d = np.zeros(64, np.float32)
    lib().oracle_jpeg_divisors(_ptr(q), _ptr(d))
    return d


def jpeg_fdct_quant_plane(plane: np.ndarray, div: np.ndarray, blocks_w: int | None = None,
                          blocks_h: int | None = None, want_coef: bool = False):
    plane = np.ascontiguousarray(plane, np.uint8)
    h, w = plane.shape
    bw = blocks_w or (w + 7) // 8
    bh = blocks_h or (h + 7) // 8
    out = np.zeros((bh * bw, 64), np.int16)
    coef = np.zeros((bh * bw, 64), np.float32) if want_coef else None
    div = np.ascontiguousarray(div, np.float32)
    lib().oracle_jpeg_fdct_quant_plane(_ptr(plane), w, w, h, bw, bh, _ptr(div), _ptr(out),
                                       _ptr(coef) if want_coef else None)
    return (out, coef) if want_coef else out


def jpeg_decode_planes(data: bytes):
    """oracle/jpeg_decode_oracle.c: (info dict, [planes at their own resolution, cropped to the component size])."""
    l = lib()
    l.oracle_jpeg_decode.restype = C.c_int
    l.oracle_jpeg_decode.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    info = (C.c_int * 12)()
    rc = l.oracle_jpeg_decode(data, len(data), info, None, None)
    if rc:
        raise ValueError(f"oracle_jpeg_decode rc={rc}")
    w, h, nc = info[0], info[1], info[2]
    hs, vs = [info[3], info[5], info[7]][:nc], [info[4], info[6], info[8]][:nc]
    hmax, vmax = max(hs), max(vs)
    mw, mh = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    planes = [np.zeros((mh * 8 * vs[c], mw * 8 * hs[c]), np.uint8) for c in range(nc)]
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes] + [None] * (3 - nc))
    pitch = (C.c_int * 3)(*[p.shape[1] for p in planes] + [0] * (3 - nc))
    rc = l.oracle_jpeg_decode(data, len(data), info, ptrs, pitch)
    if rc:
        raise ValueError(f"oracle_jpeg_decode rc={rc}")
    d = dict(width=w, height=h, components=nc, h=hs, v=vs, restart=info[9], adobe=info[10], scans=info[11])
    crop = [planes[c][: -(-h * vs[c] // vmax), : -(-w * hs[c] // hmax)] for c in range(nc)]
    return d, crop, planes


def jpeg_colour_matrix(cs_in: int, cs_out: int) -> np.ndarray:
    """oracle/jpeg_oracle.c: the 3 x 4 affine map of the encoder's colour stage on 8-bit code values (1 = RGB, 2 = BT.601, 3 = BT.601 256 levels, 4 = BT.709)"""
    m = np.zeros(12, np.float32)
    if lib().oracle_jpeg_colour_matrix(cs_in, cs_out, _ptr(m)):
        raise ValueError("oracle_jpeg_colour_matrix")
    return m.reshape(3, 4)


def jpeg_colour_convert(fmt: str, cs_in: int, cs_out: int, src: np.ndarray, w: int, h: int) -> np.ndarray:
    """fmt "RGB" (3 bytes per pixel, whatever they mean), "UYVY", or "UYVY444" (UYVY -> 3 bytes per pixel, every pixel with its pair's chroma); packed lines"""
    src = np.ascontiguousarray(src, np.uint8).ravel()
    out = np.zeros(3 * w * h, np.uint8) if fmt == "UYVY444" else np.zeros_like(src)
    if lib().oracle_jpeg_colour_convert({"RGB": 0, "UYVY": 1, "UYVY444": 2}[fmt], cs_in, cs_out, _ptr(src), _ptr(out), w, h):
        raise ValueError("oracle_jpeg_colour_convert")
    return out


def jpeg_decode_coeffs(data: bytes):
    """The entropy decoder of oracle/jpeg_decode_oracle.c alone: (info dict, [quantised coefficients of each component as the stream codes them:
    (blocks of the MCU-padded grid in raster order, 64) int16, zig-zag order -- jpeg_fdct_quant_plane's layout])."""
    l = lib()
    l.oracle_jpeg_decode_coeffs.restype = C.c_int
    l.oracle_jpeg_decode_coeffs.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    info = (C.c_int * 12)()
    rc = l.oracle_jpeg_decode_coeffs(data, len(data), info, None)
    if rc:
        raise ValueError(f"oracle_jpeg_decode_coeffs rc={rc}")
    w, h, nc = info[0], info[1], info[2]
    hs, vs = [info[3], info[5], info[7]][:nc], [info[4], info[6], info[8]][:nc]
    hmax, vmax = max(hs), max(vs)
    mw, mh = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    coefs = [np.zeros((mh * vs[c] * mw * hs[c], 64), np.int16) for c in range(nc)]
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in coefs] + [None] * (3 - nc))
    rc = l.oracle_jpeg_decode_coeffs(data, len(data), info, ptrs)
    if rc:
        raise ValueError(f"oracle_jpeg_decode_coeffs rc={rc}")
    return dict(width=w, height=h, components=nc, h=hs, v=vs, restart=info[9], adobe=info[10], scans=info[11]), coefs


ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
    27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)
