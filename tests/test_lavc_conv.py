"""the lavc pixel-format converters (libavcodec/to_lavc_vid_conv.c, from_lavc_vid_conv.c) on the GPU against the
reference's OWN functions, compiled from /root/reference where they lie (oracle/_ref/libugref_lavc.so: the two reference files +
oracle/lavc_stub/, a stand-in for the few FFmpeg declarations they name).  The prebuilt .so travels to the GPU box; the GPU tests
call it through ctypes on the box's CPU and compare byte for byte -- whole planes / whole buffers including the padding."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LAVC = os.path.join(HERE, "..", "oracle", "_ref", "libugref_lavc.so")

TO_AV = [("UYVY", "yuv420p"), ("UYVY", "yuv422p"), ("UYVY", "yuv444p"), ("UYVY", "yuvj444p"), ("UYVY", "nv12"), ("UYVY", "vuya"), ("UYVY", "vuyx"),
         ("v210", "yuv420p10le"), ("v210", "yuv422p10le"), ("v210", "yuv444p10le"), ("v210", "yuv444p16le"), ("v210", "p010le"),
         ("v210", "p210le"), ("v210", "xv30le"), ("v210", "y210le"), ("v210", "y212le"),
         ("RGB", "bgr0"), ("RGB", "gbrp"), ("RGB", "yuv444p"), ("RGBA", "gbrp"), ("RGBA", "bgra"),
         ("Y216", "y210le"), ("Y216", "y212le"), ("Y216", "p010le"), ("Y216", "yuv422p10le"), ("Y216", "yuv422p16le"), ("Y216", "yuv444p16le"),
         ("Y416", "xv30le"), ("Y416", "yuv444p"), ("Y416", "yuv444p10le"), ("Y416", "yuv444p12le"), ("Y416", "yuv444p16le"),
         ("R10k", "yuv444p10le"), ("R10k", "yuv444p12le"), ("R10k", "yuv444p16le"), ("R10k", "yuv422p10le"), ("R10k", "yuv420p10le"),
         ("R10k", "gbrp10le"), ("R10k", "gbrp16le"), ("R10k", "x2rgb10le"), ("R10k", "bgr0"),
         ("R12L", "yuv444p10le"), ("R12L", "yuv444p12le"), ("R12L", "yuv444p16le"), ("R12L", "yuv422p10le"), ("R12L", "yuv422p12le"),
         ("R12L", "yuv422p16le"), ("R12L", "p210le"), ("R12L", "ayuv64le"), ("R12L", "gbrp12le"), ("R12L", "gbrp16le"),
         ("RG48", "yuv444p10le"), ("RG48", "yuv444p12le"), ("RG48", "yuv444p16le"), ("RG48", "gbrp12le")]
FORWARDED_TO = {("UYVY", "yuv420p"), ("UYVY", "yuv422p"), ("UYVY", "nv12"), ("v210", "p010le"), ("RGB", "bgr0"), ("RGBA", "bgra"),
                ("Y216", "p010le"), ("R12L", "gbrp12le"), ("R12L", "gbrp16le"), ("Y216", "y210le"), ("Y216", "y212le")}

FROM_AV = [("yuv420p10le", "v210"), ("yuv420p10le", "UYVY"), ("yuv420p10le", "RGB"), ("yuv420p10le", "RGBA"), ("yuv420p10le", "R10k"),
           ("yuv422p10le", "v210"), ("yuv422p10le", "UYVY"), ("yuv422p10le", "RGB"), ("yuv422p10le", "RGBA"), ("yuv422p10le", "R10k"),
           ("yuv444p10le", "v210"), ("yuv444p10le", "UYVY"), ("yuv444p10le", "RGB"), ("yuv444p10le", "RGBA"),
           ("yuv444p12le", "v210"), ("yuv444p12le", "UYVY"), ("yuv444p16le", "v210"), ("yuv444p16le", "UYVY"),
           ("p210le", "v210"), ("p210le", "UYVY"), ("p010le", "v210"), ("p010le", "UYVY"),
           ("yuv420p", "v210"), ("yuv420p", "UYVY"), ("yuv420p", "RGB"), ("yuv420p", "RGBA"),
           ("yuv422p", "v210"), ("yuv422p", "UYVY"), ("yuv422p", "RGB"), ("yuv422p", "RGBA"),
           ("yuv444p", "v210"), ("yuv444p", "UYVY"), ("yuv444p", "RGB"), ("yuv444p", "RGBA"), ("yuv444p", "VUYA"),
           ("yuvj420p", "RGB"), ("yuvj422p", "RGBA"), ("yuvj444p", "UYVY"),
           ("nv12", "UYVY"), ("nv12", "RGB"), ("nv12", "RGBA"), ("gbrap", "RGB"), ("gbrap", "RGBA"), ("gbrp", "RGB"), ("gbrp", "RGBA"),
           ("rgb24", "UYVY"), ("rgb24", "RGBA"),
           ("gbrp10le", "R10k"), ("gbrp10le", "RGB"), ("gbrp10le", "RGBA"), ("gbrp10le", "RG48"), ("gbrp12le", "R12L"), ("gbrp12le", "R10k"),
           ("gbrp12le", "RGB"), ("gbrp12le", "RGBA"), ("gbrp12le", "RG48"), ("gbrp16le", "R12L"), ("gbrp16le", "R10k"), ("gbrp16le", "RG48"),
           ("yuv444p10le", "R10k"), ("yuv444p10le", "R12L"), ("yuv444p10le", "RG48"), ("yuv444p10le", "Y416"),
           ("yuv444p12le", "R10k"), ("yuv444p12le", "R12L"), ("yuv444p12le", "RG48"), ("yuv444p12le", "Y416"),
           ("yuv444p16le", "R10k"), ("yuv444p16le", "R12L"), ("yuv444p16le", "RG48"), ("yuv444p16le", "Y416"),
           ("xv30le", "UYVY"), ("xv30le", "v210"), ("xv30le", "Y416"), ("y210le", "UYVY"), ("y210le", "v210"), ("y210le", "Y416"),
           ("y212le", "UYVY"), ("y212le", "v210"), ("y212le", "Y416"), ("ayuv64le", "v210"), ("ayuv64le", "Y416"),
           ("vuya", "UYVY"), ("vuyx", "UYVY"), ("vuya", "Y416"), ("vuyx", "Y416"), ("rgb48le", "RGBA"), ("rgb48le", "R12L")]


class StubAVFrame(C.Structure):  # oracle/lavc_stub/ug_lavc_stub.h
    _fields_ = [("data", C.c_void_p * 8), ("linesize", C.c_int * 8), ("width", C.c_int), ("height", C.c_int), ("format", C.c_int),
                ("pts", C.c_int64), ("colorspace", C.c_int), ("color_range", C.c_int), ("opaque", C.c_void_p), ("stub_buf", C.c_void_p * 8)]


_ref = None


def ref():
    global _ref
    if _ref is None:
        if not os.path.exists(REF_LAVC):
            pytest.skip("oracle/_ref/libugref_lavc.so not built (make -C oracle ref_lavc where /root/reference exists)")
        r = C.CDLL(REF_LAVC)
        r.ug_stub_frame_new.restype = C.POINTER(StubAVFrame)
        r.ug_stub_frame_new.argtypes = [C.c_int, C.c_int, C.c_int]
        r.ug_stub_pixfmt_by_name.argtypes = [C.c_char_p]
        r.ug_stub_plane_rows.argtypes = [C.c_int, C.c_int, C.c_int]
        r.av_frame_free.argtypes = [C.POINTER(C.POINTER(StubAVFrame))]
        r.get_codec_from_name.argtypes = [C.c_char_p]
        r.vc_get_linesize.argtypes = [C.c_uint, C.c_int]
        r.to_lavc_vid_conv_init.restype = C.c_void_p
        r.to_lavc_vid_conv_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        r.to_lavc_vid_conv.restype = C.POINTER(StubAVFrame)
        r.to_lavc_vid_conv.argtypes = [C.c_void_p, C.c_void_p]
        r.to_lavc_vid_conv_destroy.argtypes = [C.POINTER(C.c_void_p)]
        r.get_av_to_uv_conversion.restype = C.c_void_p
        r.get_av_to_uv_conversion.argtypes = [C.c_int, C.c_int]
        r.av_to_uv_convert.restype = None
        r.av_to_uv_convert.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(StubAVFrame), C.c_int, C.POINTER(C.c_int)]
        r.av_to_uv_conversion_destroy.argtypes = [C.POINTER(C.c_void_p)]
        r.get_color_coeffs.restype = C.c_void_p
        r.get_color_coeffs.argtypes = [C.c_int, C.c_int]
        _ref = r
    return _ref


def plane_arrays(r, fr, h, extra_row=False):
    """numpy views (rows, linesize) of a stub frame's planes; extra_row: with the spare line the stub allocates behind each plane (some
    reference converters read whole groups past the end of the last line)"""
    out = []
    for i in range(4):
        if not fr.data[i]:
            break
        rows = r.ug_stub_plane_rows(fr.format, i, h) + (1 if extra_row else 0)
        out.append(np.ctypeslib.as_array(C.cast(fr.data[i], C.POINTER(C.c_uint8)), shape=(rows, fr.linesize[i])))
    return out


def ref_uv_to_av(uv, av, src, w, h):
    r = ref()
    st = r.to_lavc_vid_conv_init(r.get_codec_from_name(uv.encode()), w, h, r.ug_stub_pixfmt_by_name(av.encode()), 1)
    assert st, (uv, av, "the reference has no such conversion")
    fr = r.to_lavc_vid_conv(st, src.ctypes.data).contents
    planes = [p.copy() for p in plane_arrays(r, fr, h)]
    stp = C.c_void_p(st)
    r.to_lavc_vid_conv_destroy(C.byref(stp))
    return planes


def depth_of(av):
    for d in (10, 12, 16):
        if f"{d}le" in av:
            return d
    return 10 if av in ("p010le", "p210le") else 8


def make_frame(r, av, w, h, seed, colorspace, color_range):
    """a stub frame filled with random samples of the format's depth (P010 / P210: in the high bits)"""
    frp = r.ug_stub_frame_new(r.ug_stub_pixfmt_by_name(av.encode()), w, h)
    assert frp, av
    fr = frp.contents
    fr.colorspace, fr.color_range = colorspace, color_range
    rng = np.random.default_rng(seed)
    d = depth_of(av)
    for p in plane_arrays(r, fr, h):
        if d == 8 or av in ("xv30le", "y210le", "y212le", "ayuv64le", "rgb48le"):
            p[:] = rng.integers(0, 256, p.shape)
        else:
            v = rng.integers(0, 1 << d, (p.shape[0], p.shape[1] // 2)).astype("<u2")
            if av in ("p010le", "p210le"):
                v = v << 6
            p[:] = v.view(np.uint8)
    return frp


def ref_av_to_uv(frp, av, uv, w, h, pitch, shifts):
    r = ref()
    conv = r.get_av_to_uv_conversion(r.ug_stub_pixfmt_by_name(av.encode()), r.get_codec_from_name(uv.encode()))
    assert conv, (av, uv, "the reference has no such conversion")
    dst = np.zeros(pitch * h + 64, np.uint8)
    sh = (C.c_int * 3)(*shifts)
    r.av_to_uv_convert(conv, dst.ctypes.data, frp, pitch, sh)
    cp = C.c_void_p(conv)
    r.av_to_uv_conversion_destroy(C.byref(cp))
    return dst[: pitch * h].reshape(h, pitch)


def test_color_coefficient_tables():
    """the product's constant tables == get_color_coeffs of the compiled reference (no GPU needed: host function)"""
    from oracle import pyoracle as O
    from ultragrid_amd import lib
    r = ref()
    for cs in (1, 2):
        for depth in (0, 8, 10, 12, 16):
            want = O._RefCoeffs.from_address(r.get_color_coeffs(cs, depth))
            want = [getattr(want, n) for n, _ in O._RefCoeffs._fields_]
            got = (C.c_int * 14)()
            assert lib.load().ug_hip_color_coeffs(cs, depth, got) == 0
            assert list(got) == want, (cs, depth)


def test_compute_color_coeffs():
    """ug_hip_compute_color_coeffs == compute_color_coeffs of the compiled reference (color_space.c:193-197), incl. the BT.601 / 709 / 2020 weights"""
    from oracle import pyoracle as O
    from ultragrid_amd import lib
    r = O.ref()
    r.compute_color_coeffs.restype = O._RefCoeffs
    r.compute_color_coeffs.argtypes = [C.c_double, C.c_double, C.c_int]
    for kr, kb in [(0.299, 0.114), (0.2126, 0.0722), (0.2627, 0.0593), (0.212, 0.087), (0.3, 0.11)]:
        for depth in (0, 8, 10, 12, 16):
            want = r.compute_color_coeffs(kr, kb, depth)
            want = [getattr(want, n) for n, _ in O._RefCoeffs._fields_]
            got = (C.c_int * 14)()
            assert lib.load().ug_hip_compute_color_coeffs(kr, kb, depth, got) == 0
            assert list(got) == want, (kr, kb, depth)


def test_reference_has_every_row_we_claim():
    r = ref()
    for uv, av in TO_AV:
        st = r.to_lavc_vid_conv_init(r.get_codec_from_name(uv.encode()), 48, 8, r.ug_stub_pixfmt_by_name(av.encode()), 1)
        assert st, (uv, av)
        stp = C.c_void_p(st)
        r.to_lavc_vid_conv_destroy(C.byref(stp))
    for av, uv in FROM_AV:
        conv = r.get_av_to_uv_conversion(r.ug_stub_pixfmt_by_name(av.encode()), r.get_codec_from_name(uv.encode()))
        assert conv, (av, uv)
        cp = C.c_void_p(conv)
        r.av_to_uv_conversion_destroy(C.byref(cp))


def test_conversion_tables_are_covered():
    """every row of the reference's two conversion tables (read from its source here, where /root/reference exists) is either
    supported by the library or on the short list of rows deliberately left out"""
    import re
    from ultragrid_amd import lib
    base = "/root/reference/src/libavcodec/"
    if not os.path.exists(base):
        pytest.skip("no /root/reference here")
    L = lib.load()

    def name(f):
        return {"XV30": "xv30le", "Y210": "y210le", "Y212": "y212le", "AYUV64": "ayuv64le", "AYUV64LE": "ayuv64le"}.get(f, f.lower())
    rows = re.findall(r"\{ *(\w+), *AV_PIX_FMT_(\w+), *(\w+) *\}", open(base + "to_lavc_vid_conv.c").read())
    rows = [r for r in rows if r[0] != "VIDEO_CODEC_NONE"]
    missing = {(c, f) for c, f, _ in rows if not L.ug_hip_uv_to_av_supported(c.encode(), name(f).encode())}
    assert len(rows) == 57 and not missing, missing
    rows = re.findall(r"\{ *AV_PIX_FMT_(\w+), *(\w+),\s*(\w+),\s*(\w+) *\}", open(base + "from_lavc_vid_conv.c").read())
    missing = {(f, c) for f, c, _, _ in rows if not L.ug_hip_av_to_uv_supported(name(f).encode(), c.encode())}
    assert len(rows) == 105, len(rows)
    assert missing == {("Y212", "Y216"), ("Y210", "Y216"), ("AYUV64", "UYVY"), ("VDPAU", "HW_VDPAU"), ("DRM_PRIME", "DRM_PRIME")}, missing


@pytest.mark.gpu
@pytest.mark.parametrize("uv,av", TO_AV)
def test_gpu_uv_to_av(hip, uv, av):
    import torch
    r = ref()
    assert hip.L.load().ug_hip_uv_to_av_supported(uv.encode(), av.encode()) == 1
    sizes = [(48, 8), (96, 4), (1920, 2)]
    if (uv, av) not in FORWARDED_TO:
        sizes += [(54, 6)] + ([(49, 5), (7, 3)] if (uv, av) != ("v210", "yuv420p10le") else [(50, 6)])
    if (uv, av) == ("v210", "p010le"):
        sizes += [(50, 7), (1280, 10), (2048, 5), (54, 5), (7, 6)]  # to_planar.c:80-94,139-150: odd last line, width % 6 margin
    for i, (w, h) in enumerate(sizes):
        ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        src = np.random.default_rng(i).integers(0, 256, ls * h + 64).astype(np.uint8)
        want = ref_uv_to_av(uv, av, src, w, h)
        planes = [torch.zeros(p.shape, dtype=torch.uint8, device="cuda") for p in want]
        hip.uv_to_av(uv, av, torch.from_numpy(src).cuda(), w, h, planes)
        torch.cuda.synchronize()
        if (uv, av) == ("Y216", "p010le"):
            # y216_to_p010le finds the odd luma line by running on from the even one (to_planar.c:191-199), so with an AVFrame whose
            # linesize is padded it writes that line into the even line's padding and leaves the odd line unwritten.  The product puts
            # it where the frame says; compare with the restatement (pinned to the reference on tight planes, tests/test_planar_api.py)
            from oracle import planar_oracle as PO
            ty, tc = PO.to_planar("y216_to_p010le", src, w, h)
            assert np.array_equal(planes[0].cpu().numpy()[:, : 2 * w].view(np.uint16), ty)
            assert np.array_equal(planes[1].cpu().numpy()[:, : tc.shape[1] * 2].view(np.uint16), tc)
            continue
        for k, (p, wnt) in enumerate(zip(planes, want)):
            assert np.array_equal(p.cpu().numpy(), wnt), (uv, av, w, h, k)


@pytest.mark.gpu
@pytest.mark.parametrize("av,uv", FROM_AV)
def test_gpu_av_to_uv(hip, av, uv):
    import torch
    r = ref()
    assert hip.L.load().ug_hip_av_to_uv_supported(av.encode(), uv.encode()) == 1
    cases = [((48, 8), 1, 1, (0, 8, 16)), ((96, 4), 5, 2, (16, 8, 0)), ((54, 6), 2, 1, (8, 16, 0)), ((50, 7), 6, 1, (0, 8, 16)), ((1920, 2), 1, 2, (0, 8, 16))]
    for i, ((w, h), cs, rng_, shifts) in enumerate(cases):
        if av.startswith("yuvj"):
            rng_ = 2
        frp = make_frame(r, av, w, h, 10 + i, cs, rng_)
        fr = frp.contents
        pitch = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        want = ref_av_to_uv(frp, av, uv, w, h, pitch, shifts)
        planes = [torch.from_numpy(p.copy()).cuda() for p in plane_arrays(r, fr, h, extra_row=True)]
        dst = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
        hip.av_to_uv(av, uv, planes, w, h, dst, pitch, shifts, colorspace=cs, color_range=rng_)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        r.av_frame_free(C.byref(frp))
        if uv == "R12L" and w % 8:
            # the reference packs uninitialised stack behind a ragged line end (from_planar.c:74-82): compare the bits of real pixels
            nbits = 36 * w
            full, rem = nbits // 8, nbits % 8
            assert np.array_equal(got[:, :full], want[:, :full]), (av, uv, w, h)
            if rem:
                mask = (1 << rem) - 1
                assert np.array_equal(got[:, full] & mask, want[:, full] & mask), (av, uv, w, h)
            continue
        assert np.array_equal(got, want), (av, uv, w, h, cs, rng_, np.argwhere(got != want)[:4])


@pytest.mark.gpu
def test_gpu_lavc_conv_errors(hip):
    import torch
    lib = hip.L.load()
    assert lib.ug_hip_uv_to_av_supported(b"UYVY", b"no_such_format") == 0
    assert lib.ug_hip_av_to_uv_supported(b"yuv420p", b"DXT1") == 0
    f = hip.AvFrame()
    assert lib.ug_hip_uv_to_av(b"UYVY", b"bogus", None, C.byref(f), None) == hip.L.EUNSUPP
    src = torch.zeros(4 * 48 * 8, dtype=torch.uint8, device="cuda")
    assert lib.ug_hip_uv_to_av(b"UYVY", b"yuv444p", src.data_ptr(), C.byref(f), None) == hip.L.EINVAL  # no planes


@pytest.mark.gpu
def test_gpu_v210_planar_round_trip_4k(hip):
    """v210 -> yuv422p10le (to_lavc) -> v210 (from_lavc) is the identity on the sample bits at 3840x2160 (full-size property)"""
    import torch
    w, h = 3840, 2160
    ls = w // 6 * 16
    g = torch.Generator(device="cuda").manual_seed(3)
    src = (torch.randint(0, 1 << 30, (h, ls // 4), generator=g, device="cuda", dtype=torch.int64) & 0x3FFFFFFF).to(torch.int32)
    planes = [torch.zeros((h, 2 * w), dtype=torch.uint8, device="cuda"), torch.zeros((h, w), dtype=torch.uint8, device="cuda"), torch.zeros((h, w), dtype=torch.uint8, device="cuda")]
    hip.uv_to_av("v210", "yuv422p10le", src, w, h, planes)
    back = torch.zeros((h, ls), dtype=torch.uint8, device="cuda")
    hip.av_to_uv("yuv422p10le", "v210", planes, w, h, back, ls)
    torch.cuda.synchronize()
    assert torch.equal(back.view(torch.int32), src)


# ---- the same comparisons against committed vectors (tests/golden/lavc_ref.npz, made by tests/golden/make_lavc_golden.py from the compiled
# ---- reference): these run on a box that has no oracle/_ref
def _golden():
    return np.load(os.path.join(HERE, "golden", "lavc_ref.npz"))


@pytest.mark.gpu
def test_gpu_uv_to_av_vs_committed_vectors(hip):
    import torch
    g = _golden()
    keys = sorted({k.rsplit("|", 1)[0] for k in g.files if k.startswith("to|")})
    assert len(keys) >= 90
    for key in keys:
        _, uv, av, dims = key.split("|")
        if (uv, av) == ("Y216", "p010le"):
            continue  # the reference's own output is wrong for padded line sizes (see test_gpu_uv_to_av)
        w, h = map(int, dims.split("x"))
        want = [g[f"{key}|p{k}"] for k in range(4) if f"{key}|p{k}" in g.files]
        planes = [torch.zeros(p.shape, dtype=torch.uint8, device="cuda") for p in want]
        hip.uv_to_av(uv, av, torch.from_numpy(g[key + "|in"]).cuda(), w, h, planes)
        torch.cuda.synchronize()
        for k, (p, wnt) in enumerate(zip(planes, want)):
            assert np.array_equal(p.cpu().numpy(), wnt), (key, k)


@pytest.mark.gpu
def test_gpu_av_to_uv_vs_committed_vectors(hip):
    import torch
    g = _golden()
    keys = sorted({k.rsplit("|", 1)[0] for k in g.files if k.startswith("from|")})
    assert len(keys) >= 170
    for key in keys:
        _, av, uv, dims, cs, rng_ = key.split("|")
        w, h = map(int, dims.split("x"))
        want = g[key + "|out"]
        planes = [torch.from_numpy(g[f"{key}|p{k}"]).cuda() for k in range(4) if f"{key}|p{k}" in g.files]
        pitch = want.shape[1]
        dst = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
        hip.av_to_uv(av, uv, planes, w, h, dst, pitch, (0, 8, 16), colorspace=int(cs), color_range=int(rng_))
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        if uv == "R12L" and w % 8:
            full = 36 * w // 8
            assert np.array_equal(got[:, :full], want[:, :full]), key
            continue
        assert np.array_equal(got, want), key


def _random_sizes(tag: str, n: int = 5):
    rng = np.random.default_rng(abs(hash(tag)) % (1 << 31) if False else sum(map(ord, tag)))
    # widths from 16: the SSE loop of the reference's yuv420p_to_uyvy runs `x < width - 15` on an unsigned width and crashes below that
    return [(int(rng.integers(16, 131)), int(rng.integers(1, 10))) for _ in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("uv,av", [p for p in TO_AV if p not in FORWARDED_TO])
def test_gpu_uv_to_av_random_sizes(hip, uv, av):
    """ragged frames: five pseudo-random sizes per conversion (16..130 x 1..9), whole planes against the compiled reference"""
    import torch
    r = ref()
    for (w, h) in _random_sizes(uv + av):
        if (uv, av) == ("v210", "yuv420p10le") and h % 2:
            h += 1  # the reference reads and writes one line past an odd-height picture (to_lavc_vid_conv.c:204-208)
        ls = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        src = np.random.default_rng(w * 31 + h).integers(0, 256, ls * (h + 1) + 64).astype(np.uint8)
        want = ref_uv_to_av(uv, av, src, w, h)
        planes = [torch.zeros(p.shape, dtype=torch.uint8, device="cuda") for p in want]
        hip.uv_to_av(uv, av, torch.from_numpy(src).cuda(), w, h, planes)
        torch.cuda.synchronize()
        for k, (p, wnt) in enumerate(zip(planes, want)):
            assert np.array_equal(p.cpu().numpy(), wnt), (uv, av, w, h, k)


@pytest.mark.gpu
@pytest.mark.parametrize("av,uv", FROM_AV)
def test_gpu_av_to_uv_random_sizes(hip, av, uv):
    import torch
    r = ref()
    for i, (w, h) in enumerate(_random_sizes(av + uv)):
        cs, rng_ = (1, 1) if i % 2 else (6, 2)
        frp = make_frame(r, av, w, h, 90 + i, cs, rng_)
        pitch = r.vc_get_linesize(w, r.get_codec_from_name(uv.encode()))
        want = ref_av_to_uv(frp, av, uv, w, h, pitch, (0, 8, 16))
        planes = [torch.from_numpy(p.copy()).cuda() for p in plane_arrays(r, frp.contents, h, extra_row=True)]
        dst = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
        hip.av_to_uv(av, uv, planes, w, h, dst, pitch, (0, 8, 16), colorspace=cs, color_range=rng_)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        r.av_frame_free(C.byref(frp))
        if uv == "R12L" and w % 8:
            full = 36 * w // 8
            assert np.array_equal(got[:, :full], want[:, :full]), (av, uv, w, h)
            continue
        assert np.array_equal(got, want), (av, uv, w, h, np.argwhere(got != want)[:4])
