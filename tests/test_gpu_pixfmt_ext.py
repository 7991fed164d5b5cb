"""The pairs of decoders[] (src/pixfmt_conv.c:3041-3103) outside the v210 / UYVY / RGB / RGBA core (csrc/pixfmt_ext.hip): every one
against the reference's own line converter, obtained from get_decoder_from_to() of the compiled reference (oracle/_ref/libugref.so,
which travels to the GPU box) and run line by line on the box's CPU exactly as tools/convert.cpp:43-48 does.  Whole output buffers
are compared, padding included."""
import ctypes as C

import numpy as np
import pytest

PAIRS = [("DVS10", "v210"), ("R10k", "RGBA"), ("R10k", "RG48"), ("R10k", "Y416"), ("R10k", "RGB"), ("R10k", "UYVY"),
         ("R12L", "RGBA"), ("R12L", "RGB"), ("R12L", "RG48"), ("R12L", "R10k"), ("R12L", "Y416"), ("R12L", "UYVY"),
         ("RGBA", "R12L"), ("RGB", "R12L"), ("RGBA", "RG48"), ("RGB", "RG48"), ("UYVY", "RG48"),
         ("RG48", "R12L"), ("RG48", "R10k"), ("RG48", "RGB"), ("RG48", "RGBA"), ("RG48", "v210"), ("RG48", "Y216"), ("RG48", "Y416"),
         ("Y416", "RG48"), ("RGBA", "VUYA"), ("YUYV", "RGB"), ("RGBA", "R10k"), ("UYVY", "Y216"), ("UYVY", "Y416"),
         ("VUYA", "Y416"), ("VUYA", "UYVY"), ("VUYA", "RGB"), ("Y216", "UYVY"), ("Y216", "v210"), ("Y416", "UYVY"), ("Y416", "v210"),
         ("Y416", "R12L"), ("Y416", "R10k"), ("Y416", "RGB"), ("Y416", "RGBA"), ("v210", "Y216"), ("v210", "Y416")]
SIZES = [(48, 4), (50, 3), (127, 5), (96, 2), (1920, 2), (7, 3), (2, 1), (1936, 3), (4100, 2)]   # aligned lines take the 128-bit kernels, the others do not
FILL = 0x5A   # what the destination holds beforehand: bytes a converter leaves alone must still hold it, on both sides
DEC = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)


def aligned(n, fill=None, rng=None):
    buf = np.zeros(n + 128, np.uint8)
    off = (-buf.ctypes.data) % 64
    v = buf[off: off + n]
    if rng is not None:
        v[:] = rng.integers(0, 256, n)
    return v


def ref_frame(po, i, o, src, w, h, sh):
    r = po.ref()
    r.get_codec_from_name.argtypes = [C.c_char_p]
    ci, co = r.get_codec_from_name(i.encode()), r.get_codec_from_name(o.encode())
    fn = r.get_decoder_from_to(ci, co)
    assert fn, (i, o)
    sls, dls, dsz = r.vc_get_linesize(w, ci), r.vc_get_linesize(w, co), r.vc_get_size(w, co)
    out = aligned(dls * h + 64)
    out[:] = FILL
    for y in range(h):
        DEC(fn)(out.ctypes.data + y * dls, src.ctypes.data + y * sls, dsz, *sh)
    return out[: dls * h], sls, dls


def run_pair(hip, po, i, o, sizes=SIZES):
    import torch
    L = hip.L
    assert L.load().ug_hip_pixfmt_supported(L.PF_NAMES[i], L.PF_NAMES[o]) == 1
    for k, (w, h) in enumerate(sizes):
        for sh in [(0, 8, 16), (16, 8, 0)]:
            r = po.ref()
            r.get_codec_from_name.argtypes = [C.c_char_p]
            sls = r.vc_get_linesize(w, r.get_codec_from_name(i.encode()))
            src = aligned(sls * h + 64, rng=np.random.default_rng(100 * k + sh[0]))
            want, sls, dls = ref_frame(po, i, o, src, w, h, sh)
            dsrc = torch.from_numpy(src.copy()).cuda()
            ddst = torch.full((dls * h,), FILL, dtype=torch.uint8, device="cuda")
            rc = L.load().ug_hip_pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], dsrc.data_ptr(), ddst.data_ptr(), w, h, 0, 0, *sh, None)
            assert rc == 0, (i, o, L.last_error() if hasattr(L, "last_error") else rc)
            torch.cuda.synchronize()
            got = ddst.cpu().numpy()
            if not np.array_equal(got, want):
                bad = np.flatnonzero(got != want)
                raise AssertionError((i, o, w, h, sh, "first mismatching byte offsets in line:", sorted(set((bad % dls).tolist()))[:24]))


@pytest.mark.gpu
@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_ext_pair_vs_compiled_reference(hip, po, pair):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    run_pair(hip, po, *pair)


@pytest.mark.gpu
def test_dvs10_to_uyvy(hip, po):
    """vc_copylineDVS10 (pixfmt_conv.c:690-721) reads and writes 64-bit words: line sizes that keep them aligned"""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    run_pair(hip, po, "DVS10", "UYVY", sizes=[(48, 4), (96, 2), (1920, 2)])


def test_every_decoder_pair_is_supported(po):
    """decoders[] read from the reference source (CPU container): every pair is answered by ug_hip_pixfmt_supported"""
    import os
    import re
    from ultragrid_amd import lib
    path = "/root/reference/src/pixfmt_conv.c"
    if not os.path.exists(path):
        pytest.skip("no /root/reference here")
    txt = open(path).read()
    tab = txt[txt.index("static const struct decoder_item decoders[]"):]
    tab = tab[: tab.index("};")]
    rows = re.findall(r"\{ *(\w+), *(\w+), *(\w+) *\}", tab)
    assert len(rows) == 61
    L = lib.load()
    missing = [(i, o) for _, i, o in rows if not L.ug_hip_pixfmt_supported(lib.PF_NAMES[i], lib.PF_NAMES[o])]
    assert not missing, missing


@pytest.mark.gpu
def test_ext_pairs_vs_committed_vectors(hip):
    """tests/golden/pixfmt_ext_ref.npz (tests/golden/make_lavc_golden.py, from the compiled reference): needs no oracle/_ref"""
    import os
    import torch
    L = hip.L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pixfmt_ext_ref.npz"))
    keys = sorted({k.rsplit("|", 1)[0] for k in g.files})
    assert len(keys) >= 85
    for key in keys:
        i, o, dims = key.split("|")
        w, h = map(int, dims.split("x"))
        want = g[key + "|out"]
        dsrc = torch.from_numpy(g[key + "|in"]).cuda()
        ddst = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
        assert L.load().ug_hip_pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], dsrc.data_ptr(), ddst.data_ptr(), w, h, 0, 0, 0, 8, 16, None) == 0, key
        torch.cuda.synchronize()
        assert np.array_equal(ddst.cpu().numpy(), want), key


@pytest.mark.gpu
@pytest.mark.parametrize("func,bpp_in,bpp_out", [("vc_copylineUYVYtoGrayscale", 2, 1), ("vc_copylineABGRtoRGB", 4, 3), ("vc_copylineBGRAtoRGB", 4, 3),
                                                 ("vc_copylineToRGBA_inplace", 4, 4)])
def test_exported_line_functions(hip, po, func, bpp_in, bpp_out):
    """the converters pixfmt_conv.h exports outside decoders[], against the same symbols of the compiled reference -- its portable build
    (no -msse4.1): the SSSE3 branch of vc_copylineABGRtoRGB never advances `src` in its tail loop (pixfmt_conv.c:828-834) and repeats one
    pixel over the last 4-7 of a line, the same slip as vc_copylineRGBAtoRGB's (oracle/Makefile)"""
    import torch
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    fn = DEC((func, po.ref(scalar=True)))
    for (w, h) in [(48, 4), (50, 3), (7, 2), (1920, 2)]:
        sp = (w * bpp_in + 3) // 4 * 4 + (64 if bpp_in == 2 and w % 2 else 0)
        dp, L_ = w * bpp_out + 8, w * bpp_out
        src = aligned(sp * h + 64, rng=np.random.default_rng(w))
        want = aligned(dp * h + 64)
        want[:] = 0x5A
        for y in range(h):
            fn(want.ctypes.data + y * dp, src.ctypes.data + y * sp, L_, 16, 0, 8)
        dsrc = torch.from_numpy(src.copy()).cuda()
        ddst = torch.full((dp * h + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        rc = hip.L.load().ug_hip_pixfmt_line_func(func.encode(), dsrc.data_ptr(), ddst.data_ptr(), w, h, sp, dp, L_, 16, 0, 8, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(ddst.cpu().numpy(), want), (func, w, h)


def test_best_decoder_choice_equals_the_reference(po):
    """ug_hip_pixfmt_best == get_best_decoder_from (pixfmt_conv.c:3126-3172) for every source codec and 400 random candidate sets"""
    from ultragrid_amd import lib
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    r = po.ref()
    r.get_codec_from_name.argtypes = [C.c_char_p]
    r.get_best_decoder_from.restype = C.c_void_p
    r.get_best_decoder_from.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    names = ["RGBA", "UYVY", "YUYV", "VUYA", "R10k", "R12L", "v210", "DVS10", "RGB", "BGR", "RG48", "Y216", "Y416"]
    ref_id = {n: r.get_codec_from_name(n.encode()) for n in names}
    back = {v: k for k, v in ref_id.items()}
    rng = np.random.default_rng(1)
    L = lib.load()
    for src in names:
        for _ in range(400 // len(names) + 1):
            cand = list(rng.choice(names, size=rng.integers(1, 6), replace=False))
            rc = (C.c_int * (len(cand) + 1))(*[ref_id[c] for c in cand], 0)
            rout = C.c_int(0)
            fn = r.get_best_decoder_from(ref_id[src], rc, C.byref(rout))
            mc = (C.c_int * (len(cand) + 1))(*[lib.PF_NAMES[c] for c in cand], 0)
            mout = C.c_int(0)
            rv = L.ug_hip_pixfmt_best(lib.PF_NAMES[src], mc, C.byref(mout))
            if not fn:
                assert rv == lib.EUNSUPP, (src, cand)
            else:
                assert rv == 0 and mout.value == lib.PF_NAMES[back[rout.value]], (src, cand, back[rout.value])


@pytest.mark.gpu
@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_ext_pair_random_sizes(hip, po, pair):
    """six pseudo-random frame sizes per pair (1..130 x 1..6)"""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(sum(map(ord, pair[0] + pair[1])))
    run_pair(hip, po, *pair, sizes=[(int(rng.integers(1, 131)), int(rng.integers(1, 7))) for _ in range(6)])
