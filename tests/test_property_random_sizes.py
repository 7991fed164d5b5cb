"""Randomised-geometry parity (hypothesis): arbitrary widths/heights, all conversion pairs.
CPU: restatement vs the compiled reference (when oracle/_ref is present).  GPU: HIP kernels vs the oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from ultragrid_amd import synth

PAIRS = [("v210", "UYVY"), ("YUYV", "UYVY"), ("UYVY", "YUYV"), ("UYVY", "RGB"), ("UYVY", "RGBA"), ("RGB", "UYVY"), ("BGR", "UYVY"),
         ("RGBA", "UYVY"), ("RG48", "UYVY"), ("v210", "RGB"), ("v210", "RG48"), ("RGBA", "RGB"), ("RGB", "RGBA"), ("RGBA", "RGBA"),
         ("RGB", "RGB"), ("BGR", "RGB"), ("UYVY", "v210")]
SHIFTS = [(0, 8, 16), (16, 8, 0), (8, 16, 24), (24, 16, 8)]
common = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(max_examples=120, **common)
@given(pair=st.sampled_from(PAIRS), w=st.integers(1, 260), h=st.integers(1, 9), sh=st.sampled_from(SHIFTS), seed=st.integers(0, 2 ** 16))
def test_cpu_restatement_vs_compiled_reference(po, pair, w, h, sh, seed):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    i, o = pair
    src = synth.s1_random(i, w, h, salt=seed)
    assert np.array_equal(po.convert_frame(i, o, src, w, h, sh), po.ref_convert_frame(i, o, src, w, h, sh, scalar=True)), (pair, w, h, sh)


@settings(max_examples=60, **common)
@given(w=st.integers(1, 300), h=st.integers(1, 40), seed=st.integers(0, 2 ** 16))
def test_cpu_uyvy_to_i420_vs_compiled_reference(po, w, h, seed):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    src = synth.s1_random("UYVY", w, h, salt=seed)
    for a, b in zip(po.uyvy_to_i420(src, w, h), po.uyvy_to_i420(src, w, h, use_ref=True)):
        assert np.array_equal(a, b), (w, h)


@pytest.mark.gpu
@settings(max_examples=80, **common)
@given(pair=st.sampled_from(PAIRS), w=st.integers(1, 400), h=st.integers(1, 12), sh=st.sampled_from(SHIFTS), seed=st.integers(0, 2 ** 16))
def test_gpu_pixfmt_random_geometry(hip, po, pair, w, h, sh, seed):
    import torch
    from ultragrid_amd import lib as L
    i, o = pair
    src = synth.s1_random(i, w, h, salt=seed)
    dev = torch.from_numpy(np.concatenate([src, np.zeros(64, np.uint8)])).cuda()
    got = hip.pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], dev, w, h, sh).cpu().numpy()
    assert np.array_equal(got, po.convert_frame(i, o, src, w, h, sh)), (pair, w, h, sh)


@pytest.mark.gpu
@settings(max_examples=60, **common)
@given(fmt=st.sampled_from(["RGB", "RGBA", "UYVY", "v210"]), out=st.sampled_from(["dxt1", "dxt5"]), bw=st.integers(1, 70), bh=st.integers(1, 12),
       mirror=st.booleans(), kind=st.sampled_from(["S1", "S2", "S4"]), seed=st.integers(0, 2 ** 16))
def test_gpu_dxt_random_geometry(hip, po, fmt, out, bw, bh, mirror, kind, seed):
    import torch
    from ultragrid_amd import lib as L
    w, h = (12 * bw if fmt == "v210" else 4 * bw), 4 * bh
    src = synth.frame(kind, fmt, w, h, seed)
    pin = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY, "v210": po.IN_V210}[fmt]
    oid_p, oid_l = (po.OUT_DXT1, L.DXT1) if out == "dxt1" else (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)
    hh = -h if mirror else h
    got = hip.dxt_encode(L.PF_NAMES[fmt], oid_l, torch.from_numpy(src).cuda(), w, hh).cpu().numpy()
    assert np.array_equal(got, po.dxt_encode(pin, oid_p, src, w, hh)), (fmt, out, w, h, mirror, kind)
    dec = hip.dxt_decode(oid_l, L.PF_RGBA, torch.from_numpy(got).cuda(), w, h).cpu().numpy()
    assert np.array_equal(dec, po.dxt_decode(oid_p, "RGBA", got, w, h))


@pytest.mark.gpu
@settings(max_examples=25, **common)
@given(w=st.integers(1, 200).map(lambda x: 2 * x), h=st.integers(1, 120), q=st.sampled_from([35, 75, 95]), seed=st.integers(0, 2 ** 16))
def test_gpu_jpeg_coeffs_random_geometry(hip, po, w, h, q, seed):
    import torch
    src = synth.s1_random("UYVY", w, h, salt=seed)
    y, u, v = po.uyvy_to_i420(src, w, h)
    mw, mh = (w + 15) // 16, (h + 15) // 16
    dl, dc = po.jpeg_divisors(po.jpeg_qtable(q, 0)), po.jpeg_divisors(po.jpeg_qtable(q, 1))
    want = (po.jpeg_fdct_quant_plane(y, dl, 2 * mw, 2 * mh), po.jpeg_fdct_quant_plane(u, dc, mw, mh), po.jpeg_fdct_quant_plane(v, dc, mw, mh))
    got = hip.uyvy_to_jpeg420_coeffs(torch.from_numpy(src).cuda(), w, h, hip.jpeg_divisors_device(q, "cuda"))
    for g, wn in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), wn), (w, h, q)
