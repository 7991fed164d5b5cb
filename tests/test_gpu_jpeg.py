"""GPU parity: HIP FDCT+quantise vs oracle/jpeg_oracle.c -- identical fp32 coefficients (0 ULP; the
north-star bound is 1 ULP) and identical int16 output."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_oracle.npz"))


def test_committed_golden(hip, po):
    import torch
    y, u, v = po.uyvy_to_i420(GOLD["in_uyvy"], 40, 24)
    for q in (50, 75, 90):
        div = hip.jpeg_divisors_device(q, "cuda")
        for comp, plane in ((0, y), (1, u)):
            out, coef = hip.jpeg_fdct_quant_plane(torch.from_numpy(plane).cuda(), div[64 * comp: 64 * comp + 64].contiguous(), want_coef=True)
            assert np.array_equal(out.cpu().numpy(), GOLD[f"q{q}_c{comp}_out"])
            assert np.array_equal(coef.cpu().numpy().view(np.uint32), GOLD[f"q{q}_c{comp}_coef"].view(np.uint32))


@pytest.mark.parametrize("shape", [(8, 8), (64, 96), (61, 93), (1080, 1920), (2160, 3840)], ids=str)
def test_plane_bit_exact(hip, po, shape):
    import torch
    h, w = shape
    rng = np.random.default_rng(h * w)
    for kind in ("rand", "flat127", "smooth"):
        if kind == "rand":
            plane = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == "flat127":
            plane = np.full((h, w), 127, np.uint8)  # test/gpujpeg_test.cpp:78
        else:
            plane = synth.s2_video("UYVY", w + (w & 1), h).reshape(h, -1, 4)[..., 1::2].reshape(h, -1)[:, :w].copy()
        for q in (75, 35):
            div = po.jpeg_divisors(po.jpeg_qtable(q, 0))
            want, wcoef = po.jpeg_fdct_quant_plane(plane, div, want_coef=True)
            got, gcoef = hip.jpeg_fdct_quant_plane(torch.from_numpy(plane).cuda(), torch.from_numpy(div).cuda(), want_coef=True)
            # float DCT: ULP distance between HIP and the fp32 CPU restatement (bound 1, measured 0)
            a, b = gcoef.cpu().numpy().view(np.int32).astype(np.int64), wcoef.view(np.int32).astype(np.int64)
            ulp = np.abs(np.where(a < 0, -(a & 0x7FFFFFFF), a) - np.where(b < 0, -(b & 0x7FFFFFFF), b))
            assert ulp.max() <= 1, (kind, q, ulp.max())
            assert ulp.max() == 0
            assert np.array_equal(got.cpu().numpy(), want), (kind, q)
        if h * w > 10 ** 6:
            break


@pytest.mark.parametrize("dims", [(16, 16), (40, 24), (50, 37), (1920, 1080), (3840, 2160)], ids=str)
def test_fused_uyvy_420_pipeline(hip, po, dims):
    """BASELINE.json configs[3]: UYVY -> planar 4:2:0 (uyvy_to_i420 rounding) -> FDCT+quant, fused on the GPU,
    vs the reference's uyvy_to_i420 (oracle/_ref when present, else its restatement) + the FDCT oracle."""
    import torch
    w, h = dims
    src = synth.s2_video("UYVY", w, h) if w % 2 == 0 else synth.s1_random("UYVY", w, h)
    y, u, v = po.uyvy_to_i420(src, w, h, use_ref=po.have_ref())
    mw, mh = (w + 15) // 16, (h + 15) // 16
    q = 75
    dl, dc = po.jpeg_divisors(po.jpeg_qtable(q, 0)), po.jpeg_divisors(po.jpeg_qtable(q, 1))
    want = (po.jpeg_fdct_quant_plane(y, dl, 2 * mw, 2 * mh), po.jpeg_fdct_quant_plane(u, dc, mw, mh), po.jpeg_fdct_quant_plane(v, dc, mw, mh))
    got = hip.uyvy_to_jpeg420_coeffs(torch.from_numpy(src).cuda(), w, h, hip.jpeg_divisors_device(q, "cuda"))
    for g, wnt, name in zip(got, want, "Y Cb Cr".split()):
        assert np.array_equal(g.cpu().numpy(), wnt), name
    # unfused GPU path (uyvy_to_i420 kernel + per-plane FDCT) gives the same coefficients
    gy, gu, gv = hip.uyvy_to_i420(torch.from_numpy(src).cuda(), w, h)
    div = hip.jpeg_divisors_device(q, "cuda")
    assert torch.equal(hip.jpeg_fdct_quant_plane(gy, div[:64].contiguous(), 2 * mw, 2 * mh), got[0])
    assert torch.equal(hip.jpeg_fdct_quant_plane(gu, div[64:].contiguous(), mw, mh), got[1])
