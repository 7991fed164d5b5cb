"""CPU: the JPEG FDCT+quantise specification (oracle/jpeg_oracle.c).  PARITY UNPINNED against the
reference (the stage lives in the external, un-vendored libgpujpeg); checked against an fp64
scipy DCT, the T.81 tables, and committed regression outputs."""
import os

import numpy as np
import scipy.fft

from ultragrid_amd import synth

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_oracle.npz"))
AAN = np.array([1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379])


def _blocks(plane):
    h, w = plane.shape
    bh, bw = (h + 7) // 8, (w + 7) // 8
    p = np.pad(plane, ((0, bh * 8 - h), (0, bw * 8 - w)), mode="edge")
    return p.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3).reshape(-1, 8, 8)


def test_qtables(po):
    q50 = po.jpeg_qtable(50, 0)
    assert q50[:8].tolist() == [16, 11, 10, 16, 24, 40, 51, 61] and q50[63] == 99          # T.81 Table K.1
    assert po.jpeg_qtable(50, 1)[:4].tolist() == [17, 18, 24, 47]                          # Table K.2
    assert (po.jpeg_qtable(100, 0) == 1).all() and po.jpeg_qtable(1, 0).max() == 255
    assert po.jpeg_qtable(75, 0)[0] == 8   # (16*50+50)/100
    assert sorted(po.ZIGZAG.tolist()) == list(range(64)) and po.ZIGZAG[:6].tolist() == [0, 1, 8, 16, 9, 2]


def test_regression_vs_committed(po):
    y, u, v = po.uyvy_to_i420(GOLD["in_uyvy"], 40, 24)
    for q in (50, 75, 90):
        for comp, plane in ((0, y), (1, u)):
            div = po.jpeg_divisors(po.jpeg_qtable(q, comp))
            out, coef = po.jpeg_fdct_quant_plane(plane, div, want_coef=True)
            assert np.array_equal(out, GOLD[f"q{q}_c{comp}_out"])
            assert np.array_equal(coef.view(np.uint32), GOLD[f"q{q}_c{comp}_coef"].view(np.uint32))  # bit pattern
            assert np.array_equal(po.jpeg_qtable(q, comp), GOLD[f"q{q}_c{comp}_qtable"])


def test_against_fp64_dct(po):
    """(b) of the tolerance contract: vs scipy.fft.dctn (fp64, T.81 A.3.3 normalisation):
    AAN-scaled coefficients agree to fp32 round-off; quantised values differ by <= 1 step, on < 1e-3 of the
    coefficients, and only where the exact quotient is within 1e-3 of a rounding tie."""
    rng = np.random.default_rng(11)
    for kind in ("S1", "S2", "flat127", "edge"):
        if kind == "S1":
            plane = rng.integers(0, 256, (64, 96), dtype=np.uint8)
        elif kind == "S2":
            plane = synth.s2_video("UYVY", 96, 64).reshape(64, 48, 4)[..., 1::2].reshape(64, 96)
        elif kind == "flat127":
            plane = np.full((64, 96), 127, np.uint8)     # test/gpujpeg_test.cpp:78 fixture
        else:
            plane = rng.integers(0, 256, (61, 93), dtype=np.uint8)  # ragged: edge replication
        for q in (50, 90):
            qt = po.jpeg_qtable(q, 0)
            out, coef = po.jpeg_fdct_quant_plane(plane, po.jpeg_divisors(qt), want_coef=True)
            blk = _blocks(plane).astype(np.float64) - 128.0
            ref = scipy.fft.dctn(blk, type=2, norm="ortho", axes=(1, 2))          # == T.81 FDCT
            # scipy's orthonormal DCT-II is exactly the T.81 A.3.3 FDCT; AAN output = FDCT * 8*aan_u*aan_v
            aan_ref = ref * (AAN[:, None] * AAN[None, :]) * 8.0
            err = np.abs(coef.reshape(-1, 8, 8) - aan_ref)
            assert err.max() <= 2e-3, (kind, err.max())   # |coef| up to ~1.6e4 -> fp32 round-off of the butterfly
            want = np.rint(ref / qt.reshape(8, 8).astype(np.float64)).astype(np.int64)
            got = out.astype(np.int64)[:, np.argsort(po.ZIGZAG)].reshape(-1, 8, 8)  # undo zig-zag
            diff = np.abs(got - want)
            assert diff.max() <= 1, kind
            if kind != "flat127":  # flat 127: every DC is the exact tie -8/16 = -0.5
                assert (diff != 0).mean() < 1e-3, (kind, (diff != 0).mean())
            # every disagreement sits on a rounding tie of the exact quotient (multiply-by-reciprocal
            # breaks exact .5 ties, which are common for DC = sum/8 over integer samples)
            frac = np.abs(np.abs(ref / qt.reshape(8, 8)) % 1.0 - 0.5)
            assert (frac[diff != 0] < 1e-3).all(), kind
