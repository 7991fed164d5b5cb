"""GPU parity: HIP pixel-format kernels vs the reference's compiled C (oracle/_ref, when it travelled),
the committed fixtures generated from it, and the restatement -- bit-exact (integer work)."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixfmt_ref.npz"))
PAIRS = sorted({tuple(k.split("_")[1:3]) for k in GOLD.files if k.startswith("in_")})


def _conv(hip, i, o, src, w, h, sh=(0, 8, 16)):
    import torch
    from ultragrid_amd import lib as L
    # the reference's decoders may over-read the source by up to MAX_PADDING (pixfmt_conv.h:70-74)
    dev = torch.from_numpy(np.concatenate([src, np.zeros(64, np.uint8)])).cuda()
    return hip.pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], dev, w, h, sh).cpu().numpy()


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_vs_committed_reference_fixtures(hip, pair):
    i, o = pair
    n = 0
    for k in GOLD.files:
        if not k.startswith(f"out_{i}_{o}_"):
            continue
        _, _, _, dims, rs, gs, bs = k.split("_")
        w, h = map(int, dims.split("x"))
        got = _conv(hip, i, o, GOLD[f"in_{i}_{o}_{dims}"], w, h, (int(rs), int(gs), int(bs)))
        assert np.array_equal(got, GOLD[k]), k
        n += 1
    assert n >= 4


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_vs_oracle_ragged_and_aligned(hip, po, pair):
    i, o = pair
    for (w, h) in [(2, 2), (6, 1), (50, 3), (127, 5), (96, 4), (1920, 8), (3840, 4)]:
        for sh in [(0, 8, 16), (16, 8, 0)]:
            src = synth.s1_random(i, w, h, salt=w)
            want = po.convert_frame(i, o, src, w, h, sh)
            if po.have_ref():  # the real thing, when oracle/_ref travelled to this box
                assert np.array_equal(want, po.ref_convert_frame(i, o, src, w, h, sh, scalar=True))
            got = _conv(hip, i, o, src, w, h, sh)
            assert np.array_equal(got, want), (i, o, w, h, sh)


def test_identity_copies(hip, po):
    for f in ("UYVY", "v210", "YUYV"):
        src = synth.s1_random(f, 96, 4)
        assert np.array_equal(_conv(hip, f, f, src, 96, 4), src)


def test_baseline_config0_1080p_uyvy_to_rgb(hip, po):
    """BASELINE.json configs[0]: 1920x1080 UYVY->RGB -- GPU vs the reference CPU path, S1/S2/S3 content."""
    w, h = 1920, 1080
    for kind in ("S1", "S2", "S3"):
        src = synth.frame(kind, "UYVY", w, h)
        want = po.ref_convert_frame("UYVY", "RGB", src, w, h) if po.have_ref() else po.convert_frame("UYVY", "RGB", src, w, h)
        assert np.array_equal(_conv(hip, "UYVY", "RGB", src, w, h), want), kind


def test_full_size_roundtrip_properties(hip):
    """8K: UYVY -> v210 -> UYVY is the identity (8-bit samples survive <<2 then >>2); YUYV swap is an involution."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 7680, 4320
    src = torch.from_numpy(synth.s1_random("UYVY", w, 16)).cuda().repeat(h // 16)
    v = hip.pixfmt_convert(L.PF_UYVY, L.PF_V210, src, w, h)
    assert v.numel() == 20480 * h
    assert torch.equal(hip.pixfmt_convert(L.PF_V210, L.PF_UYVY, v, w, h), src)
    y = hip.pixfmt_convert(L.PF_UYVY, L.PF_YUYV, src, w, h)
    assert torch.equal(hip.pixfmt_convert(L.PF_YUYV, L.PF_UYVY, y, w, h), src)


def test_planar(hip, po):
    import torch
    for k in GOLD.files:
        if k.startswith("i420_in_"):
            dims = k.split("_")[2]
            w, h = map(int, dims.split("x"))
            y, u, v = hip.uyvy_to_i420(torch.from_numpy(GOLD[k]).cuda(), w, h)
            assert np.array_equal(y.cpu().numpy(), GOLD[f"i420_y_{dims}"]) and np.array_equal(u.cpu().numpy(), GOLD[f"i420_u_{dims}"]) \
                and np.array_equal(v.cpu().numpy(), GOLD[f"i420_v_{dims}"]), dims
        if k.startswith("p010_in_"):
            dims = k.split("_")[2]
            w, h = map(int, dims.split("x"))
            y, uv = hip.v210_to_p010le(torch.from_numpy(GOLD[k]).cuda(), w, h)
            assert np.array_equal(y.cpu().numpy().view(np.uint16), GOLD[f"p010_y_{dims}"]) and \
                np.array_equal(uv.cpu().numpy().view(np.uint16), GOLD[f"p010_uv_{dims}"]), dims
    # reference test pattern (test/codec_conversions_test.cpp:28-84) and full-size 4K (config 3 front end)
    for w, h in [(1, 2), (2, 1), (16, 1), (16, 16), (127, 255), (3840, 2160)]:
        src = synth.s1_random("UYVY", w, h, salt=1) if w > 16 else np.tile(np.frombuffer(b"uyvY", np.uint8), ((w + 1) // 2) * h)
        y, u, v = hip.uyvy_to_i420(torch.from_numpy(src).cuda(), w, h)
        for a, b in zip((y, u, v), po.uyvy_to_i420(src, w, h)):
            assert np.array_equal(a.cpu().numpy(), b), (w, h)
    src = synth.s1_random("v210", 1920, 8)
    y, uv = hip.v210_to_p010le(torch.from_numpy(src).cuda(), 1920, 8)
    wy, wuv = po.v210_to_p010le(src, 1920, 8)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), wy) and np.array_equal(uv.cpu().numpy().view(np.uint16), wuv)


@pytest.mark.gpu
def test_v210_to_p010le_ragged_geometry(hip, po):
    """every geometry the reference converts (to_planar.c:80-94,139-150) -- 1280x720, 2048x1080, 50x7, narrow and odd
    ones -- whole planes incl. the bytes the reference writes into the line padding, for paddings that do and do not hold the whole
    last group.  Checked against the restatement and, where oracle/_ref travelled, against the compiled reference itself."""
    import torch
    from test_oracle_pixfmt import P010_PADS, P010_RAGGED
    for w, h in P010_RAGGED + [(3838, 2159)]:
        src = synth.s1_random("v210", w, h, salt=w + h)
        dsrc = torch.from_numpy(src).cuda()
        for yp, up in P010_PADS:
            y, uv = hip.v210_to_p010le(dsrc, w, h, yp, up, 0x5A5A)
            got = (y.cpu().numpy().view(np.uint16), uv.cpu().numpy().view(np.uint16))
            wants = [po.v210_to_p010le(src, w, h, False, yp, up, 0x5A5A)]
            if po.have_ref():
                wants.append(po.v210_to_p010le(src, w, h, True, yp, up, 0x5A5A))
            for want in wants:
                for k, (a, b) in enumerate(zip(got, want)):
                    assert np.array_equal(a, b), (w, h, yp, up, k)
    # what the reference cannot convert without reading in front of its planes is refused, not guessed
    for w, h in [(50, 4), (7, 2), (1280, 3)]:
        with pytest.raises(Exception):
            hip.v210_to_p010le(torch.zeros(po.linesize(w, "v210") * h, dtype=torch.uint8, device="cuda"), w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("v210", "UYVY"), ("UYVY", "RGB"), ("RGB", "UYVY"), ("R10k", "RGB"), ("UYVY", "RGBA")])
def test_batch_equals_per_frame(hip, po, pair):
    """ug_hip_pixfmt_convert_batch: frames one picture apart (a single launch over frames * height lines) and frames at a padded stride
    (frame-by-frame launches) both equal the single-frame conversions."""
    import torch
    from ultragrid_amd import lib as L
    i, o = pair
    l = L.load()
    w, h, n = 96, 6, 4
    sls, dls = l.ug_hip_linesize(L.PF_NAMES[i], w), l.ug_hip_linesize(L.PF_NAMES[o], w)
    rng = np.random.default_rng(11)
    for pad in (0, 64):
        sfs, dfs = sls * h + pad, dls * h + pad
        src = torch.from_numpy(rng.integers(0, 256, n * sfs + 64, dtype=np.uint8)).cuda()
        dst = torch.zeros(n * dfs, dtype=torch.uint8, device="cuda")
        rc = l.ug_hip_pixfmt_convert_batch(L.PF_NAMES[i], L.PF_NAMES[o], src.data_ptr(), dst.data_ptr(), w, h, 0, 0, 0, 8, 16, n, sfs, dfs, None)
        assert rc == 0, L.last_error()
        torch.cuda.synchronize()
        for f in range(n):
            one = torch.zeros(dls * h, dtype=torch.uint8, device="cuda")
            assert l.ug_hip_pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], src[f * sfs:].data_ptr(), one.data_ptr(), w, h, 0, 0, 0, 8, 16, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(dst[f * dfs: f * dfs + dls * h], one), (pair, pad, f)


@pytest.mark.gpu
def test_convert_rejects_wild_shifts_and_short_pitches(hip):
    import torch
    from ultragrid_amd import lib as L
    l = L.load()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    assert l.ug_hip_pixfmt_convert(L.PF_RGB, L.PF_RGBA, buf.data_ptr(), buf[32768:].data_ptr(), 16, 4, 0, 0, 32, 8, 16, None) == L.EINVAL
    assert l.ug_hip_pixfmt_convert(L.PF_RGB, L.PF_RGBA, buf.data_ptr(), buf[32768:].data_ptr(), 16, 4, 0, 0, 0, 8, -1, None) == L.EINVAL
    assert l.ug_hip_pixfmt_convert(L.PF_RGB, L.PF_RGBA, buf.data_ptr(), buf[32768:].data_ptr(), 16, 4, 0, 60, 0, 8, 16, None) == L.EINVAL   # dst pitch < 64
    assert l.ug_hip_pixfmt_convert(L.PF_RGB, L.PF_RGBA, buf.data_ptr(), buf[32768:].data_ptr(), 16, 4, 40, 0, 0, 8, 16, None) == L.EINVAL   # src pitch < 48
    assert l.ug_hip_pixfmt_convert(L.PF_RGB, L.PF_RGBA, buf.data_ptr(), buf[32768:].data_ptr(), 16, 4, 48, 64, 0, 8, 16, None) == 0
