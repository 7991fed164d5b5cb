"""GPU parity: HIP DXT decoders vs oracle/dxt_decode_oracle.c (itself pinned to the reference's dxt62tga tool)."""
import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu


def _dec(hip, L, in_l, out_name, blocks, w, h, sh=(0, 8, 16)):
    import torch
    return hip.dxt_decode(in_l, L.PF_NAMES[out_name], torch.from_numpy(blocks).cuda(), w, h, sh).cpu().numpy()


@pytest.mark.parametrize("out", ["RGB", "BGR", "RGBA", "UYVY"])
@pytest.mark.parametrize("fmt", ["dxt1", "dxt5ycocg"])
def test_decode_bit_exact(hip, po, fmt, out):
    from ultragrid_amd import lib as L
    in_p, in_l = (po.OUT_DXT1, L.DXT1) if fmt == "dxt1" else (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)
    rng = np.random.default_rng(17)
    for (w, h) in [(4, 4), (64, 16), (200, 64), (1920, 32)]:
        cases = [po.dxt_encode(po.IN_RGB, in_p, synth.frame(k, "RGB", w, h), w, h) for k in ("S1", "S2", "S4")]
        cases.append(rng.integers(0, 256, cases[0].size, dtype=np.uint8))  # arbitrary bitstream: both alpha modes, 3-colour DXT1
        for blocks in cases:
            for sh in ([(0, 8, 16), (16, 8, 0)] if out == "RGBA" else [(0, 8, 16)]):
                got = _dec(hip, L, in_l, out, blocks, w, h, sh)
                want = po.dxt_decode(in_p, out, blocks, w, h, sh)
                assert np.array_equal(got, want), (fmt, out, w, h, sh)


def test_matches_reference_dxt62tga_tool(hip, po):
    from ultragrid_amd import lib as L
    if not po.have_ref():
        pytest.skip("oracle/_ref/dxt62tga not built")
    w, h = 256, 64
    blocks = po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, synth.s2_video("UYVY", w, h), w, h)
    got = _dec(hip, L, L.DXT5_YCOCG, "RGB", blocks, w, h).reshape(h, w, 3)
    assert np.array_equal(got, po.ref_dxt62tga(blocks, w, h))


def test_full_size_round_trip_psnr(hip, po):
    """4K UYVY -> DXT5-YCoCg -> RGB entirely on the GPU vs the reference's own UYVY->RGB of the source (Q14)."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 3840, 2160
    yy, xx = np.mgrid[0:64, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 40.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 63.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0 + xx / 300.0)], -1)
    rgb = np.tile(rgb.clip(0, 255).astype(np.uint8), (h // 64 + 1, 1, 1))[:h]
    src = torch.from_numpy(np.ascontiguousarray(rgb).ravel()).cuda()
    uyvy = hip.pixfmt_convert(L.PF_RGB, L.PF_UYVY, src, w, h)
    ref_rgb = hip.pixfmt_convert(L.PF_UYVY, L.PF_RGB, uyvy, w, h).cpu().numpy().astype(np.float64)
    for out_l, floor_db in ((L.DXT5_YCOCG, 38.0), (L.DXT1, 35.0)):
        blocks = hip.dxt_encode(L.PF_UYVY, out_l, uyvy, w, h)
        dec = hip.dxt_decode(out_l, L.PF_RGB, blocks, w, h).cpu().numpy().astype(np.float64)
        psnr = 10 * np.log10(255.0 ** 2 / np.mean((dec - ref_rgb) ** 2))
        assert psnr > floor_db, psnr


def test_decode_error_codes(hip):
    import torch
    from ultragrid_amd import lib as L
    b = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_decode(L.DXT1, L.PF_RGB, b, 18, 4)
    assert e.value.rc == L.EINVAL
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_decode(L.DXT1, L.PF_V210, b, 48, 4)
    assert e.value.rc == L.EUNSUPP
