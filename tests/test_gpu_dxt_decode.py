"""GPU parity: HIP DXT decoders vs oracle/dxt_decode_oracle.c (itself pinned to the reference's dxt62tga tool)."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu


def _dec(hip, L, in_l, out_name, blocks, w, h, sh=(0, 8, 16)):
    import torch
    return hip.dxt_decode(in_l, L.PF_NAMES[out_name], torch.from_numpy(blocks).cuda(), w, h, sh).cpu().numpy()


@pytest.mark.parametrize("out", ["RGB", "BGR", "RGBA", "UYVY"])
@pytest.mark.parametrize("fmt", ["dxt1", "dxt1_yuv", "dxt5ycocg"])
def test_decode_bit_exact(hip, po, fmt, out):
    from ultragrid_amd import lib as L
    in_p, in_l = {"dxt1": (po.OUT_DXT1, L.DXT1), "dxt1_yuv": (po.OUT_DXT1_YUV, L.DXT1_YUV), "dxt5ycocg": (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)}[fmt]
    rng = np.random.default_rng(17)
    for (w, h) in [(4, 4), (64, 16), (200, 64), (1920, 32)]:
        if fmt == "dxt1_yuv":
            cases = [po.dxt_encode(po.IN_UYVY_RAW, po.OUT_DXT1, synth.frame(k, "UYVY", w, h), w, h) for k in ("S1", "S2", "S4")]
        else:
            cases = [po.dxt_encode(po.IN_RGB, in_p, synth.frame(k, "RGB", w, h), w, h) for k in ("S1", "S2", "S4")]
        cases.append(rng.integers(0, 256, cases[0].size, dtype=np.uint8))  # arbitrary bitstream: both alpha modes, 3-colour DXT1
        for blocks in cases:
            for sh in ([(0, 8, 16), (16, 8, 0)] if out == "RGBA" else [(0, 8, 16)]):
                got = _dec(hip, L, in_l, out, blocks, w, h, sh)
                want = po.dxt_decode(in_p, out, blocks, w, h, sh)
                assert np.array_equal(got, want), (fmt, out, w, h, sh)


@pytest.mark.parametrize("fmt", ["dxt1", "dxt5ycocg"])
def test_rgba_shifts_with_alpha_in_front(hip, po, fmt):
    """component shifts that are not a permutation of {0, 8, 16} -- (8, 16, 24), (24, 16, 8), (24, 0, 8) -- place the channels
    where vc_copylineRGBA would (dxt_glsl.c:178) and 0xFF in the free byte; shifts that are not whole bytes are refused."""
    import torch
    from ultragrid_amd import lib as L
    in_p, in_l = {"dxt1": (po.OUT_DXT1, L.DXT1), "dxt5ycocg": (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)}[fmt]
    w, h = 512, 64
    enc = po.dxt_encode(po.IN_RGB, in_p, synth.frame("S2", "RGB", w, h), w, h)
    rnd = np.random.default_rng(5).integers(0, 256, enc.size, dtype=np.uint8)
    for blocks in (enc, rnd):
        for sh in [(8, 16, 24), (24, 16, 8), (24, 0, 8), (0, 16, 24), (16, 8, 0)]:
            got = _dec(hip, L, in_l, "RGBA", blocks, w, h, sh)
            want = po.dxt_decode(in_p, "RGBA", blocks, w, h, sh)
            assert np.array_equal(got, want), (fmt, sh)
    dst = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(enc).cuda()
    for sh in [(4, 8, 16), (0, 8, 32), (-8, 0, 8)]:
        assert L.load().ug_hip_dxt_decode(in_l, L.PF_RGBA, src.data_ptr(), dst.data_ptr(), w, h, 0, *sh, None) == L.EINVAL


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
        hip.dxt_decode(L.DXT1, L.PF_UYVY, b, 17, 4)         # (18 x 4 is a picture like any other: tests/test_gpu_dxt_edge.py)
    assert e.value.rc == L.EINVAL
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_decode(L.DXT1, L.PF_V210, b, 48, 4)
    assert e.value.rc == L.EUNSUPP


def test_dxt1_yuv_round_trip(hip, po):
    """UYVY -> DXT1_YUV (ug_hip_dxt_encode with UG_DXT1_YUV == UYVY_RAW -> DXT1) -> RGB through the display matrix: close to
    the BT.601-style conversion of the source the display shader implements (display_dxt1_yuv_fp.glsl:21-32)."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 512, 128
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 40.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 63.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0 + xx / 300.0)], -1).clip(0, 255).astype(np.uint8)
    uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
    dev = torch.from_numpy(uyvy).cuda()
    a = hip.dxt_encode(L.PF_UYVY, L.DXT1_YUV, dev, w, h)
    b = hip.dxt_encode(L.PF_UYVY_RAW, L.DXT1, dev, w, h)
    assert torch.equal(a, b) and a.numel() == w * h // 2
    assert np.array_equal(a.cpu().numpy(), po.dxt_encode(po.IN_UYVY_RAW, po.OUT_DXT1, uyvy, w, h))
    dec = hip.dxt_decode(L.DXT1_YUV, L.PF_RGB, a, w, h).cpu().numpy().reshape(h, w, 3).astype(float)
    u = uyvy.reshape(h, w // 2, 4).astype(float) / 255.0
    Y = 1.1643 * (np.stack([u[..., 1], u[..., 3]], -1).reshape(h, w) - 0.0625)
    U = 1.1384 * (np.repeat(u[..., 0], 2, axis=1) - 0.5)
    V = 1.1384 * (np.repeat(u[..., 2], 2, axis=1) - 0.5)
    ref = np.stack([Y + 1.5958 * V, Y - 0.39173 * U - 0.81290 * V, Y + 2.017 * U], -1).clip(0, 1) * 255
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((dec - ref) ** 2))
    assert psnr > 30, psnr
    with pytest.raises(RuntimeError):
        hip.dxt_encode(L.PF_RGB, L.DXT1_YUV, torch.zeros(64 * 64 * 3, dtype=torch.uint8, device="cuda"), 64, 64)


def test_constant_divisor_quotients_are_ieee_exact(hip):
    """The decoders' x / {255, 31, 63, 7, 5, 3} are computed as multiply + residual fma + correction fma; the device self-test compares
    them with the IEEE division for every numerator a DXT block can produce (all 256 x 256 alpha endpoint pairs x all table entries of
    both interpolation modes, all 5- / 6-bit endpoint pairs and their thirds)."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    n = C.c_uint(12345)
    assert L.load().ug_hip_selftest_dxt_decode(C.byref(n), torch.cuda.current_stream().cuda_stream) == 0
    assert n.value == 0


@pytest.mark.parametrize("out", ["RGBA", "RGB", "UYVY"])
def test_dxt5_fixed_point_path_equals_the_fp64_statements(hip, po, out):
    """Round 3: the DXT5-YCoCg decoder computes every sample in 32-bit fixed point, keeps it where it lies clear of an integer boundary and
    decodes the other blocks again with dxt62tga.c's fp64 statements.  Over 3.1 M blocks -- arbitrary bit patterns (both alpha modes,
    every palette), encoder output of video noise, and blocks built to sit ON boundaries (equal endpoints: every sample an exact integer
    + 0.5 ... ) -- the product (mode 0) equals the fp64-only kernel (mode 1) byte for byte; the guard fires (counter > 0) at about the
    predicted rate; and the fixed-point-only kernel (mode 2) may differ from mode 1 ONLY inside blocks the guard fired on -- shown by
    decoding the same frame twice and comparing block-wise.  The oracle pins mode 1 (test_decode_bit_exact runs in mode 0 too)."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    lib = L.load()
    w, h = 4096, 4096 * 3
    nblk = (w // 4) * (h // 4)
    g = torch.Generator(device="cuda").manual_seed(20260925)
    rnd = torch.randint(0, 2 ** 31 - 1, (nblk, 4), generator=g, device="cuda", dtype=torch.int32) ^ (torch.randint(0, 2, (nblk, 4), generator=g, device="cuda", dtype=torch.int32) << 31)
    third = nblk // 3
    # second third: encoder output of S2 video noise (real streams)
    noise = torch.from_numpy(synth.s2_video("UYVY", w, 256)).cuda().view(256, -1).repeat(16, 1)[: h // 3].contiguous()
    enc = hip.dxt_encode(L.PF_UYVY, L.DXT5_YCOCG, noise.view(-1), w, h // 3)
    rnd.view(torch.uint8).view(-1)[third * 16: third * 16 + enc.numel()] = enc
    # last third: adversarial structure -- equal alpha endpoints and/or equal colour endpoints (samples that are integers or integers + 0.5 in exact arithmetic)
    tail = rnd[2 * third:]
    a = tail[:, 0] & 0xFF
    tail[: tail.shape[0] // 2, 0] = (tail[: tail.shape[0] // 2, 0] & ~0xFFFF) | a[: tail.shape[0] // 2] | (a[: tail.shape[0] // 2] << 8)
    c = tail[tail.shape[0] // 4:, 2] & 0xFFFF
    tail[tail.shape[0] // 4:, 2] = c | (c << 16)
    blocks = rnd.view(torch.uint8).view(-1)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {}
    try:
        for mode in (1, 0, 2):
            counter.zero_()
            assert lib.ug_hip_dxt_decode_debug(mode, counter.data_ptr()) == 0
            res[mode] = hip.dxt_decode(L.DXT5_YCOCG, L.PF_NAMES[out], blocks, w, h)
            torch.cuda.synchronize()
            res[f"n{mode}"] = int(counter.item())
    finally:
        lib.ug_hip_dxt_decode_debug(0, None)
    assert torch.equal(res[0], res[1]), "the product differs from dxt62tga.c's fp64 statements"
    assert res["n1"] == 0 and res["n0"] == res["n2"] > 0
    # up to 48 distinct samples per block, 8 of 2^20 positions guarded: a few blocks in 10^5
    assert 1e-6 < res["n0"] / nblk < 0.01, res["n0"] / nblk
    bpp = {"RGBA": 4, "RGB": 3, "UYVY": 2}[out]
    diff = (res[2].view(h, w * bpp) != res[1].view(h, w * bpp))
    bad_blocks = int(diff.view(h // 4, 4, w // 4, 4 * bpp).any(dim=3).any(dim=1).sum().item())
    assert bad_blocks <= res["n0"], (bad_blocks, res["n0"])       # whatever differs without the fallback lies inside guarded blocks
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"dxt5_fixed_point_stats_{out}.txt"), "w") as f:
            f.write(f"DXT5-YCoCg -> {out}: {nblk} blocks; guarded (decoded again exactly): {res['n0']} = {res['n0'] / nblk:.3e}; blocks that differ from the fp64 "
                    f"statements when the fallback is switched off (mode 2): {bad_blocks}; product (mode 0) == fp64-only (mode 1): True\n")
    # The ORACLE over the whole set (no path may rest on a self-comparison): all 3.1 M blocks, every guarded one included,
    # product (mode 0) byte for byte against oracle/dxt_decode_oracle.c (OpenMP C; itself pinned to the reference's dxt62tga tool).
    want = po.dxt_decode(po.OUT_DXT5YCOCG, out, blocks.cpu().numpy(), w, h)
    got = res[0].cpu().numpy()
    if not np.array_equal(got, want):
        d = (got != want).reshape(h, w * bpp)
        rows, cols = np.nonzero(d)
        raise AssertionError(f"DXT5-YCoCg -> {out}: {int(d.sum())} bytes differ from the oracle, first at line {rows[0]} byte {cols[0]}")


@pytest.mark.parametrize("fmt", ["dxt1", "dxt1_yuv"])
@pytest.mark.parametrize("out", ["RGBA", "RGB", "UYVY"])
def test_dxt1_decode_full_4k_vs_oracle(hip, po, fmt, out):
    """A whole 3840x2160 frame of DXT1 / DXT1_YUV blocks -- half encoder output of video noise, half arbitrary bit patterns (3-colour mode,
    equal endpoints) -- against the oracle (dxt62tga.c:24-106 semantics), every output format the module offers."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 3840, 2160
    in_p, in_l = {"dxt1": (po.OUT_DXT1, L.DXT1), "dxt1_yuv": (po.OUT_DXT1_YUV, L.DXT1_YUV)}[fmt]
    src = torch.from_numpy(synth.s2_video("UYVY", w, h // 2)).cuda()
    enc = hip.dxt_encode(L.PF_UYVY_RAW if fmt == "dxt1_yuv" else L.PF_UYVY, L.DXT1, src, w, h // 2)
    g = torch.Generator(device="cuda").manual_seed(4 + len(out))
    rnd = torch.randint(0, 256, (enc.numel(),), generator=g, device="cuda", dtype=torch.uint8)
    eq = rnd.view(-1, 8)[::7]
    eq[:, 2:4] = eq[:, 0:2]                                   # every 7th random block: equal endpoints
    blocks = torch.cat([enc.view(-1), rnd])
    got = hip.dxt_decode(in_l, L.PF_NAMES[out], blocks, w, h).cpu().numpy()
    want = po.dxt_decode(in_p, out, blocks.cpu().numpy(), w, h)
    assert np.array_equal(got, want), (fmt, out, int((got != want).sum()))


