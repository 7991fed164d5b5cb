"""GPU parity at frame sizes that are not multiples of 4 (dxt_util.h:59-67, dxt_encoder.c:362-394; VERDICT r5 "What's missing" #3):
product == oracle == the reference's shaders executed at such sizes (tests/golden/dxt_glsl_ref.npz "edge_*"), encode and decode, through
the C ABI.  The rule: the stream holds (w+3)/4 x (h+3)/4 blocks; columns / lines past the picture repeat its last column / line."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu
GLSL_GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dxt_glsl_ref.npz"))

FMTS = ["RGB", "RGBA", "UYVY", "v210", "YUV444", "UYVY_RAW"]


def _ids(po, L):
    pin = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY, "v210": po.IN_V210, "YUV444": po.IN_YUV444, "UYVY_RAW": po.IN_UYVY_RAW}
    lin = {"RGB": L.PF_RGB, "RGBA": L.PF_RGBA, "UYVY": L.PF_UYVY, "v210": L.PF_V210, "YUV444": L.PF_YUV444, "UYVY_RAW": L.PF_UYVY_RAW}
    return pin, lin


def _base(fmt):
    return {"YUV444": "RGB", "UYVY_RAW": "UYVY"}.get(fmt, fmt)


def _edge_cases():
    for key in GLSL_GOLD.files:
        if key.startswith("edge_out_"):
            _, _, size, fmt, mode = key.split("_")
            w, h = (int(x) for x in size.split("x"))
            yield w, h, fmt, mode


@pytest.mark.parametrize("w,h,fmt,mode", sorted(set(_edge_cases())))
def test_product_equals_the_executed_reference_shaders(hip, po, w, h, fmt, mode):
    """The committed outputs of compress_dxt5ycocg_fp.glsl / compress_dxt1_fp.glsl (+ yuv422_to_yuv444.glsl) run by Mesa llvmpipe at sizes
    that are not multiples of 4 -- the product reproduces every block (and so does the oracle: tests/test_oracle_dxt.py)."""
    import torch
    from ultragrid_amd import lib as L
    src = GLSL_GOLD[f"edge_in_{w}x{h}_{fmt}"]
    gold = GLSL_GOLD[f"edge_out_{w}x{h}_{fmt}_{mode}"]
    lin = {"RGB": L.PF_RGB, "RGBA": L.PF_RGBA, "UYVY": L.PF_UYVY}[fmt]
    lout = {"dxt5": L.DXT5_YCOCG, "dxt1": L.DXT1, "dxt1yuv": L.DXT1_YUV}[mode]
    buf = torch.zeros(max(src.size, 16), dtype=torch.uint8, device="cuda")
    buf[:src.size] = torch.from_numpy(src.ravel()).cuda()
    got = hip.dxt_encode(lin, lout, buf, w, h).cpu().numpy()
    assert got.size == gold.size == hip.dxt_size(lout, w, h)
    assert np.array_equal(got, gold)


SIZES = [(1, 1), (2, 2), (3, 5), (6, 4), (10, 8), (14, 9), (254, 7), (258, 10), (770, 6), (1366, 12), (1998, 8), (50, 3), (4, 5), (8, 6), (16, 7)]


@pytest.mark.parametrize("ties", ["even", "away"])
@pytest.mark.parametrize("out", ["dxt1", "dxt5ycocg"])
@pytest.mark.parametrize("fmt", FMTS)
def test_encode_bit_exact_vs_oracle(hip, po, fmt, out, ties):
    """every input format x every residue of width and height mod 4, one block column up to several waves wide; top-down and bottom-up
    (negative height: flipped first, then padded)"""
    import torch
    from ultragrid_amd import lib as L
    pin, lin = _ids(po, L)
    oid_p, oid_l = (po.OUT_DXT1, L.DXT1) if out == "dxt1" else (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)
    for (w, h) in SIZES:
        if (w & 1) and _base(fmt) in ("UYVY", "v210"):
            continue
        src = synth.s1_random(_base(fmt), w, h, salt=w + h)
        buf = torch.zeros(max(src.size, 16), dtype=torch.uint8, device="cuda")
        buf[:src.size] = torch.from_numpy(src).cuda()
        for hh in (h, -h):
            got = hip.dxt_encode(lin[fmt], oid_l, buf, w, hh, ties=None if ties == "even" else L.TIES_AWAY).cpu().numpy()
            want = po.dxt_encode(pin[fmt], oid_p, src, w, hh, ties=ties)
            bad = np.nonzero(got != want)[0]
            assert got.size == want.size and bad.size == 0, f"{fmt}->{out} {w}x{hh}: {bad.size} bytes differ, first block {bad[0] // (8 if out == 'dxt1' else 16)}"


@pytest.mark.parametrize("mode", [(720, 486, "UYVY"), (720, 486, "v210"), (2048, 858, "UYVY"), (1998, 1080, "RGB"), (1366, 768, "UYVY"), (1366, 768, "RGBA"),
                                  (1366, 768, "v210"), (1366, 768, "RGB")], ids=lambda m: f"{m[0]}x{m[1]}-{m[2]}")
def test_real_video_modes_full_frames(hip, po, mode):
    """NTSC 720x486, 2K scope 2048x858, 2K flat 1998x1080, WXGA 1366x768 (the modes VERDICT r5 names): whole frames, video-like content"""
    import torch
    from ultragrid_amd import lib as L
    w, h, fmt = mode
    pin, lin = _ids(po, L)
    src = synth.frame("S2", fmt, w, h, 3) if not (fmt == "v210" and w % 6) else synth.s1_random("v210", w, h, 3)
    dev = torch.from_numpy(src).cuda()
    for oid_p, oid_l in ((po.OUT_DXT5YCOCG, L.DXT5_YCOCG), (po.OUT_DXT1, L.DXT1)):
        got = hip.dxt_encode(lin[fmt], oid_l, dev, w, h).cpu().numpy()
        assert np.array_equal(got, po.dxt_encode(pin[fmt], oid_p, src, w, h, threads=0))


def test_pitch_batch_and_canaries(hip, po):
    """a pitch larger than the line, several frames per launch with strides, and guard bytes behind every destination: nothing is written
    past (w+3)/4 x (h+3)/4 blocks; an exactly sized source (last line ends the allocation's payload) is enough"""
    import torch
    from ultragrid_amd import lib as L
    pin, lin = _ids(po, L)
    rng = np.random.default_rng(4)
    for fmt, w, h, pitch in (("RGB", 10, 6, 31), ("RGB", 1366, 5, 4100), ("RGBA", 9, 7, 40), ("UYVY", 14, 9, 32), ("UYVY", 1366, 6, 2736), ("v210", 14, 5, 128), ("v210", 1366, 7, 3840)):
        frames = 3
        fstride = (pitch * h + 15) // 16 * 16
        src = rng.integers(0, 256, (frames, fstride), dtype=np.uint8)
        if fmt == "v210":
            src = (src.view(np.uint32) & 0x3FFFFFFF).view(np.uint8)
        per = po.dxt_size(po.OUT_DXT5YCOCG, w, h)
        dstride = per + 32
        dst = torch.full((frames * dstride,), 0xA5, dtype=torch.uint8, device="cuda")
        dev = torch.from_numpy(src.ravel()).cuda()
        rc = L.load().ug_hip_dxt_encode_batch(lin[fmt], L.DXT5_YCOCG, dev.data_ptr(), dst.data_ptr(), w, h, pitch, frames, fstride, dstride, None)
        assert rc == 0, (fmt, w, h, L.last_error())
        out = dst.cpu().numpy().reshape(frames, dstride)
        for f in range(frames):
            want = po.dxt_encode(pin[fmt], po.OUT_DXT5YCOCG, src[f], w, h, pitch=pitch)
            assert np.array_equal(out[f, :per], want), (fmt, w, h, f)
            assert (out[f, per:] == 0xA5).all(), (fmt, w, h, f)


def test_refusals(hip, po):
    import torch
    from ultragrid_amd import lib as L
    l = L.load()
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    for fmt in (L.PF_UYVY, L.PF_UYVY_RAW, L.PF_V210):       # a 4:2:2 line is made of pixel pairs
        assert l.ug_hip_dxt_encode(fmt, L.DXT1, p, p + 2048, 7, 4, 0, None) == L.EINVAL
    assert l.ug_hip_dxt_encode(L.PF_RGBA, L.DXT1, p, p + 2048, 6, 4, 26, None) == L.EINVAL   # RGBA pitch % 4
    assert l.ug_hip_dxt_encode(L.PF_UYVY, L.DXT1, p, p + 2048, 6, 4, 14, None) == L.EINVAL   # UYVY pitch % 4
    assert l.ug_hip_dxt_encode(L.PF_RGB, L.DXT1, p, p + 2048, 6, 4, 19, None) == 0           # 3 * width bytes per line: any pitch
    assert l.ug_hip_dxt_encode(L.PF_RGB, L.DXT1, p, p + 2048, 8, 4, 25, None) == L.EINVAL    # ... when the width is a multiple of 4: pitch % 4 as before
    for fn in (l.ug_hip_rgb_to_dxt1, l.ug_hip_yuv_to_dxt1, l.ug_hip_rgb_to_dxt6, l.ug_hip_yuv_to_dxt6):  # cuda_dxt.h's own limit (cuda_dxt.cu:745)
        assert fn(p, p + 2048, 6, 4, None) == L.EINVAL and fn(p, p + 2048, 8, 6, None) == L.EINVAL and fn(p, p + 2048, 8, 4, None) == 0
    assert l.ug_hip_dxt_decode(L.DXT1, L.PF_UYVY, p, p + 2048, 7, 4, 0, 0, 8, 16, None) == L.EINVAL   # UYVY out: even width
    assert l.ug_hip_dxt_size(L.DXT5_YCOCG, 1366, 766) == 1368 * 768 and l.ug_hip_dxt_size(L.DXT1, 5, 5) == 32
    torch.cuda.synchronize()


DEC_SIZES = [(1, 1), (2, 2), (3, 5), (6, 4), (10, 8), (14, 9), (254, 7), (258, 10), (1366, 12), (1998, 8), (4, 5), (16, 7)]


@pytest.mark.parametrize("ties", ["even", "away"])
@pytest.mark.parametrize("out", ["RGBA", "RGB", "BGR", "UYVY"])
@pytest.mark.parametrize("fmt", ["dxt5", "dxt1", "dxt1yuv"])
def test_decode_bit_exact_vs_oracle(hip, po, fmt, out, ties):
    """arbitrary block contents (every decoder path: both alpha modes, 3-colour DXT1 blocks, the guarded fixed-point blocks) at every
    residue of width / height; guard bytes behind the picture stay untouched"""
    import torch
    from ultragrid_amd import lib as L
    pf = {"dxt5": (po.OUT_DXT5YCOCG, L.DXT5_YCOCG, 16), "dxt1": (po.OUT_DXT1, L.DXT1, 8), "dxt1yuv": (po.OUT_DXT1_YUV, L.DXT1_YUV, 8)}[fmt]
    lout = {"RGBA": L.PF_RGBA, "RGB": L.PF_RGB, "BGR": L.PF_BGR, "UYVY": L.PF_UYVY}[out]
    rng = np.random.default_rng(8)
    for (w, h) in DEC_SIZES:
        if out == "UYVY" and (w & 1):
            continue
        nblk = ((w + 3) // 4) * ((h + 3) // 4)
        blocks = rng.integers(0, 256, nblk * pf[2], dtype=np.uint8)
        for shifts in (((0, 8, 16), (16, 8, 0), (8, 16, 24)) if out == "RGBA" else ((0, 8, 16),)):
            want = po.dxt_decode(pf[0], out, blocks, w, h, shifts=shifts, ties=ties)
            dst = torch.full((want.size + 64,), 0x5A, dtype=torch.uint8, device="cuda")
            src = torch.zeros(max(16, blocks.size), dtype=torch.uint8, device="cuda")
            src[:blocks.size] = torch.from_numpy(blocks).cuda()
            rc = L.load().ug_hip_dxt_decode_ex(pf[1], lout, src.data_ptr(), dst.data_ptr(), w, h, 0, *shifts, L.TIES_EVEN if ties == "even" else L.TIES_AWAY, None)
            assert rc == 0, (fmt, out, w, h, L.last_error())
            got = dst.cpu().numpy()
            assert np.array_equal(got[:want.size], want), (fmt, out, w, h, shifts)
            assert (got[want.size:] == 0x5A).all(), (fmt, out, w, h)


def _psnr(a, b):
    m = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m)


@pytest.mark.parametrize("size", [(720, 486), (2048, 858), (1366, 768)])
def test_round_trip_quality_matches_the_aligned_case(hip, po, size):
    """encode -> decode of a picture whose size is not a multiple of 4 is as good as the same content at the next smaller multiple of 4"""
    import torch
    from ultragrid_amd import lib as L
    w, h = size
    src = synth.frame("S2", "UYVY", w, h, 5)
    dev = torch.from_numpy(src).cuda()
    for oid in (L.DXT5_YCOCG, L.DXT1):
        back = hip.dxt_decode(oid, L.PF_UYVY, hip.dxt_encode(L.PF_UYVY, oid, dev, w, h), w, h).cpu().numpy().reshape(h, 2 * w)
        p_edge = _psnr(back, src.reshape(h, 2 * w))
        wa, ha = w // 4 * 4, h // 4 * 4
        crop = np.ascontiguousarray(src.reshape(h, 2 * w)[:ha, :2 * wa])
        back_a = hip.dxt_decode(oid, L.PF_UYVY, hip.dxt_encode(L.PF_UYVY, oid, torch.from_numpy(crop.ravel()).cuda(), wa, ha), wa, ha).cpu().numpy().reshape(ha, 2 * wa)
        p_al = _psnr(back_a, crop)
        assert p_edge >= p_al - 0.05, (size, oid, p_edge, p_al)
        # the last column / last line themselves are as good as the rest
        assert _psnr(back[h - (h % 4 or 4):], src.reshape(h, 2 * w)[h - (h % 4 or 4):]) >= p_al - 3.0
