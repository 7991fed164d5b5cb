"""GPU parity: HIP DXT encoders (through the C ABI) vs oracle/dxt_oracle.c -- bit-exact."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dxt_oracle.npz"))

FMTS = ["RGB", "RGBA", "UYVY", "v210", "YUV444", "UYVY_RAW"]
OUTS = ["dxt1", "dxt5ycocg"]


def _ids(po, L):
    pin = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY, "v210": po.IN_V210, "YUV444": po.IN_YUV444,
           "UYVY_RAW": po.IN_UYVY_RAW}
    lin = {"RGB": L.PF_RGB, "RGBA": L.PF_RGBA, "UYVY": L.PF_UYVY, "v210": L.PF_V210, "YUV444": L.PF_YUV444,
           "UYVY_RAW": L.PF_UYVY_RAW}
    return pin, lin


def _src(kind, fmt, w, h, salt=0):
    base = {"YUV444": "RGB", "UYVY_RAW": "UYVY"}.get(fmt, fmt)
    if base == "v210" and w % 6 and kind in ("S2", "S3"):   # the smooth generators write whole 6-pixel groups
        return synth.s1_random("v210", w, h, salt + len(kind))
    return synth.frame(kind, base, w, h, salt) if kind != "S3" or base in ("UYVY", "RGB") else synth.frame("S2", base, w, h, salt)


def _run(hip, po, fmt, out, src, w, h, ties="even", threads=1):
    """ties = "even": the library DEFAULT (plain ug_hip_dxt_encode) vs the oracle's default; "away": the explicit option on both sides"""
    import torch
    from ultragrid_amd import lib as L
    pin, lin = _ids(po, L)
    oid_p, oid_l = (po.OUT_DXT1, L.DXT1) if out == "dxt1" else (po.OUT_DXT5YCOCG, L.DXT5_YCOCG)
    got = hip.dxt_encode(lin[fmt], oid_l, torch.from_numpy(src).cuda(), w, h, ties=None if ties == "even" else L.TIES_AWAY).cpu().numpy()
    want = po.dxt_encode(pin[fmt], oid_p, src, w, h, ties=ties, threads=threads)
    return got, want


@pytest.mark.parametrize("ties", ["even", "away"])
@pytest.mark.parametrize("out", OUTS)
@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("kind", ["S1", "S2", "S3", "S4"])
def test_bit_exact_small(hip, po, fmt, out, kind, ties):
    # v210: widths that are not multiples of 12 (partial last unit of a line: 1280x720, 2048x1080 class) included
    for (w, h) in [(48, 16), (192, 64), (1920, 36), (4, 4), (8, 8), (52, 8), (200, 16), (1280, 8), (2048, 4)] if fmt == "v210" else [(4, 4), (8, 8), (48, 16), (200, 64), (1920, 36)]:
        src = _src(kind, fmt, w, h, salt=w)
        got, want = _run(hip, po, fmt, out, src, w, h, ties)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{fmt}->{out} {kind} {w}x{h}: {bad.size} bytes differ, first block {bad[0] // (8 if out == 'dxt1' else 16)}"


@pytest.mark.parametrize("out", OUTS)
@pytest.mark.parametrize("fmt", ["RGB", "UYVY", "v210"])
def test_vertical_mirror(hip, po, fmt, out):
    w, h = 96, 32
    src = _src("S1", fmt, w, h)
    got, want = _run(hip, po, fmt, out, src, w, -h)
    assert np.array_equal(got, want)


def test_committed_golden(hip, po):
    import torch
    from ultragrid_amd import lib as L
    _, lin = _ids(po, L)
    for k in GOLD.files:
        if not k.startswith("out"):
            continue
        tag, kind, name, oname = k.split("_")
        src = GOLD[f"in_{kind}_{name}"]
        h = -16 if tag == "outm" else 16
        got = hip.dxt_encode(lin[name], L.DXT1 if oname == "dxt1" else L.DXT5_YCOCG, torch.from_numpy(src).cuda(), 48, h,
                             ties=L.TIES_AWAY if tag == "outa" else None)
        assert np.array_equal(got.cpu().numpy(), GOLD[k]), k


@pytest.mark.parametrize("cfg", [("RGB", "dxt1", 1920, 1080), ("UYVY", "dxt5ycocg", 3840, 2160)],
                         ids=["cfg1-1080p-RGB-DXT1", "cfg2-4K-UYVY-DXT5"])
def test_baseline_configs_full_size(hip, po, cfg):
    """BASELINE.json configs[1] and configs[2] at full size, bit-exact vs the oracle (S2 content)."""
    fmt, out, w, h = cfg
    src = _src("S2", fmt, w, h)
    got, want = _run(hip, po, fmt, out, src, w, h)
    assert np.array_equal(got, want)


def test_8k_v210_full_frame(hip, po):
    """configs[4] at 7680x4320, the WHOLE frame against the oracle (run on all host cores by row bands), plus the size-independent
    properties: v210 == (v210->UYVY on the GPU) -> UYVY encoder, batch == per-frame, mirror == encode of the flipped frame."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 7680, 4320
    one = synth.s2_video("v210", w, 48)
    reps = h // 48
    frame = np.tile(one.reshape(48, -1), (reps, 1))
    rng = np.random.default_rng(9)
    frame[:, :4096] ^= rng.integers(0, 256, (h, 4096), dtype=np.uint8)  # de-periodise: every block row differs
    frame = (frame.view(np.uint32) & 0x3FFFFFFF).view(np.uint8).ravel()  # pad bits 0
    dev = torch.from_numpy(frame).cuda()
    full = hip.dxt_encode(L.PF_V210, L.DXT5_YCOCG, dev, w, h)
    want = po.dxt_encode(po.IN_V210, po.OUT_DXT5YCOCG, frame, w, h, threads=0)
    assert np.array_equal(full.cpu().numpy(), want)
    pitch = 20480
    uyvy = hip.pixfmt_convert(L.PF_V210, L.PF_UYVY, dev, w, h)
    assert torch.equal(hip.dxt_encode(L.PF_UYVY, L.DXT5_YCOCG, uyvy, w, h), full)
    two = torch.cat([dev, dev.flip(0).contiguous()])
    b = hip.dxt_encode_batch(L.PF_V210, L.DXT5_YCOCG, two, w, h, 2, dev.numel())
    assert torch.equal(b[: full.numel()], full)
    flipped = torch.from_numpy(np.ascontiguousarray(frame.reshape(h, pitch)[::-1])).cuda().ravel()
    assert torch.equal(hip.dxt_encode(L.PF_V210, L.DXT5_YCOCG, dev, w, -h), hip.dxt_encode(L.PF_V210, L.DXT5_YCOCG, flipped, w, h))


@pytest.mark.parametrize("size", [(1280, 720), (2048, 1080)])
def test_v210_widths_not_divisible_by_12(hip, po, size):
    """1280x720 and 2048x1080 v210 (the reference takes them: vc_copylinev210 with its partial-group tail, pixfmt_conv.c:121-130, then the
    width % 4 encoder, cuda_dxt.cpp:206-220, cuda_dxt.cu:745): fused kernel == oracle == (compiled-reference-style v210->UYVY) -> UYVY
    encoder, random 10-bit content."""
    import torch
    from ultragrid_amd import lib as L
    w, h = size
    src = synth.s1_random("v210", w, h, salt=w)
    dev = torch.from_numpy(src).cuda()
    as_uyvy = po.convert_frame("v210", "UYVY", src, w, h)
    for out_l, out_p in ((L.DXT5_YCOCG, po.OUT_DXT5YCOCG), (L.DXT1, po.OUT_DXT1)):
        got = hip.dxt_encode(L.PF_V210, out_l, dev, w, h).cpu().numpy()
        assert np.array_equal(got, po.dxt_encode(po.IN_UYVY, out_p, as_uyvy, w, h, threads=0))
        assert np.array_equal(got, po.dxt_encode(po.IN_V210, out_p, src, w, h, threads=0))
        got_m = hip.dxt_encode(L.PF_V210, out_l, dev, w, -h).cpu().numpy()
        assert np.array_equal(got_m, po.dxt_encode(po.IN_V210, out_p, src, w, -h, threads=0))


def test_cuda_dxt_h_shaped_entry_points(hip, po):
    """ug_hip_{rgb,yuv}_to_dxt{1,6} + ug_hip_yuv422_to_yuv444 == the reference's two-pass CUDA call sequence
    (cuda_dxt.cpp:223-260)."""
    import torch
    from ultragrid_amd import lib as L
    l = L.load()
    w, h = 192, 64
    st = torch.cuda.current_stream().cuda_stream
    uy = synth.s1_random("UYVY", w, h)
    d_uy = torch.from_numpy(uy).cuda()
    d444 = hip.yuv422_to_yuv444(d_uy, w * h)
    assert np.array_equal(d444.cpu().numpy(), po.yuv422_to_yuv444(uy, w * h))
    for fn, pf, oid, size in [(l.ug_hip_yuv_to_dxt6, po.IN_YUV444, po.OUT_DXT5YCOCG, w * h), (l.ug_hip_yuv_to_dxt1, po.IN_YUV444, po.OUT_DXT1, w * h // 2)]:
        out = torch.empty(size, dtype=torch.uint8, device="cuda")
        assert fn(d444.data_ptr(), out.data_ptr(), w, h, st) == 0
        assert np.array_equal(out.cpu().numpy(), po.dxt_encode(pf, oid, d444.cpu().numpy(), w, h))
        assert np.array_equal(out.cpu().numpy(), po.dxt_encode(po.IN_UYVY, oid, uy, w, h))  # fused == two-pass
    rgb = synth.s1_random("RGB", w, h)
    d_rgb = torch.from_numpy(rgb).cuda()
    for fn, oid, size in [(l.ug_hip_rgb_to_dxt6, po.OUT_DXT5YCOCG, w * h), (l.ug_hip_rgb_to_dxt1, po.OUT_DXT1, w * h // 2)]:
        out = torch.empty(size, dtype=torch.uint8, device="cuda")
        assert fn(d_rgb.data_ptr(), out.data_ptr(), w, -h, st) == 0   # negative height = bottom-up
        assert np.array_equal(out.cpu().numpy(), po.dxt_encode(po.IN_RGB, oid, rgb, w, -h))


def test_batch_and_pitch(hip, po):
    import torch
    from ultragrid_amd import lib as L
    w, h, n = 96, 32, 5
    frames = [synth.s1_random("UYVY", w, h, salt=i) for i in range(n)]
    dev = torch.from_numpy(np.concatenate(frames)).cuda()
    out = hip.dxt_encode_batch(L.PF_UYVY, L.DXT5_YCOCG, dev, w, h, n, frames[0].size).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i * w * h: (i + 1) * w * h], po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, frames[i], w, h))
    # padded pitch
    pitch = 2 * w + 64
    padded = np.zeros((h, pitch), np.uint8)
    padded[:, : 2 * w] = frames[0].reshape(h, 2 * w)
    got = hip.dxt_encode(L.PF_UYVY, L.DXT1, torch.from_numpy(padded.ravel()).cuda(), w, h, pitch=pitch).cpu().numpy()
    assert np.array_equal(got, po.dxt_encode(po.IN_UYVY, po.OUT_DXT1, frames[0], w, h))


def test_error_codes_on_device(hip):
    import torch
    from ultragrid_amd import lib as L
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_encode(L.PF_UYVY, L.DXT1, buf, 17, 4)       # (18 x 4 RGB is a picture like any other: tests/test_gpu_dxt_edge.py)
    assert e.value.rc == L.EINVAL
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_encode(L.PF_RG48, L.DXT1, buf, 16, 4)
    assert e.value.rc == L.EUNSUPP
    with pytest.raises(ValueError):
        hip.dxt_encode(L.PF_RGB, L.DXT1, torch.zeros(64, dtype=torch.uint8), 4, 4)  # host tensor: no CPU fallback


def test_alpha_reference_form_path(po):
    """The DXT5 kernel counts alpha thresholds by binary search when they are monotone (always, for real
    content) and falls back to the reference's 7-compare form otherwise.  The fallback is unreachable with
    byte-quantised input in practice, so a test build that forces it (-DUG_FORCE_ALPHA_LINEAR, built by
    __graft_entry__.build()) is checked against the oracle in a subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(root, "ultragrid_amd", "libug_mi355x_alphalinear.so")
    if not os.path.exists(alt):
        pytest.skip("alphalinear test build missing (run __graft_entry__.build())")
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import pyoracle as po
from ultragrid_amd import codec, lib, synth
assert "alphalinear" in lib.LIB_PATH
for kind in ("S1", "S2", "S4"):
    for (fmt, pf, pin) in (("UYVY", lib.PF_UYVY, po.IN_UYVY), ("RGB", lib.PF_RGB, po.IN_RGB)):
        src = synth.frame(kind, fmt, 192, 64)
        got = codec.dxt_encode(pf, lib.DXT5_YCOCG, torch.from_numpy(src).cuda(), 192, 64).cpu().numpy()
        assert np.array_equal(got, po.dxt_encode(pin, po.OUT_DXT5YCOCG, src, 192, 64)), (kind, fmt)
print("OK")
''' % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UG_MI355X_LIB=alt), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_extreme_and_degenerate_content(hip, po):
    """All-0 / all-255 / super-white / super-black / checkerboard frames: degenerate min == max blocks, clamps on both
    sides, Y outside [0,1] after the unclamped YUV->RGB (compress_dxt5ycocg_fp.glsl:12-23)."""
    import torch
    from ultragrid_amd import lib as L
    w, h = 96, 32
    pats = {"zeros": np.zeros(2 * w * h, np.uint8), "ones": np.full(2 * w * h, 255, np.uint8),
            "superwhite": np.tile(np.array([128, 255, 128, 254], np.uint8), w * h // 2),
            "superblack": np.tile(np.array([128, 0, 128, 1], np.uint8), w * h // 2),
            "chroma_extremes": np.tile(np.array([0, 128, 255, 128, 255, 127, 0, 129], np.uint8), w * h // 4),
            "checker": np.tile(np.array([16, 235, 240, 16, 240, 16, 16, 235], np.uint8), w * h // 4)}
    for name, src in pats.items():
        for out_l, out_p in ((L.DXT5_YCOCG, po.OUT_DXT5YCOCG), (L.DXT1, po.OUT_DXT1)):
            got = hip.dxt_encode(L.PF_UYVY, out_l, torch.from_numpy(src).cuda(), w, h).cpu().numpy()
            assert np.array_equal(got, po.dxt_encode(po.IN_UYVY, out_p, src, w, h)), name
            rgb = src[: 3 * w * h] if src.size >= 3 * w * h else np.resize(src, 3 * w * h)
            got = hip.dxt_encode(L.PF_RGB, out_l, torch.from_numpy(np.ascontiguousarray(rgb)).cuda(), w, h).cpu().numpy()
            assert np.array_equal(got, po.dxt_encode(po.IN_RGB, out_p, rgb, w, h)), name


def test_fast_index_stages_on_their_boundaries(hip, po):
    """The encoders pick ONE exact comparison per pixel from the pixel's position on the palette segment / among the alpha thresholds
    (dxt_encode.hip, UG_DXT_FAST_INDEX) and run the reference's full form for a whole wave that holds a block outside the
    precondition.  Content built for the seams: waves that mix flat blocks (coincident end points -> full form) with busy ones,
    two-colour blocks whose end points are one quantisation step apart (shortest non-zero segment), pixels ON the segment at
    multiples of 1/6 of it (every zone border and every bisector), luma ramps that put alpha values on the thresholds, and
    YUV-derived values outside [0, 1]."""
    import torch
    from ultragrid_amd import lib as L
    rng = np.random.default_rng(77)
    w, h = 512, 32  # 128 blocks per block row = two full waves per row
    bw, bh = w // 4, h // 4

    def blocks_to_rgb(bl):  # bl: (bh, bw, 4, 4, 3) uint8
        return np.ascontiguousarray(bl.transpose(0, 2, 1, 3, 4).reshape(h, w, 3)).ravel()

    frames = {}
    # 1. every 7th block flat, the rest noise
    bl = rng.integers(0, 256, (bh, bw, 4, 4, 3), dtype=np.uint8)
    flat = (np.arange(bh * bw).reshape(bh, bw) % 7) == 3
    bl[flat] = rng.integers(0, 256, (int(flat.sum()), 1, 1, 3), dtype=np.uint8)
    frames["flat_among_noise"] = bl
    # 2. two colours per block, 1..3 LSB apart in one or more channels
    base = rng.integers(8, 247, (bh, bw, 1, 1, 3), dtype=np.int32)
    delta = rng.integers(0, 4, (bh, bw, 1, 1, 3), dtype=np.int32)
    pick = rng.integers(0, 2, (bh, bw, 4, 4, 1), dtype=np.int32)
    frames["two_colours_close"] = (base + delta * pick).astype(np.uint8)
    # 3. pixels on the segment between two random colours at k/6 (+- 1 LSB of rounding)
    a = rng.integers(0, 256, (bh, bw, 1, 1, 3)).astype(np.float64)
    b = rng.integers(0, 256, (bh, bw, 1, 1, 3)).astype(np.float64)
    t = rng.integers(0, 7, (bh, bw, 4, 4, 1)).astype(np.float64) / 6.0
    frames["on_the_segment"] = np.clip(np.rint(a + (b - a) * t) + rng.integers(-1, 2, (bh, bw, 4, 4, 3)), 0, 255).astype(np.uint8)
    # 4. grey ramps of every slope: alpha (luma) values on and around the thresholds, chroma flat
    lo = rng.integers(0, 200, (bh, bw, 1, 1, 1)); step = rng.integers(0, 4, (bh, bw, 1, 1, 1))
    ramp = lo + step * np.arange(16).reshape(1, 1, 4, 4, 1)
    frames["grey_ramps"] = np.broadcast_to(np.clip(ramp, 0, 255), (bh, bw, 4, 4, 3)).astype(np.uint8)
    for name, bl in frames.items():
        rgb = blocks_to_rgb(bl)
        for out_l, out_p in ((L.DXT5_YCOCG, po.OUT_DXT5YCOCG), (L.DXT1, po.OUT_DXT1)):
            for ties in ("even", "away"):
                got = hip.dxt_encode(L.PF_RGB, out_l, torch.from_numpy(rgb).cuda(), w, h, ties=None if ties == "even" else L.TIES_AWAY).cpu().numpy()
                want = po.dxt_encode(po.IN_RGB, out_p, rgb, w, h, ties=ties)
                assert np.array_equal(got, want), (name, out_p, ties, int(np.count_nonzero(got != want)))
        # the same bytes read as UYVY (2 B / px: the first two thirds of the buffer): unclamped YUV -> RGB, luma below 0 and above 1
        uy = np.ascontiguousarray(rgb[: 2 * w * h])
        for out_l, out_p in ((L.DXT5_YCOCG, po.OUT_DXT5YCOCG), (L.DXT1, po.OUT_DXT1)):
            got = hip.dxt_encode(L.PF_UYVY, out_l, torch.from_numpy(uy).cuda(), w, h).cpu().numpy()
            assert np.array_equal(got, po.dxt_encode(po.IN_UYVY, out_p, uy, w, h)), (name, "UYVY", out_p)


def test_fast_index_stage_coverage_by_content(hip, po):
    """Which form runs is a property of the content (ug_hip_dxt_encode_stats): video-like frames (S2) stay in the fast stages for all but
    a few per cent of the waves (S2's saturated chroma drives the unclamped RGB, and with it the luma range of some blocks, out of
    [0, 1]: end points clamp together) and flat frames for every wave (a one-colour block gets the full form for its one value inside the fast stage); a frame whose blocks hold
    two chroma values one LSB apart (end points quantise to the same value over non-flat chroma) sends every wave through the
    reference's full colour form.  The alpha stage never leaves the fast form on byte content.  All three are bit-equal to the oracle."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    l = L.load()
    w, h = 1920, 1080
    waves = (h // 4) * ((w // 4 + 63) // 64)
    st = (C.c_ulonglong * 2)()
    almost = np.tile(np.array([200, 90, 60, 90, 201, 90, 60, 90], np.uint8), w * h // 4)
    frames = {"S2": (synth.s2_video("UYVY", w, h), None), "flat": (np.full(2 * w * h, 90, np.uint8), 0), "almost_flat": (almost, waves)}
    seen = []
    for out, pout in ((L.DXT5_YCOCG, po.OUT_DXT5YCOCG), (L.DXT1, po.OUT_DXT1)):
        for name, (src, want_full) in frames.items():
            assert l.ug_hip_dxt_encode_stats(None, 1) == 0
            got = hip.dxt_encode(L.PF_UYVY, out, torch.from_numpy(src).cuda(), w, h).cpu().numpy()
            assert l.ug_hip_dxt_encode_stats(st, 1) == 0
            seen.append(f"{name} -> {'DXT5-YCoCg' if out == L.DXT5_YCOCG else 'DXT1'}: colour full form {st[0]} / alpha full form {st[1]} of {waves} waves")
            if want_full is None:
                assert st[0] < 0.02 * waves and st[1] < 0.05 * waves, seen[-1]
            else:
                assert (st[0], st[1]) == (want_full, 0), seen[-1]
            assert np.array_equal(got, po.dxt_encode(po.IN_UYVY, pout, src, w, h, threads=8)), (name, out)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dxt_fast_index_coverage.txt", "w") as f:
        f.write("\n".join(seen) + "\n")


def test_concurrent_streams_threads(hip, po):
    """Distinct streams driven from distinct threads (the tile fan-out of video_compress.cpp:441-490)."""
    import threading
    import torch
    from ultragrid_amd import lib as L
    w, h = 384, 128
    srcs = [synth.s1_random("UYVY", w, h, salt=i) for i in range(4)]
    outs = [None] * 4

    def work(i):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            d = torch.from_numpy(srcs[i]).cuda()
            for _ in range(20):
                o = hip.dxt_encode(L.PF_UYVY, L.DXT5_YCOCG, d, w, h)
            st.synchronize()
            outs[i] = o.cpu().numpy()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(4):
        assert np.array_equal(outs[i], po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, srcs[i], w, h)), i


def test_shipped_library_reproduces_the_reference_glsl_shaders(hip, po):
    """THE PIN.  The product library in its default mode against the reference's own GLSL encoders run on Mesa llvmpipe
    (tests/golden/dxt_glsl_ref.npz, generated by executing compress_dxt5ycocg_fp.glsl / compress_dxt1_fp.glsl / yuv422_to_yuv444.glsl):
    EVERY block identical, the 4096-block uniform-random frames included.  In "ties away" mode (the CUDA text's roundf / left-to-right
    dot) the library equals the oracle's same mode and differs from the executed shaders only in a few tie blocks."""
    import os
    import torch
    from ultragrid_amd import lib as L
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "dxt_glsl_ref.npz"))
    w, h = (int(x) for x in gold["size"])
    total = same_away = n = 0
    for key in gold.files:
        if not key.startswith("out_"):
            continue
        _, kind, fmt, mode = key.split("_")
        src = gold[f"in_{kind}_{fmt}"]
        in_l = L.PF_UYVY_RAW if mode == "dxt1yuv" else L.PF_NAMES[fmt]
        out_l = L.DXT5_YCOCG if mode == "dxt5" else L.DXT1
        dev = torch.from_numpy(src).cuda()
        got = hip.dxt_encode(in_l, out_l, dev, w, h).cpu().numpy()            # plain ug_hip_dxt_encode: the default
        assert np.array_equal(got, gold[key]), key
        assert np.array_equal(hip.dxt_encode(in_l, out_l, dev, w, h, ties=L.TIES_EVEN).cpu().numpy(), gold[key]), key
        n += got.size
        away = hip.dxt_encode(in_l, out_l, dev, w, h, ties=L.TIES_AWAY).cpu().numpy()
        pin = po.IN_UYVY_RAW if mode == "dxt1yuv" else {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY}[fmt]
        assert np.array_equal(away, po.dxt_encode(pin, po.OUT_DXT5YCOCG if mode == "dxt5" else po.OUT_DXT1, src, w, h, ties="away")), key
        bs = 16 if mode == "dxt5" else 8
        eq = (away.reshape(-1, bs) == gold[key].reshape(-1, bs)).all(axis=1)
        total += eq.size
        same_away += int(eq.sum())
    for fmt in ("RGB", "UYVY"):
        src = synth.s1_random(fmt, 512, 128, salt=77)
        for mode in ("dxt5", "dxt1"):
            got = hip.dxt_encode(L.PF_NAMES[fmt], L.DXT5_YCOCG if mode == "dxt5" else L.DXT1, torch.from_numpy(src).cuda(), 512, 128).cpu().numpy()
            assert np.array_equal(got, gold["big_%s_%s" % (fmt, mode)]), (fmt, mode)
            n += got.size
    assert n > 250_000   # bytes of compressed blocks compared with the executed shaders
    assert total > 7000 and 0.97 < same_away / total < 1.0, (same_away, total)   # S3 (flat colour bars) sits on round() ties in every white block


def test_unknown_tie_rule_is_rejected(hip):
    import torch
    from ultragrid_amd import lib as L
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(L.UgHipError) as e:
        hip.dxt_encode(L.PF_RGB, L.DXT1, buf, 16, 4, ties=7)
    assert e.value.rc == L.EINVAL


def test_encoder_strength_reductions_are_ieee_exact(hip):
    """div14(): x / 14.0f as multiply + two fma, compared on the device with the IEEE division for x = 0 and EVERY fp32 bit pattern in
    [2^-100, 1]: zero mismatches (the kernel routes the sliver (0, 2^-100), where the quotient goes denormal, to the division itself)."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    n = C.c_uint(777)
    assert L.load().ug_hip_selftest_dxt_encode(C.byref(n), torch.cuda.current_stream().cuda_stream) == 0
    assert n.value == 0
