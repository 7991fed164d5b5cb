"""CPU: the strict-fp32 DXT restatement (oracle/dxt_oracle.c).  The reference has no CPU DXT encoder and no known-answer test, but
its encoders are GLSL shaders and those run on Mesa llvmpipe: the restatement in its default mode ("ties even") is PINNED to them
block for block (second half of this file, tests/golden/dxt_glsl_ref.npz).  Also checked: (a) regression against committed oracle
outputs in both tie modes, (b) the bitstream decodes -- with the restated reference CPU decoder cuda_dxt/dxt62tga.c -- to an image
close to the source, (c) structural S3TC invariants, and (d) a pure-Python/numpy-float32 re-derivation of one block from
SURVEY.md Appendix A."""
import os

import numpy as np
import pytest

from ultragrid_amd import synth

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dxt_oracle.npz"))
F32 = np.float32


def psnr(a, b):
    m = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m)


def test_regression_vs_committed_oracle_outputs(po):
    """out_/outm_ = default mode (ties even), outa_ = ties away"""
    fmts = {"RGB": po.IN_RGB, "RGBA": po.IN_RGBA, "UYVY": po.IN_UYVY, "v210": po.IN_V210}
    n = 0
    for k in GOLD.files:
        if not k.startswith("out"):
            continue
        tag, kind, name, oname = k.split("_")
        src = GOLD[f"in_{kind}_{name}"]
        h = -16 if tag == "outm" else 16
        got = po.dxt_encode(fmts[name], po.OUT_DXT1 if oname == "dxt1" else po.OUT_DXT5YCOCG, src, 48, h, ties="away" if tag == "outa" else "even")
        assert np.array_equal(got, GOLD[k]), k
        n += 1
    assert n == 72


@pytest.mark.parametrize("out", ["dxt1", "dxt5ycocg"])
def test_decode_psnr_gate(po, out):
    w, h = 288, 128
    oid = po.OUT_DXT1 if out == "dxt1" else po.OUT_DXT5YCOCG
    # smooth natural-image-like content (S2's per-pixel chroma noise is a stress input, not a quality gate)
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0),
                    128 + 80 * np.sin(yy / 9.0)], -1)
    rgb = (rgb + np.random.default_rng(2).normal(0, 2, rgb.shape)).clip(0, 255).astype(np.uint8)
    dec = po.dxt_decode_rgb(oid, po.dxt_encode(po.IN_RGB, oid, rgb, w, h), w, h)
    assert psnr(rgb, dec) > (35.0 if out == "dxt1" else 38.0)
    # UYVY path: compare with the reference's own UYVY->RGB (Q14) of the same frame
    uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
    ref_rgb = po.convert_frame("UYVY", "RGB", uyvy, w, h).reshape(h, w, 3)
    dec = po.dxt_decode_rgb(oid, po.dxt_encode(po.IN_UYVY, oid, uyvy, w, h), w, h)
    assert psnr(ref_rgb, dec) > (35.0 if out == "dxt1" else 38.0)
    # v210 path through the same content
    v210 = po.convert_frame("UYVY", "v210", uyvy, w, h) if w % 48 == 0 else None
    if v210 is not None:
        dec = po.dxt_decode_rgb(oid, po.dxt_encode(po.IN_V210, oid, v210, w, h), w, h)
        assert psnr(ref_rgb, dec) > (35.0 if out == "dxt1" else 38.0)


def test_structural_invariants(po):
    w, h = 128, 64
    for kind in ("S1", "S2", "S4"):
        src = synth.frame(kind, "RGB", w, h)
        d5 = po.dxt_encode(po.IN_RGB, po.OUT_DXT5YCOCG, src, w, h).view(np.uint32).reshape(-1, 4)
        a0, a1 = d5[:, 0] & 0xFF, (d5[:, 0] >> 8) & 0xFF
        assert (a0 >= a1).all()          # alpha0 = maxY, alpha1 = minY (glsl:252-259)
        c0, c1 = d5[:, 2] & 0xFFFF, d5[:, 2] >> 16
        assert ((c0 & 0x1F) == (c1 & 0x1F)).all() and ((c0 & 0x1F) <= 3).all() and ((c0 & 0x1F) != 2).all()  # scale-1 in {0,1,3}
        d1 = po.dxt_encode(po.IN_RGB, po.OUT_DXT1, src, w, h).view(np.uint16).reshape(-1, 4)
        assert (d1[:, 0] >= d1[:, 1]).all()  # 4-colour mode (compress_dxt1_fp.glsl:116-123)
    assert po.dxt_encode(po.IN_RGB, po.OUT_DXT1, synth.s1_random("RGB", 16, 8), 16, 8).size == 16 * 8 // 2  # dxt_util.h:59-67
    assert po.dxt_encode(po.IN_RGB, po.OUT_DXT1, np.zeros(18 * 5 * 3, np.uint8), 18, 5).size == 20 * 8 // 2  # both dimensions rounded up to 4


def test_format_equivalences(po):
    w, h = 96, 32
    uy = synth.s1_random("UYVY", w, h)
    y444 = po.yuv422_to_yuv444(uy, w * h)
    for oid in (po.OUT_DXT1, po.OUT_DXT5YCOCG):
        # the reference's two-pass CUDA path (422->444 kernel, then cuda_yuv_to_dxt*) == our fused definition
        assert np.array_equal(po.dxt_encode(po.IN_YUV444, oid, y444, w, h), po.dxt_encode(po.IN_UYVY, oid, uy, w, h))
        # v210 == vc_copylinev210 (>>2) then UYVY (cuda_dxt.cpp:162,213-218)
        v = synth.s1_random("v210", w, h)
        as_uyvy = po.convert_frame("v210", "UYVY", v, w, h)
        assert np.array_equal(po.dxt_encode(po.IN_V210, oid, v, w, h), po.dxt_encode(po.IN_UYVY, oid, as_uyvy, w, h))
        # RGBA ignores alpha
        rgba = synth.s1_random("RGBA", w, h)
        rgb = po.convert_frame("RGBA", "RGB", rgba, w, h)
        assert np.array_equal(po.dxt_encode(po.IN_RGBA, oid, rgba, w, h), po.dxt_encode(po.IN_RGB, oid, rgb, w, h))
        # negative height = vertical mirror of the source (cuda_dxt.cu:652-655)
        flipped = np.ascontiguousarray(rgb.reshape(h, w * 3)[::-1])
        assert np.array_equal(po.dxt_encode(po.IN_RGB, oid, rgb, w, -h), po.dxt_encode(po.IN_RGB, oid, flipped, w, h))


def _dxt5_block_numpy(rgb, ties="even"):
    """One block per SURVEY.md Appendix A in numpy float32 scalar arithmetic (independent re-derivation)."""
    f = F32
    off = f(128.0 / 255.0)
    px = rgb.astype(np.float32) * f(0.00392156862745)
    Y = ((px[:, 0] + f(2) * px[:, 1]) + px[:, 2]) * f(0.25)
    Co = ((f(2) * px[:, 0] - f(2) * px[:, 2]) * f(0.25)) + off
    Cg = (((-px[:, 0] + f(2) * px[:, 1]) - px[:, 2]) * f(0.25)) + off
    mn = [Y.min(), Co.min(), Cg.min()]; mx = [Y.max(), Co.max(), Cg.max()]
    midx, midy = (mx[1] + mn[1]) * f(0.5), (mx[2] + mn[2]) * f(0.5)
    cov = f(0)
    for i in range(16):
        cov = f(cov + f((Co[i] - midx) * (Cg[i] - midy)))
    if cov < 0:
        mx[2], mn[2] = mn[2], mx[2]
    m = max(abs(f(mn[1] - off)), abs(f(mn[2] - off)), abs(f(mx[1] - off)), abs(f(mx[2] - off)))
    scale = 1
    if m < f(64.0 / 255.0): scale = 2
    if m < f(32.0 / 255.0): scale = 4
    fs = f(scale)
    if ties == "away":
        rnd = lambda x: int(np.floor(np.float64(x) + 0.5))  # x >= 0: half away == floor(x + .5) in fp64 (exact)
    else:
        rnd = lambda x: int(np.rint(np.float64(x)))         # ties to even (Mesa's GLSL round)
    q = [f(31), f(63)]
    imax, imin, emx, emn = [], [], [], []
    for k in range(2):
        a = f(f(f(mx[1 + k] - off) * fs) + off); b = f(f(f(mn[1 + k] - off) * fs) + off)
        inset = f(f(f(a - b) / f(16)) - f((8.0 / 255.0) / 16.0))
        b = min(f(1), max(f(0), f(b + inset))); a = min(f(1), max(f(0), f(a - inset)))
        imax.append(rnd(f(a * q[k]))); imin.append(rnd(f(b * q[k])))
    w2 = ((imax[0] << 11) | (imax[1] << 5) | (scale - 1)) | (((imin[0] << 11) | (imin[1] << 5) | (scale - 1)) << 16)
    ex = lambda v, k: ((v << 3) | (v >> 2)) if k == 0 else ((v << 2) | (v >> 4))
    inv255 = f(1.0 / 255.0)
    for k in range(2):
        emx.append(f(f(f(f(ex(imax[k], k)) * inv255) - off) / fs + off)); emn.append(f(f(f(f(ex(imin[k], k)) * inv255) - off) / fs + off))
    q1, q2 = f(1.0 / 3.0), f(2.0 / 3.0)
    lerp = lambda a, b, t: f(f(a * f(f(1) - t)) + f(b * t))
    c = [emx, emn, [lerp(emx[k], emn[k], q1) for k in range(2)], [lerp(emx[k], emn[k], q2) for k in range(2)]]
    w3 = 0
    for i in range(16):
        d = [f(f(f(Co[i] - c[k][0]) * f(Co[i] - c[k][0])) + f(f(Cg[i] - c[k][1]) * f(Cg[i] - c[k][1]))) for k in range(4)]
        b0, b1, b2, b3, b4 = d[0] > d[3], d[1] > d[2], d[0] > d[2], d[1] > d[3], d[2] > d[3]
        w3 |= (int(b0 & b4) | (int((b1 & b2) | (b0 & b3)) << 1)) << (2 * i)
    inset = f(f(f(mx[0] - mn[0]) / f(32)) - f((16.0 / 255.0) / 32.0))
    mnY = min(f(1), max(f(0), f(mn[0] + inset))); mxY = min(f(1), max(f(0), f(mx[0] - inset)))
    w0 = (rnd(f(mnY * f(255))) << 8) | rnd(f(mxY * f(255)))
    mid = f(f(mxY - mnY) / f(14))
    ab = [None, f(mnY + mid)] + [f(f(f(f(f(8 - k) * mxY) + f(f(k - 1) * mnY)) * f(1.0 / 7.0)) + mid) for k in range(2, 8)]
    w1 = 0
    for i in range(16):
        idx = 1 + sum(int(Y[i] <= ab[k]) for k in range(1, 8))
        idx &= 7
        idx ^= int(2 > idx)
        if i < 6:
            w0 |= (idx << (3 * i + 16)) & 0xFFFFFFFF
            if i == 5:
                w1 = idx >> 1
        else:
            w1 |= idx << (3 * i - 16)
    return [w0, w1, w2, w3]


def test_independent_numpy_float32_rederivation(po):
    rng = np.random.default_rng(5)
    with np.errstate(all="ignore"):
        for trial in range(40):
            if trial % 4 == 0:
                blk = rng.integers(0, 256, (16, 3), dtype=np.uint8)
            elif trial % 4 == 1:  # low-contrast: exercises scale 2 / 4
                blk = (128 + rng.integers(-6, 7, (16, 3))).astype(np.uint8)
            elif trial % 4 == 2:  # flat
                blk = np.full((16, 3), rng.integers(0, 256), np.uint8)
            else:                 # smooth gradient
                blk = (np.linspace(40, 200, 16)[:, None] + rng.integers(-3, 4, (16, 3))).clip(0, 255).astype(np.uint8)
            for ties in ("even", "away"):
                got = po.dxt_encode(po.IN_RGB, po.OUT_DXT5YCOCG, blk.reshape(4, 12), 4, 4, ties=ties).view(np.uint32).tolist()
                assert got == _dxt5_block_numpy(blk, ties), (trial, ties)


def test_decode_oracle_pinned_to_reference_tool(po):
    """oracle/dxt_decode_oracle.c == the reference's own CPU decoder cuda_dxt/dxt62tga.c (compiled to
    oracle/_ref/dxt62tga), on encoder output and on arbitrary bitstreams (both alpha interpolation modes)."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/dxt62tga not built (no /root/reference on this box)")
    w, h = 96, 32
    rng = np.random.default_rng(3)
    cases = [po.dxt_encode(po.IN_RGB, po.OUT_DXT5YCOCG, synth.frame(k, "RGB", w, h), w, h) for k in ("S1", "S2", "S4")]
    cases += [rng.integers(0, 256, w * h, dtype=np.uint8) for _ in range(3)]
    for blocks in cases:
        assert np.array_equal(po.dxt_decode(po.OUT_DXT5YCOCG, "RGB", blocks, w, h).reshape(h, w, 3), po.ref_dxt62tga(blocks, w, h))


def test_round_u32_identity_used_by_the_hip_encoder():
    """dxt_encode.hip (UG_DXT_TIES_AWAY) computes (uint32_t) roundf(x) as `x < 0.5 ? 0 : (uint32_t)(x + 0.5f)`.  Pure IEEE arithmetic, so it is
    checked here on the CPU: every float within 512 ulps of k and k + 0.5 (k = 0..256) and 4M random floats in [0, 256].
    (An exhaustive sweep of all floats in [0, 2^22] was run once with the same result: no mismatch.)"""
    rng = np.random.default_rng(7)
    centres = np.concatenate([np.arange(0, 257, dtype=np.float32), np.arange(0, 257, dtype=np.float32) + np.float32(0.5)])
    bits = centres.view(np.uint32).astype(np.int64)[:, None] + np.arange(-512, 513)[None, :]
    near = bits[bits >= 0].astype(np.uint32).view(np.float32)
    x = np.concatenate([near, rng.uniform(0, 256, 4_000_000).astype(np.float32), rng.uniform(0, 1, 1_000_000).astype(np.float32)])
    s = (x + np.float32(0.5)).astype(np.float32)
    got = np.where(x < np.float32(0.5), 0, s.astype(np.uint32))
    want = np.floor(x.astype(np.float64) + 0.5).astype(np.uint32)  # roundf for x >= 0: half away from zero
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# The pin: the reference's OWN GLSL encoders, executed by Mesa llvmpipe (oracle/glsl_ref.c), committed as
# tests/golden/dxt_glsl_ref.npz by tests/golden/make_glsl_golden.py.
# ---------------------------------------------------------------------------------------------------------------------
GLSL_GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dxt_glsl_ref.npz"))
_PIN = {"RGB": "IN_RGB", "RGBA": "IN_RGBA", "UYVY": "IN_UYVY"}


def _glsl_cases():
    for key in GLSL_GOLD.files:
        if key.startswith("out_"):
            _, kind, fmt, mode = key.split("_")
            yield kind, fmt, mode


def _oracle_for(po, fmt, mode, src, w, h, ties="even"):
    pin = po.IN_UYVY_RAW if mode == "dxt1yuv" else getattr(po, _PIN[fmt])
    return po.dxt_encode(pin, po.OUT_DXT5YCOCG if mode == "dxt5" else po.OUT_DXT1, src, w, h, ties=ties)


@pytest.mark.parametrize("kind,fmt,mode", sorted(set(_glsl_cases())))
def test_restatement_reproduces_the_reference_glsl_shaders(po, kind, fmt, mode):
    """Bit-for-bit, every block: oracle/dxt_oracle.c == compress_dxt5ycocg_fp.glsl / compress_dxt1_fp.glsl (+ yuv422_to_yuv444.glsl)
    run on llvmpipe, in the oracle's default mode (the two implementation-defined GLSL choices set as Mesa makes them)."""
    w, h = (int(x) for x in GLSL_GOLD["size"])
    src = GLSL_GOLD[f"in_{kind}_{fmt}"]
    assert np.array_equal(_oracle_for(po, fmt, mode, src, w, h), GLSL_GOLD[f"out_{kind}_{fmt}_{mode}"])


@pytest.mark.parametrize("fmt", ["RGB", "UYVY"])
def test_restatement_vs_reference_glsl_big_frames_and_the_tie_rule(po, fmt):
    """4096-block uniform-random frames: identical to the shaders in the default mode (Mesa's choices); in "ties away" mode (roundf as
    in the reference's CUDA port, dot() left to right) only a few blocks differ, each by one endpoint LSB (an exact .5 tie of round())
    or one palette index (a distance near-tie)."""
    w, h = 512, 128
    src = synth.s1_random(fmt, w, h, salt=77)
    crc = int(np.sum(src.astype(np.uint64) * (np.arange(src.size, dtype=np.uint64) % 251 + 1)))
    assert crc == int(GLSL_GOLD[f"big_{fmt}_crc"][0]), "synthetic frame generator changed: regenerate tests/golden/dxt_glsl_ref.npz"
    for mode in ("dxt5", "dxt1"):
        gold = GLSL_GOLD[f"big_{fmt}_{mode}"]
        assert np.array_equal(_oracle_for(po, fmt, mode, src, w, h), gold), mode
        dflt = _oracle_for(po, fmt, mode, src, w, h, ties="away")
        bs = 16 if mode == "dxt5" else 8
        differ = np.flatnonzero((dflt.reshape(-1, bs) != gold.reshape(-1, bs)).any(axis=1))
        assert differ.size <= 0.01 * gold.size // bs, (mode, differ.size)        # ties are rare
        for i in differ:
            a, b = dflt.reshape(-1, bs)[i].astype(int), gold.reshape(-1, bs)[i].astype(int)
            changed = np.flatnonzero(a != b)
            assert changed.size == 1 and (abs(a[changed[0]] - b[changed[0]]) == 1 or bin(a[changed[0]] ^ b[changed[0]]).count("1") <= 2), (mode, i, a, b)


# ---- sizes that are not multiples of 4 (dxt_util.h:59-67; VERDICT r5 "What's missing" #3) ----
def _golden_generator():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_glsl_golden", os.path.join(os.path.dirname(__file__), "golden", "make_glsl_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _edge_cases():
    for key in GLSL_GOLD.files:
        if key.startswith("edge_out_"):
            _, _, size, fmt, mode = key.split("_")
            w, h = (int(x) for x in size.split("x"))
            yield w, h, fmt, mode


@pytest.mark.parametrize("w,h,fmt,mode", sorted(set(_edge_cases())))
def test_restatement_reproduces_the_reference_glsl_shaders_at_unaligned_sizes(po, w, h, fmt, mode):
    """Widths = 1, 2, 3 (mod 4): the shaders' own GL_CLAMP_TO_EDGE fetches, the reference as it is.  Heights != 0 (mod 4): the reference's
    output for the picture padded with copies of its last line (generator: tests/golden/make_glsl_golden.py).  Bit for bit, every block."""
    src = GLSL_GOLD[f"edge_in_{w}x{h}_{fmt}"]
    gold = GLSL_GOLD[f"edge_out_{w}x{h}_{fmt}_{mode}"]
    got = _oracle_for(po, fmt, mode, src, w, h)
    assert got.size == ((w + 3) // 4 * 4) * ((h + 3) // 4 * 4) // (1 if mode == "dxt5" else 2)  # dxt_get_size
    assert np.array_equal(got, gold)


def test_unaligned_size_is_the_padded_picture(po):
    """What the rule amounts to: the blocks of a w x h picture == the blocks of the picture padded to whole blocks by repeating its last
    column and last line -- and a bottom-up source (negative height) is flipped first, then padded."""
    rng = np.random.default_rng(5)
    for w, h in ((10, 6), (7, 9), (1, 1), (33, 2), (1366, 5)):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        W4, H4 = (w + 3) // 4 * 4, (h + 3) // 4 * 4
        padded = np.pad(rgb, ((0, H4 - h), (0, W4 - w), (0, 0)), mode="edge")
        for out in (po.OUT_DXT5YCOCG, po.OUT_DXT1):
            assert np.array_equal(po.dxt_encode(po.IN_RGB, out, rgb, w, h), po.dxt_encode(po.IN_RGB, out, padded, W4, H4)), (w, h)
            flipped = np.pad(rgb[::-1], ((0, H4 - h), (0, W4 - w), (0, 0)), mode="edge")
            assert np.array_equal(po.dxt_encode(po.IN_RGB, out, rgb, w, -h), po.dxt_encode(po.IN_RGB, out, flipped, W4, H4)), (w, h)
    for w, h in ((10, 6), (2, 3), (1366, 5)):  # 4:2:2: the last pixel repeats with its own chroma pair
        uyvy = rng.integers(0, 256, (h, w // 2, 4), dtype=np.uint8)
        W4, H4 = (w + 3) // 4 * 4, (h + 3) // 4 * 4
        last = uyvy[:, -1:, :].copy()
        last[:, :, 1] = last[:, :, 3]
        padded = np.pad(np.concatenate([uyvy] + [last] * ((W4 - w) // 2), axis=1), ((0, H4 - h), (0, 0), (0, 0)), mode="edge")
        for pin in (po.IN_UYVY, po.IN_UYVY_RAW):
            assert np.array_equal(po.dxt_encode(pin, po.OUT_DXT1, uyvy, w, h), po.dxt_encode(pin, po.OUT_DXT1, padded, W4, H4)), (w, h)
    with pytest.raises(ValueError):
        po.dxt_encode(po.IN_UYVY, po.OUT_DXT1, np.zeros(12, np.uint8), 3, 2)  # 4:2:2 needs an even width


@pytest.mark.parametrize("w,h", [(16, 10), (8, 6), (8, 486)])
def test_reference_height_slip(po, w, h):
    """Record of the ONE deliberate deviation: with a height that is not a multiple of 4 the reference draws h/4 block rows
    over the whole texture height (glViewport(.., height / 4), dxt_encoder.c:380, against imageSize.y = (height + 3) / 4 * 4, :393): block
    row `by` is cut from lines floor(h * ((by + 0.5) / (h / 4) + (i - 1.5) / H4)), i = 0..3 -- a vertical resampling -- and the last block
    row of the stream is never rendered.  The committed vectors are the reference's shaders executed on exactly such sizes."""
    edge_input = _golden_generator().edge_input
    src = edge_input(w, h, "RGBA").reshape(h, w, 4)
    ref = GLSL_GOLD[f"slip_h_{w}x{h}_RGBA_dxt5"].reshape((h + 3) // 4, -1)
    H4 = (h + 3) // 4 * 4
    lines = [min(h - 1, max(0, int(np.floor(h * ((by + 0.5) / (h // 4) + (i - 1.5) / H4))))) for by in range(h // 4) for i in range(4)]
    resampled = np.ascontiguousarray(src[lines])
    assert np.array_equal(po.dxt_encode(po.IN_RGBA, po.OUT_DXT5YCOCG, resampled, w, 4 * (h // 4)).reshape(h // 4, -1), ref[:h // 4])
    assert not ref[h // 4:].any()          # unrendered (glsl_ref starts from a zeroed buffer; the reference from whatever the FBO held)
    ours = po.dxt_encode(po.IN_RGBA, po.OUT_DXT5YCOCG, src, w, h).reshape((h + 3) // 4, -1)
    assert lines[:4] != [0, 1, 2, 3] or lines[-4:] != list(range(4 * (h // 4) - 4, 4 * (h // 4)))   # it IS a resampling
    assert ours[h // 4:].any()             # ... where this implementation encodes the remaining lines


@pytest.mark.parametrize("w,h", [(10, 8), (6, 8)])
def test_reference_rgb_unpack_alignment_slip(po, w, h):
    """Second record: the reference uploads GL_RGB lines without ever setting GL_UNPACK_ALIGNMENT, so for 3 w % 4 != 0 GL takes line y at
    byte y * ((3 w + 3) & ~3) of a buffer packed at 3 w (dxt_encoder.c:562-575; dxt_glsl.cpp:169 packs at 3 w): a skewed picture, and a read
    past the end.  The committed vectors are the shaders run on a packed buffer (zero-filled behind its end)."""
    edge_input = _golden_generator().edge_input
    packed = edge_input(w, h, "RGB").ravel()
    stride = (3 * w + 3) // 4 * 4
    flat = np.concatenate([packed, np.zeros(stride * h - packed.size, np.uint8)])
    skewed = np.stack([flat[y * stride: y * stride + 3 * w] for y in range(h)])
    ref = GLSL_GOLD[f"slip_rgb_{w}x{h}_dxt1"]
    assert np.array_equal(po.dxt_encode(po.IN_RGB, po.OUT_DXT1, skewed, w, h), ref)
    assert not np.array_equal(po.dxt_encode(po.IN_RGB, po.OUT_DXT1, packed, w, h), ref)


def test_decode_oracle_at_unaligned_sizes(po):
    """Receiver side: the stream holds (w+3)/4 x (h+3)/4 blocks, the picture shown is w x h (dxt_decoder.c:146-149,368-389): decoding the
    blocks at the padded size and cropping is the same thing."""
    rng = np.random.default_rng(11)
    for w, h in ((10, 6), (7, 9), (1, 1), (1366, 5), (6, 4)):
        W4, H4 = (w + 3) // 4 * 4, (h + 3) // 4 * 4
        for fmt, bs in ((po.OUT_DXT5YCOCG, 16), (po.OUT_DXT1, 8), (po.OUT_DXT1_YUV, 8)):
            blocks = rng.integers(0, 256, W4 * H4 // 16 * bs, dtype=np.uint8)
            for out, bpp in (("RGB", 3), ("RGBA", 4), ("BGR", 3), ("UYVY", 2)):
                if out == "UYVY" and w % 2:
                    with pytest.raises(ValueError):
                        po.dxt_decode(fmt, out, blocks, w, h)
                    continue
                full = po.dxt_decode(fmt, out, blocks, W4, H4).reshape(H4, W4 * bpp)
                assert np.array_equal(po.dxt_decode(fmt, out, blocks, w, h).reshape(h, -1)[:, :w * bpp], full[:h, :w * bpp]), (w, h, fmt, out)


def test_live_reference_glsl_when_available(po):
    """In the build container the shaders are run live on other geometries than the committed vectors."""
    if not po.have_glsl_ref():
        pytest.skip("oracle/_ref/glsl_ref or /root/reference not available")
    for (w, h, fmt, mode, kind) in [(4, 4, "RGB", "dxt5", "S1"), (8, 4, "UYVY", "dxt5", "S1"), (200, 36, "RGBA", "dxt1", "S2"), (132, 60, "UYVY", "dxt1yuv", "S2"),
                                    (1920, 64, "UYVY", "dxt5", "S2"), (256, 256, "RGB", "dxt1", "S1")]:
        src = synth.frame(kind, fmt, w, h, 9)
        assert np.array_equal(_oracle_for(po, fmt, mode, src, w, h), po.ref_glsl_dxt_encode(mode, fmt.lower(), src, w, h)), (w, h, fmt, mode)
    rng = np.random.default_rng(21)
    for (w, h, fmt, mode) in [(1366, 768, "UYVY", "dxt5"), (1998, 40, "RGBA", "dxt1"), (50, 20, "RGB", "dxt5"), (2, 8, "UYVY", "dxt1yuv"), (721, 12, "RGBA", "dxt5")]:
        src = rng.integers(0, 256, (h, w * {"RGB": 3, "RGBA": 4, "UYVY": 2}[fmt]), dtype=np.uint8)
        assert np.array_equal(_oracle_for(po, fmt, mode, src, w, h), po.ref_glsl_dxt_encode(mode, fmt.lower(), src, w, h, gl_row_stride=True)), (w, h, fmt, mode)


def test_decode_oracle_vs_the_reference_gl_decoder_when_available(po):
    """Receiver side, informational bound: the reference's GL decoder (fixed-function S3TC fetch + display_*_fp.glsl, dxt_decoder.c) run
    on Mesa llvmpipe vs the decode oracle (which is bit-equal to the reference's CPU tool cuda_dxt/dxt62tga.c).  GL leaves the S3TC
    interpolation precision and the unorm conversions to the implementation, so this is a tolerance, not an identity."""
    if not po.have_glsl_ref():
        pytest.skip("oracle/_ref/glsl_ref or /root/reference not available")
    import subprocess
    import tempfile
    w, h = 256, 64
    src = synth.s2_video("UYVY", w, h)
    for mode, enc_in, enc_out, dec_in, tol in (("dec5", po.IN_UYVY, po.OUT_DXT5YCOCG, po.OUT_DXT5YCOCG, 3), ("dec1", po.IN_UYVY, po.OUT_DXT1, po.OUT_DXT1, 2),
                                               ("dec1yuv", po.IN_UYVY_RAW, po.OUT_DXT1, po.OUT_DXT1_YUV, 5)):
        blocks = po.dxt_encode(enc_in, enc_out, src, w, h)
        with tempfile.TemporaryDirectory() as d:
            a, b = os.path.join(d, "b.dxt"), os.path.join(d, "o.rgba")
            blocks.tofile(a)
            subprocess.check_call([po.GLSL_REF, "/root/reference", mode, "rgba", str(w), str(h), a, b])
            gl = np.fromfile(b, np.uint8).reshape(h, w, 4)[..., :3].astype(int)
        ours = po.dxt_decode(dec_in, "RGB", blocks, w, h).reshape(h, w, 3).astype(int)
        diff = np.abs(gl - ours)
        assert diff.max() <= tol and diff.mean() < 1.0, (mode, diff.max(), diff.mean())


def test_rgba_to_yuv422_restatement_vs_the_reference_shader_when_available(po):
    """The 4:2:2 output pass of the receiver (rgba_to_yuv422.glsl) alone, run on llvmpipe on the texels our decode oracle produces:
    the oracle's restatement is byte-identical in its default mode (exact .5 ties of the float -> unorm8 framebuffer write -- the one
    implementation-defined choice here -- to even, as Mesa does); "ties away" (half up) differs in a few bytes per million, by one."""
    if not po.have_glsl_ref():
        pytest.skip("oracle/_ref/glsl_ref or /root/reference not available")
    import subprocess
    import tempfile
    w, h = 512, 128
    for kind, salt in (("S2", 0), ("S1", 0)):
        blocks = po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, synth.frame(kind, "UYVY", w, h, salt), w, h)
        rgba = po.dxt_decode(po.OUT_DXT5YCOCG, "RGBA", blocks, w, h)
        with tempfile.TemporaryDirectory() as d:
            a, b = os.path.join(d, "i.rgba"), os.path.join(d, "o.uyvy")
            rgba.tofile(a)
            subprocess.check_call([po.GLSL_REF, "/root/reference", "rgba2uyvy", "rgba", str(w), str(h), a, b])
            gl = np.fromfile(b, np.uint8)
        assert np.array_equal(po.dxt_decode(po.OUT_DXT5YCOCG, "UYVY", blocks, w, h), gl), kind
        dflt = po.dxt_decode(po.OUT_DXT5YCOCG, "UYVY", blocks, w, h, ties="away").astype(int)
        diff = np.abs(dflt - gl.astype(int))
        assert diff.max() <= 1 and np.count_nonzero(diff) < 1e-4 * diff.size
