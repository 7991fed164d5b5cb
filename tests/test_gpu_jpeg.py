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


@pytest.mark.parametrize("dims", [(16, 16), (160, 96), (200, 120), (1920, 1080)], ids=str)
@pytest.mark.parametrize("ri", [1, 4, 7])
def test_full_jpeg_stream(hip, po, dims, ri):
    """Complete encoder (FDCT+quant + Huffman + JFIF with restart intervals): the byte stream equals the test writer fed with
    the oracle's coefficients, and an independent decoder (Pillow/libjpeg) reconstructs the picture."""
    import io
    import torch
    from PIL import Image
    from jpeg_bitstream import write_jpeg420
    w, h = dims
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    rgb = rgb.clip(0, 255).astype(np.uint8)
    uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h) if w % 2 == 0 else None
    q = 75
    enc = hip.JpegEncoder(w, h, q, ri)
    data = enc.encode(torch.from_numpy(uyvy).cuda())
    enc.close()
    y, u, v = po.uyvy_to_i420(uyvy, w, h)
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    mw, mh = (w + 15) // 16, (h + 15) // 16
    want = write_jpeg420(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, 2 * mh),
                         po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh), po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh), restart=ri)
    assert data == want, (len(data), len(want))
    img = Image.open(io.BytesIO(data))
    img.draft("YCbCr", None)
    dec = np.asarray(img)
    assert dec.shape == (h, w, 3)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((dec[..., 0].astype(float) - y.astype(float)) ** 2))
    assert psnr > 40, psnr


def test_jpeg_stream_random_content_and_stuffing(hip, po):
    """Uniform-random frames at q=100 produce long codes and 0xFF bytes: exercises ZRL runs and byte stuffing."""
    import io
    import torch
    from PIL import Image
    from jpeg_bitstream import write_jpeg420
    w, h = 128, 64
    uyvy = synth.s1_random("UYVY", w, h)
    enc = hip.JpegEncoder(w, h, 100, 2)
    data = enc.encode(torch.from_numpy(uyvy).cuda())
    y, u, v = po.uyvy_to_i420(uyvy, w, h)
    ql, qc = po.jpeg_qtable(100, 0), po.jpeg_qtable(100, 1)
    want = write_jpeg420(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 16, 8), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), 8, 4),
                         po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), 8, 4), restart=2)
    assert data == want
    assert b"\xff\x00" in data  # stuffing happened
    img = Image.open(io.BytesIO(data))
    img.draft("YCbCr", None)   # raw YCbCr planes, no RGB round trip
    dec = np.asarray(img)
    assert np.abs(dec[..., 0].astype(int) - y.astype(int)).mean() < 1.5


def _want_422(po, uyvy, w, h, q):
    y, u, v = po.uyvy_to_i422(uyvy, w, h, use_ref=po.have_ref() and w % 2 == 0)
    mw, mh = (w + 15) // 16, (h + 7) // 8
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    dl, dc = po.jpeg_divisors(ql), po.jpeg_divisors(qc)
    return (y, u, v), (ql, qc), (po.jpeg_fdct_quant_plane(y, dl, 2 * mw, mh), po.jpeg_fdct_quant_plane(u, dc, mw, mh),
                                 po.jpeg_fdct_quant_plane(v, dc, mw, mh))


@pytest.mark.parametrize("dims", [(16, 8), (40, 24), (50, 37), (51, 9), (1920, 1080), (3840, 2160)], ids=str)
def test_fused_uyvy_422_pipeline(hip, po, dims):
    """4:2:2 (the sampling the reference module selects for UYVY, gpujpeg.cpp:295-302,339): UYVY -> uyvy_to_i422 planes ->
    FDCT+quant, fused on the GPU (fast MCU-strip kernel for width % 16 == 0, generic kernel otherwise)."""
    import torch
    w, h = dims
    src = synth.s2_video("UYVY", w, h) if w % 2 == 0 else synth.s1_random("UYVY", w, h)
    _, _, want = _want_422(po, src, w, h, 75)
    got = hip.uyvy_to_jpeg_coeffs(torch.from_numpy(src).cuda(), w, h, hip.jpeg_divisors_device(75, "cuda"), 422)
    for g, wnt, name in zip(got, want, "Y Cb Cr".split()):
        assert np.array_equal(g.cpu().numpy(), wnt), name


@pytest.mark.parametrize("dims", [(16, 8), (160, 96), (200, 120), (1920, 1080)], ids=str)
@pytest.mark.parametrize("ri", [1, 4, 7])
def test_full_jpeg_stream_422(hip, po, dims, ri):
    import io
    import torch
    from PIL import Image
    from jpeg_bitstream import write_jpeg
    w, h = dims
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    uyvy = po.convert_frame("RGB", "UYVY", rgb.clip(0, 255).astype(np.uint8), w, h)
    q = 80
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=422)
    data = enc.encode(torch.from_numpy(uyvy).cuda())
    enc.close()
    (y, u, v), (ql, qc), coefs = _want_422(po, uyvy, w, h, q)
    want = write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=422)
    assert data == want, (len(data), len(want))
    img = Image.open(io.BytesIO(data))
    img.draft("YCbCr", None)
    dec = np.asarray(img)
    assert dec.shape == (h, w, 3)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((dec[..., 0].astype(float) - y.astype(float)) ** 2))
    assert psnr > 40, psnr
    # chroma really is full vertical resolution: the decoder's (upsampled) Cb at even columns tracks the source rows
    assert np.abs(dec[:, 0::2, 1].astype(int)[:, : u.shape[1]] - u.astype(int)).mean() < 2.0


def test_jpeg_encoder_rejects_unknown_subsampling(hip):
    import ctypes as C
    from ultragrid_amd import lib as L
    h = C.c_void_p()
    assert L.load().ug_hip_jpeg_encoder_create_sub(64, 64, 75, 4, 411, C.byref(h)) == -2


def _smooth_rgb(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    return rgb.clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("dims", [(8, 8), (160, 96), (203, 121), (1920, 1080)], ids=str)
@pytest.mark.parametrize("ri", [1, 8])
def test_full_jpeg_stream_444_rgb(hip, po, dims, ri):
    """RGB input, the reference module's choice for RGB frames (gpujpeg.cpp:303-305,336: 4:4:4, components stay R, G, B): stream equals
    the test writer fed with the oracle's per-component coefficients; Pillow/libjpeg decodes it as RGB."""
    import io
    import torch
    from PIL import Image
    from ultragrid_amd import lib as L
    from jpeg_bitstream import write_jpeg
    w, h = dims
    rgb = _smooth_rgb(w, h)
    q = 85
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=444)
    data = enc.encode(torch.from_numpy(rgb.ravel()).cuda(), L.PF_RGB)
    enc.close()
    ql = po.jpeg_qtable(q, 0)
    dl = po.jpeg_divisors(ql)
    bw, bh = (w + 7) // 8, (h + 7) // 8
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), dl, bw, bh) for c in range(3)]
    want = write_jpeg(w, h, ql, po.jpeg_qtable(q, 1), *coefs, restart=ri, sub=444)
    assert data == want, (len(data), len(want))
    img = Image.open(io.BytesIO(data))
    assert img.mode == "RGB" and img.size == (w, h)
    dec = np.asarray(img).astype(float)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((dec - rgb.astype(float)) ** 2))
    assert psnr > 38, psnr


@pytest.mark.parametrize("dims", [(16, 16), (50, 38), (1920, 1080)], ids=str)
def test_i420_input_equals_uyvy_420_path(hip, po, dims):
    """I420 passthrough (gpujpeg.cpp:227-236,335): feeding the planes uyvy_to_i420 produces gives the very stream the fused UYVY 4:2:0
    path produces."""
    import torch
    from ultragrid_amd import lib as L
    w, h = dims
    uyvy = synth.s2_video("UYVY", w, h)
    y, u, v = po.uyvy_to_i420(uyvy, w, h)
    planes = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
    enc = hip.JpegEncoder(w, h, 75, 3, subsampling=420)
    a = enc.encode(torch.from_numpy(uyvy).cuda())
    b = enc.encode(torch.from_numpy(planes).cuda(), L.PF_I420)
    enc.close()
    assert a == b


def test_jpeg_encoder_input_format_mismatch(hip):
    import torch
    from ultragrid_amd import lib as L
    enc = hip.JpegEncoder(64, 64, 75, 4, subsampling=444)
    with pytest.raises(RuntimeError):
        enc.encode(torch.zeros(64 * 64 * 2, dtype=torch.uint8, device="cuda"), L.PF_UYVY)
    enc.close()


@pytest.mark.parametrize("sub", [420, 422])
def test_jpeg_zrl_path(hip, po, sub):
    """Blocks whose only AC energy sits at the highest frequencies: zero runs of 16..62 -> 1..3 ZRL symbols per block (the coder's
    general 64-bit path), mixed with plain blocks in the same restart segments."""
    import io
    import torch
    from PIL import Image
    from jpeg_bitstream import write_jpeg, ZIGZAG
    w, h = 256, 64
    yy, xx = np.mgrid[0:h, 0:w]
    k = ((xx // 8) % 4 + 4)                                     # horizontal frequency 4..7 by block column
    luma = 128 + 60 * np.cos((2 * (xx % 8) + 1) * k * np.pi / 16) * np.cos((2 * (yy % 8) + 1) * 7 * np.pi / 16)
    luma[:, 128:] = 128 + 50 * np.sin(xx[:, 128:] / 9.0)        # ordinary content in the right half
    uyvy = np.empty((h, w // 2, 4), np.uint8)
    uyvy[..., 0] = 128; uyvy[..., 2] = 128
    uyvy[..., 1] = luma[:, 0::2].clip(0, 255); uyvy[..., 3] = luma[:, 1::2].clip(0, 255)
    uyvy = uyvy.ravel()
    q = 90
    planes = po.uyvy_to_i420(uyvy, w, h) if sub == 420 else po.uyvy_to_i422(uyvy, w, h)
    vy = 2 if sub == 420 else 1
    mw, mh = w // 16, h // (8 * vy)
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    cy = po.jpeg_fdct_quant_plane(planes[0], po.jpeg_divisors(ql), 2 * mw, vy * mh)
    runs = []
    for blk in cy:
        nz = np.flatnonzero(blk[1:]) + 1
        runs.append(int(np.max(np.diff(np.concatenate([[0], nz])) - 1)) if nz.size else 0)
    assert max(runs) >= 48 and sum(r > 15 for r in runs) > 50 and sum(r <= 15 for r in runs) > 50   # 3-ZRL blocks and plain blocks
    enc = hip.JpegEncoder(w, h, q, 3, subsampling=sub)
    data = enc.encode(torch.from_numpy(uyvy).cuda())
    enc.close()
    want = write_jpeg(w, h, ql, qc, cy, po.jpeg_fdct_quant_plane(planes[1], po.jpeg_divisors(qc), mw, mh),
                      po.jpeg_fdct_quant_plane(planes[2], po.jpeg_divisors(qc), mw, mh), restart=3, sub=sub)
    assert data == want
    img = Image.open(io.BytesIO(data))
    img.draft("YCbCr", None)
    assert np.abs(np.asarray(img)[..., 0].astype(int) - planes[0].astype(int)).mean() < 3


def test_jpeg_small_output_buffer_reports_needed_size(hip, po):
    """out_capacity below the stream size: UG_HIP_EINVAL and *out_len = the size it takes; with exactly that size it succeeds."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    w, h = 256, 128
    uyvy = torch.from_numpy(synth.s1_random("UYVY", w, h)).cuda()
    l = L.load()
    enc = C.c_void_p()
    assert l.ug_hip_jpeg_encoder_create_sub(w, h, 95, 4, 422, C.byref(enc)) == 0
    full = torch.empty(l.ug_hip_jpeg_encoder_max_size(enc), dtype=torch.uint8, device="cuda")
    n = C.c_size_t(0)
    st = torch.cuda.current_stream().cuda_stream
    assert l.ug_hip_jpeg_encoder_encode(enc, L.PF_UYVY, uyvy.data_ptr(), 0, full.data_ptr(), full.numel(), C.byref(n), st) == 0
    need = n.value
    ref = bytes(full[:need].cpu().numpy())
    small = torch.empty(need - 1000, dtype=torch.uint8, device="cuda")
    n2 = C.c_size_t(0)
    assert l.ug_hip_jpeg_encoder_encode(enc, L.PF_UYVY, uyvy.data_ptr(), 0, small.data_ptr(), small.numel(), C.byref(n2), st) == -1
    assert n2.value == need
    exact = torch.empty(need + 16, dtype=torch.uint8, device="cuda")[:need]
    assert l.ug_hip_jpeg_encoder_encode(enc, L.PF_UYVY, uyvy.data_ptr(), 0, exact.data_ptr(), need, C.byref(n2), st) == 0
    assert bytes(exact.cpu().numpy()) == ref
    l.ug_hip_jpeg_encoder_destroy(enc)


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [420, 422])
@pytest.mark.parametrize("dims", [(256, 64), (200, 50)])   # MCU-aligned fast path and the generic edge path
def test_batched_front_end_equals_per_frame(hip, po, sub, dims):
    """ug_hip_uyvy_to_jpeg42x_coeffs_batch (grid.z = frame): every frame of the batch == the single-frame entry point == the oracle."""
    import torch
    from ultragrid_amd import lib as L, synth
    w, h = dims
    n = 5
    frames = [synth.s2_video("UYVY", w, h, salt=i) for i in range(n)]
    fb = (frames[0].size + 15) // 16 * 16
    src = torch.zeros(n * fb, dtype=torch.uint8, device="cuda")
    for i, f in enumerate(frames):
        src[i * fb: i * fb + f.size] = torch.from_numpy(f).cuda()
    div = hip.jpeg_divisors_device(75, "cuda")
    mw = (w + 15) // 16
    mh, ybl = ((h + 15) // 16, 4) if sub == 420 else ((h + 7) // 8, 2)
    nb = mw * mh
    oy = torch.zeros((n, ybl * nb, 64), dtype=torch.int16, device="cuda")
    ocb, ocr = torch.zeros((n, nb, 64), dtype=torch.int16, device="cuda"), torch.zeros((n, nb, 64), dtype=torch.int16, device="cuda")
    rc = L.load().ug_hip_uyvy_to_jpeg42x_coeffs_batch(sub, src.data_ptr(), 0, w, h, div.data_ptr(), oy.data_ptr(), ocb.data_ptr(), ocr.data_ptr(), n, fb,
                                                      ybl * nb * 128, nb * 128, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.last_error()
    for i, f in enumerate(frames):
        sy, scb, scr = hip.uyvy_to_jpeg_coeffs(torch.from_numpy(f).cuda(), w, h, div, sub)
        assert torch.equal(oy[i], sy) and torch.equal(ocb[i], scb) and torch.equal(ocr[i], scr), i
    y, u, v = po.uyvy_to_i420(frames[2], w, h) if sub == 420 else po.uyvy_to_i422(frames[2], w, h)
    ql, qc = po.jpeg_qtable(75, 0), po.jpeg_qtable(75, 1)
    assert np.array_equal(oy[2].cpu().numpy(), po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, (2 if sub == 420 else 1) * mh))
    assert np.array_equal(ocb[2].cpu().numpy(), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh))
    assert L.load().ug_hip_uyvy_to_jpeg42x_coeffs_batch(411, src.data_ptr(), 0, w, h, div.data_ptr(), oy.data_ptr(), ocb.data_ptr(), ocr.data_ptr(), n, fb,
                                                        ybl * nb * 128, nb * 128, None) == L.EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [420, 422, 444])
def test_block_parallel_coder_equals_wave_per_segment_coder(hip, po, sub, monkeypatch):
    """The two entropy coders of the library -- one lane per block (segments of <= 64 blocks), one wave per segment (any length; forced
    with UG_JPEG_WAVE_KERNEL=1) -- must produce the same stream for every restart interval: segments that fill a wave exactly, leave
    lanes idle, end short at the end of the picture, and the lengths only the wave coder takes."""
    import torch
    w, h = 208, 88   # 13 x 6 MCUs (4:2:0): a short last segment for most intervals
    src = synth.s2_video("UYVY", w, h) if sub != 444 else synth.s1_random("RGB", w, h)
    noisy = synth.s1_random("UYVY" if sub != 444 else "RGB", w, h, salt=3)
    fmt = hip.L.PF_UYVY if sub != 444 else hip.L.PF_RGB
    for q, frame in ((75, src), (100, noisy), (20, src)):    # q = 100 on noise: > 64 bytes per block, the multi-pass variant
        dev = torch.from_numpy(frame).cuda()
        for ri in (1, 2, 3, 4, 5, 7, 10, 16, 21, 22, 40):
            monkeypatch.delenv("UG_JPEG_WAVE_KERNEL", raising=False)
            a = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
            da = a.encode(dev, fmt)
            a.close()
            monkeypatch.setenv("UG_JPEG_WAVE_KERNEL", "1")
            b = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
            db = b.encode(dev, fmt)
            b.close()
            assert da == db, (sub, q, ri, len(da), len(db))
        if q == 100:
            assert len(da) > 64 * (w // 8) * (h // 8)   # more than 64 B per luma block on average: windows overflow, several passes


# (the fused UYVY / I420 front ends take widths that are a multiple of 16 -- other widths go the two-kernel way, covered elsewhere --, packed RGB any width)
_FUSED_CASES = [(sub, dims) for dims in [(640, 88), (1040, 81), (512, 64), (48, 16), (1100, 50)] for sub in [420, 422, 444, 1420] if sub == 444 or dims[0] % 16 == 0]


@pytest.mark.gpu
@pytest.mark.parametrize("sub,dims", _FUSED_CASES)
def test_fused_encoder_equals_the_two_kernel_paths(hip, po, sub, dims, monkeypatch):
    """Round 4: for UYVY (RGB) input with a restart interval that divides the 32 (64) consecutive MCUs a workgroup takes ONE kernel does the
    forward DCT, the quantiser, the Huffman coding and the byte stuffing -- the coefficients never reach HBM.  Its stream must be the
    stream of the front end + placing coder pair (UG_JPEG_FUSED=0) and of the front end + wave-per-segment coder + compaction triple
    (UG_JPEG_WAVE_KERNEL=1): MCU rows of 40, 65, 32 and 3 MCUs (a workgroup's run of MCUs wraps from one MCU row into the next, segments
    straddle rows when the interval does not divide the row), a short last workgroup, picture heights that are no MCU multiple, low quality,
    and noise at q = 100 (blocks that overflow their private strings: the general path, several passes)."""
    import torch
    w, h = dims
    # 4:4:4 = packed RGB input, R, G, B components, 64 MCUs per workgroup (any width: 1100 = 137.5 blocks, edge blocks replicated); 4:2:x = UYVY;
    # 1420 = planar I420 input (Y, U, V planes back to back) into the 4:2:0 encoder
    planar = sub == 1420
    sub = 420 if planar else sub
    fmt, pf = ("RGB", hip.L.PF_RGB) if sub == 444 else ("UYVY", hip.L.PF_UYVY)
    src = synth.s2_video(fmt, w, h)
    noisy = synth.s1_random(fmt, w, h, salt=9)
    if planar:
        pf = hip.L.PF_I420
        src, noisy = (np.concatenate([p_.ravel() for p_ in po.uyvy_to_i420(x, w, h)]) for x in (src, noisy))
    for q, frame in ((75, src), (100, noisy), (20, src), (92, noisy)):
        dev = torch.from_numpy(frame).cuda()
        for ri in (1, 2, 4, 8, 16, 32, 64):
            if sub != 444 and ri == 64:
                continue                # (beyond the 32 MCUs of a 4:2:x workgroup: the two-kernel path, covered by the other tests)
            out = {}
            for tag, env in (("fused", {}), ("two", {"UG_JPEG_FUSED": "0"}), ("wave", {"UG_JPEG_WAVE_KERNEL": "1"}),
                             ("look", {"UG_JPEG_LOOKBACK": "1"}), ("twolook", {"UG_JPEG_FUSED": "0", "UG_JPEG_LOOKBACK": "1"}), ("ticket", {"UG_JPEG_LOOKBACK": "1", "UG_JPEG_TICKET": "1"}), ("force", {"UG_JPEG_LOOKBACK": "0"}),
                             ("flat", {"UG_JPEG_FLAT": "1"}), ("flat0", {"UG_JPEG_FLAT": "0"}), ("flatticket", {"UG_JPEG_FLAT": "1", "UG_JPEG_TICKET": "1"}), ("flattwo", {"UG_JPEG_FLAT": "1", "UG_JPEG_FUSED": "0"})):
                for k in ("UG_JPEG_FUSED", "UG_JPEG_WAVE_KERNEL", "UG_JPEG_LOOKBACK", "UG_JPEG_TICKET", "UG_JPEG_FLAT"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                e = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
                out[tag] = e.encode(dev, pf)
                out[tag + "2"] = e.encode_batch(torch.stack([dev, dev]), pf)[1]   # the same object again, two frames: the two-launch placement
                e.close()
            assert out["fused"] == out["wave"], (sub, dims, q, ri, len(out["fused"]), len(out["wave"]))
            assert out["two"] == out["wave"] and out["fused2"] == out["wave"] and out["two2"] == out["wave"], (sub, dims, q, ri)
            # the placement in one launch (decoupled look-back; the default for one-frame calls, UG_JPEG_LOOKBACK=1 for all) -- what the two-launch
            # placement (slots + gather; the default from two frames up, UG_JPEG_LOOKBACK=0 for all) falls back to when a workgroup's bytes exceed
            # its slot --, with the workgroup index from blockIdx and from a start-order ticket
            # round 5: the flat form of the one-launch placement for one-frame calls (every workgroup sums all its predecessors' byte counts itself)
            for tag in ("look", "look2", "twolook", "twolook2", "ticket", "ticket2", "force", "force2", "flat", "flat2", "flat0", "flatticket", "flattwo"):
                assert out[tag] == out["wave"], (tag, sub, dims, q, ri)
    for k in ("UG_JPEG_FUSED", "UG_JPEG_WAVE_KERNEL", "UG_JPEG_LOOKBACK", "UG_JPEG_TICKET", "UG_JPEG_FLAT"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.gpu
def test_block_parallel_coder_full_4k_frame(hip, po):
    """configs[3] size: a whole 3840x2160 frame (8100 segments at restart 4), both coders, identical streams, decodable."""
    import io
    import os
    import torch
    from PIL import Image
    w, h = 3840, 2160
    dev = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
    a = hip.JpegEncoder(w, h, 75, 4)
    da = a.encode(dev)
    a.close()
    os.environ["UG_JPEG_WAVE_KERNEL"] = "1"
    try:
        b = hip.JpegEncoder(w, h, 75, 4)
        db = b.encode(dev)
        b.close()
    finally:
        del os.environ["UG_JPEG_WAVE_KERNEL"]
    assert da == db
    assert Image.open(io.BytesIO(da)).size == (w, h)
    for flat in ("1", "0"):                       # round 5: both forms of the one-launch placement at the size they were built for (1 013 workgroups), 8K too
        os.environ["UG_JPEG_FLAT"] = flat
        try:
            c = hip.JpegEncoder(w, h, 75, 4)
            assert [c.encode(dev) for _ in range(3)] == [da] * 3, flat
            c.close()
        finally:
            del os.environ["UG_JPEG_FLAT"]
    w8, h8 = 7680, 4320
    dev8 = torch.from_numpy(synth.s2_video("UYVY", w8, h8)).cuda()
    outs = []
    for flat in ("1", "0"):
        os.environ["UG_JPEG_FLAT"] = flat
        try:
            c = hip.JpegEncoder(w8, h8, 75, 4, subsampling=422)
            outs.append(c.encode(dev8))
            c.close()
        finally:
            del os.environ["UG_JPEG_FLAT"]
    assert outs[0] == outs[1] and Image.open(io.BytesIO(outs[0])).size == (w8, h8)


@pytest.mark.parametrize("sub,fmt,ri", [(420, "UYVY", 4), (422, "UYVY", 4), (422, "UYVY", 40), (444, "RGB", 8), (420, "I420", 2)])
def test_encode_batch_equals_single_frame_calls(hip, po, sub, fmt, ri):
    """ug_hip_jpeg_encoder_encode_batch: n frames, grid.z / grid.y = frame, one synchronisation -- every stream byte-equal
    to the single-frame call's, in order; a second batch on the same object (the alternating chunk totals), a smaller one, then a single
    frame again; and a batch whose streams do not fit reports the needed sizes."""
    import ctypes as C
    import torch
    from ultragrid_amd import lib as L
    w, h, n = 208, 120, 5
    rng = np.random.default_rng(sub + ri)
    frames = []
    for f in range(n):
        rgb = np.clip(_smooth_rgb(w, h).astype(np.int32) + rng.integers(-12 * f, 12 * f + 1, (h, w, 3)), 0, 255).astype(np.uint8)
        if fmt == "RGB":
            frames.append(rgb.ravel())
        else:
            uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
            frames.append(uyvy if fmt == "UYVY" else np.concatenate([p.ravel() for p in po.uyvy_to_i420(uyvy, w, h)]))
    dev = torch.from_numpy(np.stack(frames)).cuda()
    pf = {"UYVY": L.PF_UYVY, "RGB": L.PF_RGB, "I420": L.PF_I420}[fmt]
    single = hip.JpegEncoder(w, h, 80, ri, subsampling=sub)
    want = [single.encode(dev[f], pf) for f in range(n)]
    assert len({len(x) for x in want}) > 1          # the frames really differ
    enc = hip.JpegEncoder(w, h, 80, ri, subsampling=sub)
    assert enc.encode_batch(dev, pf) == want
    assert enc.encode_batch(dev.flip(0).contiguous(), pf) == want[::-1]
    assert enc.encode_batch(dev[1:3].contiguous(), pf) == want[1:3]
    assert enc.encode(dev[4], pf) == want[4]
    # too small an output: a batch reports per frame -- the call succeeds, out_len[f] = the needed size (> capacity) for the streams that do
    # not fit, and the streams that do fit are complete (one oversize frame must not cost the whole batch)
    lens = (C.c_size_t * n)()
    cap = 1024
    out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    rc = L.load().ug_hip_jpeg_encoder_encode_batch(enc._h, pf, n, dev.data_ptr(), 0, dev.shape[1], out.data_ptr(), cap, cap, lens, None)
    assert rc == L.SUCCESS and [lens[f] for f in range(n)] == [len(x) for x in want] and min(lens) > cap
    sizes = sorted(len(x) for x in want)
    cap = (sizes[len(sizes) // 2] + 15) // 16 * 16      # the smaller streams fit, the larger ones do not
    out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    rc = L.load().ug_hip_jpeg_encoder_encode_batch(enc._h, pf, n, dev.data_ptr(), 0, dev.shape[1], out.data_ptr(), cap, cap, lens, None)
    assert rc == L.SUCCESS and [lens[f] for f in range(n)] == [len(x) for x in want]
    assert any(l > cap for l in lens) and any(l <= cap for l in lens)
    for f in range(n):
        if lens[f] <= cap:
            assert bytes(out[f, : lens[f]].cpu().numpy()) == want[f]
    one = C.c_size_t(0)
    big = max(range(n), key=lambda f: len(want[f]))
    assert L.load().ug_hip_jpeg_encoder_encode(enc._h, pf, dev[big].data_ptr(), 0, out.data_ptr(), cap, C.byref(one), None) == L.EINVAL and one.value == len(want[big])
    assert L.load().ug_hip_jpeg_encoder_encode_batch(enc._h, pf, 17, dev.data_ptr(), 0, dev.shape[1], out.data_ptr(), cap, cap, lens, None) == L.EINVAL
    enc.close()
    single.close()


def test_batches_of_varying_size_on_one_encoder(hip, po):
    """batch(4), batch(2), a single frame, batch(4) again on ONE encoder at a size with more than 256 restart segments
    (1080p, restart 2: 4 080) -- per-call state of the stream placement (round 3: chunk totals left behind by a larger batch; now the
    look-back status words, which carry the call's generation) must not leak from one call into the next."""
    import torch
    w, h = 1920, 1088
    base = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
    dev = torch.stack([torch.roll(base, 3840 * 29 * f) for f in range(4)])
    for sub, ri in ((420, 2), (422, 4), (420, 64)):
        single = hip.JpegEncoder(w, h, 75, ri, subsampling=sub)
        want = [single.encode(dev[f]) for f in range(4)]
        single.close()
        assert len(set(want)) == 4
        enc = hip.JpegEncoder(w, h, 75, ri, subsampling=sub)
        assert enc.encode_batch(dev) == want
        assert enc.encode_batch(dev[2:].contiguous()) == want[2:]
        assert enc.encode(dev[1]) == want[1]
        assert enc.encode_batch(dev) == want
        assert enc.encode_batch(dev[1:].contiguous()) == want[1:]
        assert enc.encode_batch(dev.flip(0).contiguous()) == want[::-1]
        enc.close()


def test_encode_batch_full_4k(hip, po):
    """8 distinct 4K frames in one batch == 8 single calls (the size the throughput figure is quoted on)"""
    import torch
    from ultragrid_amd import lib as L
    w, h, n = 3840, 2160, 8
    base = torch.from_numpy(synth.s2_video("UYVY", w, h)).cuda()
    dev = torch.stack([torch.roll(base, 7680 * 37 * f) for f in range(n)])
    enc = hip.JpegEncoder(w, h, 75, 4, subsampling=420)
    want = [enc.encode(dev[f]) for f in range(n)]
    assert enc.encode_batch(dev) == want
    enc.close()


@pytest.mark.gpu
def test_streams_equal_libjpeg_turbo_with_its_float_dct(hip, po):
    """Round 4: the PRODUCT's streams against an executable published implementation of the same arithmetic -- the image's libjpeg-turbo
    with dct_method = JDCT_FLOAT (IJG's float AAN DCT + float quantiser, tests/libjpeg_float.py) -- on the same samples, the same quality,
    the same restart interval: the entropy-coded bytes between SOS and EOI are IDENTICAL.
      * packed RGB -> R, G, B components 4:4:4 (libjpeg: JCS_RGB kept as JCS_RGB): any size, edge blocks included;
      * UYVY -> 4:2:2 / 4:2:0: libjpeg is handed the planes the reference's own converters make of the frame (uyvy_to_i422 / uyvy_to_i420:
        the oracle's, pinned to the compiled reference) through jpeg_write_raw_data, so that only forward DCT, quantiser and entropy
        coding are compared.  Sizes whose block grid fills whole MCUs: a block that lies wholly outside the picture is content-free
        padding, which libjpeg fills with "dummy" blocks (DC of the neighbour, no AC) and this encoder with the DCT of the replicated
        edge -- both legal, neither visible; blocks that merely straddle the edge are compared.
    libjpeg-turbo is not the library UltraGrid links (libgpujpeg, unobtainable here): towards the reference the stage stays unpinned;
    this pins it to IJG's float DCT, the formulation oracle/jpeg_oracle.c restates."""
    import torch

    import libjpeg_float as ljf
    lj = ljf.load()
    if lj is None:
        pytest.skip("no libjpeg-turbo with the IJG v8 API in this image")
    rng = np.random.default_rng(5)
    for (w, h) in ((64, 48), (1100, 50), (203, 33), (1920, 64), (8, 8), (3840, 40)):
        for q, ri in ((75, 4), (50, 8), (92, 1), (100, 16), (20, 64)):
            img = synth.frame("S2", "RGB", w, h).reshape(h, w, 3) if q != 100 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            e = hip.JpegEncoder(w, h, q, ri, subsampling=444)
            got = e.encode(torch.from_numpy(np.ascontiguousarray(img).ravel()).cuda(), hip.L.PF_RGB)
            e.close()
            assert ljf.scan_bytes(got) == ljf.scan_bytes(ljf.compress(lj, img, q, restart=ri)), ("rgb", w, h, q, ri)
    for sub, sizes in ((422, ((64, 48), (1040, 81), (1920, 1080), (28, 5), (3840, 24))), (420, ((64, 48), (1040, 80), (1036, 90), (48, 16), (1920, 1072), (3840, 32)))):
        for (w, h) in sizes:
            for q, ri in ((75, 4), (92, 1), (50, 8), (100, 2)):
                uyvy = synth.s2_video("UYVY", w, h) if q != 100 else synth.s1_random("UYVY", w, h, salt=q)
                if sub == 420:
                    y, u, v = po.uyvy_to_i420(uyvy, w, h)
                else:
                    a = uyvy.reshape(h, 2 * w)
                    y, u, v = a[:, 1::2], a[:, 0::4], a[:, 2::4]
                e = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
                got = e.encode(torch.from_numpy(uyvy).cuda(), hip.L.PF_UYVY)
                e.close()
                want = ljf.compress_planes(lj, y, u, v, w, h, sub, q, restart=ri)
                assert ljf.scan_bytes(got) == ljf.scan_bytes(want), (sub, w, h, q, ri)


@pytest.mark.gpu
def test_streams_equal_the_frozen_libjpeg_turbo_fixture(hip, po):
    """the libjpeg-turbo pin as a COMMITTED fixture (tests/golden/libjpeg_float.npz: what libjpeg-turbo 2.1.2 with JDCT_FLOAT
    produced, generator beside it) -- compared ALWAYS, whether or not the library is in the image:
      * ug_hip_jpeg_fdct_quant_plane's unquantised coefficients == jpeg_fdct_float's output BITS for all 512 committed blocks;
      * its quantised coefficients of the grey planes, entropy-coded by the test writer == libjpeg-turbo's scan bytes;
      * the encoder's streams for packed RGB (4:4:4) and UYVY (4:2:2, 4:2:0) carry libjpeg-turbo's scan bytes -- one-frame call and batch call.
    Towards libgpujpeg (what UltraGrid links; unobtainable here) the stage stays "parity unpinned"."""
    import torch

    import jpeg_bitstream as jb
    import libjpeg_float as ljf
    from test_oracle_jpeg import libjpeg_fixture
    g, meta = libjpeg_fixture()
    blocks = g["fdct_blocks"]
    plane = torch.from_numpy(np.ascontiguousarray(blocks.transpose(1, 0, 2).reshape(8, 512 * 8))).cuda()
    div = torch.from_numpy(po.jpeg_divisors(po.jpeg_qtable(75, 0))).cuda()
    _, coef = hip.jpeg_fdct_quant_plane(plane, div, want_coef=True)
    assert np.array_equal(coef.cpu().numpy().reshape(512, 64).view(np.uint32), g["fdct_out"])
    dcl, acl = jb._codes(*jb.DC_L), jb._codes(*jb.AC_L)
    seen = {}
    for j, c in enumerate(meta["cases"]):
        src, w, h, q, ri = g["in_" + c["input"]], c["w"], c["h"], c["q"], c["ri"]
        want = g[f"scan_{j}"].tobytes()
        if c["kind"] == "grey":
            d = torch.from_numpy(po.jpeg_divisors(po.jpeg_qtable(q, 0))).cuda()
            out = hip.jpeg_fdct_quant_plane(torch.from_numpy(np.ascontiguousarray(src)).cuda(), d).cpu().numpy()
            bw, pred = jb._Bits(), 0
            for u in range(out.shape[0]):
                pred = jb._block(bw, out[u], pred, dcl, acl)
            bw.flush()
            assert bytes(bw.buf) == want, c
        else:
            sub, pf = (444, hip.L.PF_RGB) if c["kind"] == "rgb" else (int(c["kind"]), hip.L.PF_UYVY)
            dev = torch.from_numpy(np.ascontiguousarray(src).ravel()).cuda()
            e = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
            assert ljf.scan_bytes(e.encode(dev, pf)) == want, c
            two = e.encode_batch(torch.stack([dev, dev]), pf)
            assert ljf.scan_bytes(two[0]) == want and two[1] == two[0], c
            e.close()
        seen[c["kind"]] = seen.get(c["kind"], 0) + 1
    assert set(seen) == {"grey", "rgb", "422", "420"} and sum(seen.values()) >= 80, seen
