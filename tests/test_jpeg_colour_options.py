"""The rest of the reference JPEG module's encoder options (src/video_compress/gpujpeg.cpp:303-305,396-405; VERDICT r5 "What's missing" #4): the colour
stage (color_space_internal = Y601 / Y601full / Y709 / RGB) and one scan per component (the default layout for RGB input).  Both sit on the
FDCT / quantiser that is UNPINNED towards libgpujpeg; what is checked is (CPU) the oracle's colour stage against the published BT.601 / BT.709
definitions in fp64 and the writer's new layouts against libjpeg (Pillow), (GPU) product == oracle bit for bit, streams == the writer's byte for
byte, and every stream decoded by libjpeg and by the product's own decoder."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(__file__))
from ultragrid_amd import synth

RGB, Y601, Y601FULL, Y709 = 1, 2, 3, 4


def _fp64_map(cs_in, cs_out):
    """the same definitions, evaluated independently in numpy float64: (3, 4) affine map on code values"""
    def from_rgb(cs):
        if cs == RGB:
            return np.hstack([np.eye(3), np.zeros((3, 1))])
        kr, kb = (0.2126, 0.0722) if cs == Y709 else (0.299, 0.114)
        kg = 1 - kr - kb
        ys, cs_, y0 = (1.0, 1.0, 0.0) if cs == Y601FULL else (219 / 255, 224 / 255, 16.0)
        m = np.array([[kr, kg, kb], [-kr / (2 * (1 - kb)), -kg / (2 * (1 - kb)), 0.5], [0.5, -kg / (2 * (1 - kr)), -kb / (2 * (1 - kr))]])
        return np.hstack([m * np.array([[ys], [cs_], [cs_]]), np.array([[y0], [128.0], [128.0]])])
    a, b = from_rgb(cs_in), from_rgb(cs_out)
    ainv = np.linalg.inv(a[:, :3])
    to_rgb = np.hstack([ainv, -ainv @ a[:, 3:]])
    return np.hstack([b[:, :3] @ to_rgb[:, :3], b[:, :3] @ to_rgb[:, 3:] + b[:, 3:]])


def test_colour_matrices_are_the_published_ones(po):
    jfif = po.jpeg_colour_matrix(RGB, Y601FULL)
    assert np.allclose(jfif, [[0.299, 0.587, 0.114, 0], [-0.168736, -0.331264, 0.5, 128], [0.5, -0.418688, -0.081312, 128]], atol=2e-6)     # ITU-T T.871 (JFIF)
    bt709 = po.jpeg_colour_matrix(RGB, Y709)
    assert np.allclose(bt709[0], [0.2126 * 219 / 255, 0.7152 * 219 / 255, 0.0722 * 219 / 255, 16], atol=1e-6)                               # BT.709, 8-bit limited range
    assert np.allclose(po.jpeg_colour_matrix(Y709, RGB), [[1.164384, 0, 1.792741, -248.101], [1.164384, -0.213249, -0.532909, 76.878], [1.164384, 2.112402, 0, -289.018]], atol=2e-3)
    for a in (RGB, Y601, Y601FULL, Y709):
        for b in (RGB, Y601, Y601FULL, Y709):
            assert np.allclose(po.jpeg_colour_matrix(a, b), _fp64_map(a, b), rtol=0, atol=3e-5), (a, b)     # float32 of the same numbers
        assert np.allclose(po.jpeg_colour_matrix(a, a), np.hstack([np.eye(3), np.zeros((3, 1))]), atol=1e-5)
    with pytest.raises(ValueError):
        po.jpeg_colour_matrix(0, 3)


@pytest.mark.parametrize("cs", [Y601, Y601FULL, Y709])
def test_colour_stage_within_one_code_value_of_fp64(po, cs):
    rng = np.random.default_rng(cs)
    w, h = 64, 48
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = po.jpeg_colour_convert("RGB", RGB, cs, rgb, w, h).reshape(h, w, 3).astype(int)
    m = _fp64_map(RGB, cs)
    want = np.clip(rgb.astype(np.float64) @ m[:, :3].T + m[:, 3], 0, 255)
    assert np.abs(got - want).max() <= 0.5 + 1e-3                                      # correctly rounded up to the float32 coefficients: well inside 1 code value
    # and back: RGB -> Y'CbCr -> RGB is the picture again up to the two roundings (and the clipping of colours outside the limited range's gamut)
    back = po.jpeg_colour_convert("RGB", cs, RGB, got.astype(np.uint8), w, h).reshape(h, w, 3).astype(int)
    assert np.abs(back - rgb).mean() < 1.0
    # 4:2:2: BT.709 limited -> BT.601: luma per pixel, chroma = the mean of the pair's two results
    uyvy = synth.s2_video("UYVY", w, h, salt=cs)
    if cs != Y709:
        out = po.jpeg_colour_convert("UYVY", Y709, cs, uyvy, w, h).reshape(h, w // 2, 4).astype(int)
        src = uyvy.reshape(h, w // 2, 4).astype(np.float64)
        m = _fp64_map(Y709, cs)
        px = lambda y: np.stack([y, src[..., 0], src[..., 2]], -1) @ m[:, :3].T + m[:, 3]
        p0, p1 = px(src[..., 1]), px(src[..., 3])
        want = np.clip(np.stack([(p0[..., 1] + p1[..., 1]) / 2, p0[..., 0], (p0[..., 2] + p1[..., 2]) / 2, p1[..., 0]], -1), 0, 255)
        assert np.abs(out - want).max() <= 0.5 + 2e-3
    else:
        assert np.array_equal(po.jpeg_colour_convert("UYVY", Y709, Y709, uyvy, w, h), uyvy)


def _coefs444(po, planes, ql, qc, w, h):
    bw, bh = (w + 7) // 8, (h + 7) // 8
    return [po.jpeg_fdct_quant_plane(np.ascontiguousarray(planes[..., c]), po.jpeg_divisors(ql if c == 0 or qc is None else qc), bw, bh) for c in range(3)]


@pytest.mark.parametrize("ri", [0, 5])
def test_writer_layouts_decode_with_libjpeg(po, ri):
    """the two new layouts of the test writer -- 4:4:4 Y'CbCr interleaved, and Y'CbCr with one scan per component -- are streams libjpeg reads, and
    (JFIF = BT.601 full range) reads back as the picture"""
    from jpeg_bitstream import write_jpeg, write_jpeg_noninterleaved
    w, h = 150, 70
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    ycc = po.jpeg_colour_convert("RGB", RGB, Y601FULL, rgb, w, h).reshape(h, w, 3)
    ql, qc = po.jpeg_qtable(90, 0), po.jpeg_qtable(90, 1)
    coefs = _coefs444(po, ycc, ql, qc, w, h)
    for data in (write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=444, ycc=True), write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri, qt_chroma=qc)):
        img = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(float)
        assert 10 * np.log10(255.0 ** 2 / np.mean((img - rgb) ** 2)) > 36
        info, crop, _ = po.jpeg_decode_planes(data)
        assert info["scans"] in (1, 3) and info["adobe"] == -1
        _, coded = po.jpeg_decode_coeffs(data)
        assert all(np.array_equal(a, b) for a, b in zip(coded, coefs))


# ------------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cs_in,cs_out", [(RGB, Y601), (RGB, Y601FULL), (RGB, Y709), (Y709, RGB), (Y601FULL, RGB)])
def test_gpu_colour_stage_rgb_bit_exact(hip, po, cs_in, cs_out):
    import torch
    from ultragrid_amd import lib as L
    assert all(np.array_equal(hip.jpeg_colour_matrix(a, b), po.jpeg_colour_matrix(a, b)) for a in range(1, 5) for b in range(1, 5))
    for (w, h) in [(1, 1), (5, 3), (257, 9), (1920, 16)]:
        src = np.random.default_rng(w).integers(0, 256, 3 * w * h, dtype=np.uint8)
        got = hip.jpeg_colour_convert(L.PF_RGB, cs_in, cs_out, torch.from_numpy(src).cuda(), w, h).cpu().numpy()
        assert np.array_equal(got, po.jpeg_colour_convert("RGB", cs_in, cs_out, src, w, h)), (w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("cs_out", [Y601, Y601FULL, Y709])
def test_gpu_colour_stage_uyvy_bit_exact(hip, po, cs_out):
    import torch
    from ultragrid_amd import lib as L
    for (w, h) in [(2, 1), (6, 3), (258, 9), (1920, 16)]:
        src = np.random.default_rng(w).integers(0, 256, 2 * w * h, dtype=np.uint8)
        got = hip.jpeg_colour_convert(L.PF_UYVY, Y709, cs_out, torch.from_numpy(src).cuda(), w, h).cpu().numpy()
        assert np.array_equal(got, po.jpeg_colour_convert("UYVY", Y709, cs_out, src, w, h)), (w, h)


def _rgb_picture(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    return (base + np.random.default_rng(w * h).normal(0, 3, base.shape)).clip(0, 255).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(16, 8), (150, 70), (640, 360), (1921, 1081)], ids=str)
@pytest.mark.parametrize("ri", [1, 4, 7, 64])
def test_gpu_rgb_one_scan_per_component(hip, po, dims, ri):
    """RGB 4:4:4, the reference's default layout for RGB input (gpujpeg.cpp:303: interleaved = 0): the stream == the test writer's, byte for byte;
    libjpeg and the product's own decoder read it back to the oracle's planes; one frame and a batch give the same bytes"""
    import torch
    from jpeg_bitstream import write_jpeg_noninterleaved
    from ultragrid_amd import lib as L
    w, h = dims
    rgb = _rgb_picture(w, h)
    q = 80
    ql = po.jpeg_qtable(q, 0)
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=444, flags=L.JPEG_NONINTERLEAVED)
    dev = torch.from_numpy(rgb.ravel()).cuda()
    data = enc.encode(dev, L.PF_RGB)
    want = write_jpeg_noninterleaved(w, h, ql, _coefs444(po, rgb, ql, None, w, h), restart=ri)
    assert data == want
    if w <= 640:
        two = enc.encode_batch(torch.stack([dev, dev.flip(0)]), L.PF_RGB)
        assert two[0] == data and two[1] != data and len(two) == 2
    enc.close()
    img = np.asarray(Image.open(io.BytesIO(data)))
    _, crop, _ = po.jpeg_decode_planes(data)
    assert all(np.array_equal(crop[c], img[..., c]) for c in range(3))
    assert 10 * np.log10(255.0 ** 2 / np.mean((img.astype(float) - rgb) ** 2)) > 33
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_RGB).cpu().numpy().reshape(h, w, 3)
    dec.close()
    assert np.array_equal(got, img)


@pytest.mark.gpu
@pytest.mark.parametrize("ri", [4, 0, 257, 300, 5000])
@pytest.mark.parametrize("cs", [0, Y601FULL])
def test_gpu_one_scan_per_component_long_restart_intervals(hip, po, ri, cs):
    """restart intervals of more than 256 blocks, and none at all, in a stream of one scan per component: the wave-per-segment coder, scan after scan
    (the block coder takes segments of up to 256 blocks); bytes == the test writer's; a batch gives the same"""
    import torch
    from jpeg_bitstream import write_jpeg_noninterleaved
    from ultragrid_amd import lib as L
    w, h, q = 640, 360, 85
    rgb = _rgb_picture(w, h)
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=444, internal_cs=cs, flags=L.JPEG_NONINTERLEAVED)
    dev = torch.from_numpy(rgb.ravel()).cuda()
    data = enc.encode(dev, L.PF_RGB)
    two = enc.encode_batch(torch.stack([dev.flip(0), dev]), L.PF_RGB)
    enc.close()
    comps = po.jpeg_colour_convert("RGB", RGB, cs, rgb, w, h).reshape(h, w, 3) if cs else rgb
    want = write_jpeg_noninterleaved(w, h, ql, _coefs444(po, comps, ql, qc if cs else None, w, h), restart=ri, qt_chroma=qc if cs else None)
    assert data == want and two[1] == data and two[0] != data
    img = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(float)
    assert 10 * np.log10(255.0 ** 2 / np.mean((img - rgb) ** 2)) > 33
    # a buffer the stream does not fit is reported with the size it takes, and nothing behind the buffer is written (the guard bytes stay)
    import ctypes as C
    l = L.load()
    h_enc = C.c_void_p()
    assert l.ug_hip_jpeg_encoder_create_ex(w, h, q, ri, 444, cs, L.JPEG_NONINTERLEAVED, C.byref(h_enc)) == 0
    for short in (len(data) // 5, len(data) - 3000):
        buf = torch.full((short + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        n = C.c_size_t(0)
        assert l.ug_hip_jpeg_encoder_encode(h_enc, L.PF_RGB, dev.data_ptr(), 0, buf.data_ptr(), short, C.byref(n), torch.cuda.current_stream().cuda_stream) == L.EINVAL
        assert n.value >= short and bool((buf[short:] == 0xA5).all())
    l.ug_hip_jpeg_encoder_destroy(h_enc)


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [Y601, Y601FULL, Y709])
@pytest.mark.parametrize("nonint", [False, True])
def test_gpu_rgb_coded_as_ycbcr(hip, po, cs, nonint):
    """RGB input with color_space_internal = Y601 / Y601full / Y709 (gpujpeg.cpp:398-403), as one interleaved scan (`:interleaved`) and as the default
    three: stream == the writer's over the oracle's colour stage + FDCT, byte for byte; Y601full is JFIF, so libjpeg gives the picture back"""
    import torch
    from jpeg_bitstream import write_jpeg, write_jpeg_noninterleaved
    from ultragrid_amd import lib as L
    w, h, q, ri = 322, 166, 85, 4
    rgb = _rgb_picture(w, h)
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=444, internal_cs=cs, flags=L.JPEG_NONINTERLEAVED if nonint else 0)
    data = enc.encode(torch.from_numpy(rgb.ravel()).cuda(), L.PF_RGB)
    enc.close()
    ycc = po.jpeg_colour_convert("RGB", RGB, cs, rgb, w, h).reshape(h, w, 3)
    coefs = _coefs444(po, ycc, ql, qc, w, h)
    want = write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri, qt_chroma=qc) if nonint else write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=444, ycc=True)
    assert data == want
    img = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(float)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((img - rgb) ** 2))
    assert psnr > (33 if cs == Y601FULL else 15), psnr          # (the limited-range spaces are not what a JFIF reader assumes: readable, not faithful)
    _, crop, _ = po.jpeg_decode_planes(data)
    assert np.abs(np.stack(crop, -1).astype(int) - ycc).mean() < 2.0


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [Y601, Y601FULL])
@pytest.mark.parametrize("sub", [422, 420])
def test_gpu_uyvy_coded_as_bt601(hip, po, cs, sub):
    """UYVY input (BT.709 limited range) with Y601 / Y601full: the stream of the converted samples -- == the plain encoder fed the oracle's conversion"""
    import torch
    from ultragrid_amd import lib as L
    w, h = 640, 368
    uyvy = synth.s2_video("UYVY", w, h, salt=9)
    enc = hip.JpegEncoder(w, h, 80, 4, subsampling=sub, internal_cs=cs)
    data = enc.encode(torch.from_numpy(uyvy).cuda(), L.PF_UYVY)
    enc.close()
    plain = hip.JpegEncoder(w, h, 80, 4, subsampling=sub)
    want = plain.encode(torch.from_numpy(po.jpeg_colour_convert("UYVY", Y709, cs, uyvy, w, h)).cuda(), L.PF_UYVY)
    same = plain.encode(torch.from_numpy(uyvy).cuda(), L.PF_UYVY)
    plain.close()
    assert data == want and data != same
    e709 = hip.JpegEncoder(w, h, 80, 4, subsampling=sub, internal_cs=Y709)       # what the samples are already: nothing to convert
    assert e709.encode(torch.from_numpy(uyvy).cuda(), L.PF_UYVY) == same
    e709.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [0, RGB, Y601, Y601FULL, Y709])
@pytest.mark.parametrize("dims", [(640, 368), (322, 166), (75, 33)], ids=str)
def test_gpu_uyvy_coded_444(hip, po, cs, dims):
    """subsampling=444 on a 4:2:2 source (UG_JPEG_INPUT_UYVY; gpujpeg.cpp:297-302 with UYVY input): every pixel with its pair's chroma, coded as the
    samples are (BT.709 Y'CbCr), as BT.601, or as R, G, B -- the stream == the 4:4:4 encoder for RGB input fed the oracle's 3 B/px picture"""
    import torch
    from jpeg_bitstream import write_jpeg
    from ultragrid_amd import lib as L
    w, h = dims
    q, ri = 80, 4
    uyvy = synth.s2_video("UYVY", w + (w & 1), h, salt=11).reshape(h, -1)[:, :(w + 1) // 2 * 4].copy().ravel()
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=444, internal_cs=cs, flags=L.JPEG_INPUT_UYVY)
    dev = torch.from_numpy(uyvy).cuda()
    data = enc.encode(dev, L.PF_UYVY)
    if w % 2 == 0:
        two = enc.encode_batch(torch.stack([dev, dev.flip(0)]), L.PF_UYVY)
        assert two[0] == data and two[1] != data
    with pytest.raises(RuntimeError):
        enc.encode(torch.zeros(3 * w * h, dtype=torch.uint8).cuda(), L.PF_RGB)          # it was created for UYVY
    enc.close()
    pic = po.jpeg_colour_convert("UYVY444", Y709, cs or Y709, uyvy, w, h).reshape(h, w, 3)
    if cs in (0, Y709):
        pairs = uyvy.reshape(h, -1, 4)
        assert np.array_equal(pic[:, 0::2, 0], pairs[:, :, 1]) and np.array_equal(pic[:, 1::2, 0], pairs[:, :w // 2, 3])
        assert np.array_equal(pic[:, 0::2, 1], pairs[:, :, 0]) and np.array_equal(pic[:, 1::2, 2], pairs[:, :w // 2, 2])
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    if cs == RGB:
        want = write_jpeg(w, h, ql, qc, *_coefs444(po, pic, ql, None, w, h), restart=ri, sub=444)     # (the header carries table 1 too; R, G, B all use table 0)
        if w % 2 == 0:    # the picture a viewer gets == the reference's own UYVY -> RGB conversion of the source, within the codec's loss
            img = np.asarray(Image.open(io.BytesIO(data))).astype(float)
            ref = po.convert_frame("UYVY", "RGB", uyvy, w, h).reshape(h, w, 3).astype(float)
            assert 10 * np.log10(255.0 ** 2 / np.mean((img - ref) ** 2)) > 28
    else:
        want = write_jpeg(w, h, ql, qc, *_coefs444(po, pic, ql, qc, w, h), restart=ri, sub=444, ycc=True)
    assert data == want
    plain = hip.JpegEncoder(w, h, q, ri, subsampling=444, internal_cs=0 if cs == RGB else cs or Y709)
    same = plain.encode(torch.from_numpy(pic.ravel()).cuda(), L.PF_RGB)
    plain.close()
    if cs == RGB:
        assert same == data          # (for the Y'CbCr spaces the plain encoder would convert the picture once more: nothing to compare)
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_RGB).cpu().numpy().reshape(h, w, 3)
    dec.close()
    _, crop, _ = po.jpeg_decode_planes(data)
    assert np.abs(np.stack(crop, -1).astype(int) - pic).mean() < 2.5 and got.shape == (h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("sub,dims", [(422, (64, 32)), (420, (640, 368)), (422, (1920, 1080)), (444, (322, 166)), (420, (150, 71))], ids=str)
def test_gpu_no_restart_intervals(hip, po, sub, dims):
    """restart_interval 0 (`-c jpeg:restart=0`, gpujpeg.cpp:345): ONE entropy-coded segment, no DRI, no RSTn -- == the test writer's stream without
    restart intervals; libjpeg, the decode oracle and the product's decoder (one lane for the whole scan) read it; a batch gives the same bytes"""
    import torch
    from jpeg_bitstream import write_jpeg
    from ultragrid_amd import lib as L
    w, h = dims
    q = 80
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    dl, dc = po.jpeg_divisors(ql), po.jpeg_divisors(qc)
    enc = hip.JpegEncoder(w, h, q, 0, subsampling=sub)
    if sub == 444:
        rgb = _rgb_picture(w, h)
        dev, fmt = torch.from_numpy(rgb.ravel()).cuda(), L.PF_RGB
        want = write_jpeg(w, h, ql, qc, *_coefs444(po, rgb, ql, None, w, h), restart=0, sub=444)
    else:
        uyvy = synth.s2_video("UYVY", w, h, salt=2)
        dev, fmt = torch.from_numpy(uyvy).cuda(), L.PF_UYVY
        y, u, v = po.uyvy_to_i422(uyvy, w, h) if sub == 422 else po.uyvy_to_i420(uyvy, w, h)
        mw, mh = (w + 15) // 16, (h + 7) // 8 if sub == 422 else (h + 15) // 16
        want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, dl, 2 * mw, mh * (1 if sub == 422 else 2)), po.jpeg_fdct_quant_plane(u, dc, mw, mh),
                          po.jpeg_fdct_quant_plane(v, dc, mw, mh), restart=0, sub=sub)
    data = enc.encode(dev, fmt)
    assert data == want and b"\xff\xdd" not in data[:700]
    if w <= 640:
        two = enc.encode_batch(torch.stack([dev, dev.flip(0)]), fmt)
        assert two[0] == data and two[1] != data
    enc.close()
    Image.open(io.BytesIO(data)).load()
    info = hip.jpeg_read_info(data) if hasattr(hip, "jpeg_read_info") else None
    assert info is None or info["restart"] == 0
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_RGB if sub == 444 else L.PF_UYVY).cpu().numpy()
    dec.close()
    _, crop, _ = po.jpeg_decode_planes(data)
    if sub == 444:
        assert np.array_equal(got.reshape(h, w, 3), np.stack(crop, -1))
    else:
        assert np.array_equal(got.reshape(h, w, 2)[..., 1], crop[0])


@pytest.mark.gpu
@pytest.mark.parametrize("sub,dims,ri", [(422, (1920, 1080), 65), (422, (640, 368), 100), (420, (1280, 720), 1000), (444, (642, 366), 300), (420, (150, 71), 44), (422, (3840, 2160), 2000)], ids=str)
def test_gpu_long_restart_intervals(hip, po, sub, dims, ri):
    """restart intervals of more than 256 blocks per segment (the block coder's limit): a lane per block codes into the per-segment buffers of the wave-per-segment coder
    (bit positions by a segmented prefix sum), that coder's compaction kernel assembles the stream -- == the test writer's, == the old kernel's (UG_JPEG_WAVE_KERNEL=1 in a
    process of its own), one frame and a batch"""
    import subprocess
    import sys
    import torch
    from jpeg_bitstream import write_jpeg
    from ultragrid_amd import lib as L
    w, h = dims
    q = 80
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    dl, dc = po.jpeg_divisors(ql), po.jpeg_divisors(qc)
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
    if sub == 444:
        rgb = _rgb_picture(w, h)
        dev, fmt = torch.from_numpy(rgb.ravel()).cuda(), L.PF_RGB
        want = write_jpeg(w, h, ql, qc, *_coefs444(po, rgb, ql, None, w, h), restart=ri, sub=444)
    else:
        uyvy = synth.s2_video("UYVY", w, h, salt=4)
        dev, fmt = torch.from_numpy(uyvy).cuda(), L.PF_UYVY
        y, u, v = po.uyvy_to_i422(uyvy, w, h) if sub == 422 else po.uyvy_to_i420(uyvy, w, h)
        mw, mh = (w + 15) // 16, (h + 7) // 8 if sub == 422 else (h + 15) // 16
        want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, dl, 2 * mw, mh * (1 if sub == 422 else 2)), po.jpeg_fdct_quant_plane(u, dc, mw, mh),
                          po.jpeg_fdct_quant_plane(v, dc, mw, mh), restart=ri, sub=sub)
    data = enc.encode(dev, fmt)
    assert data == want
    if w <= 1920:
        two = enc.encode_batch(torch.stack([dev.flip(0), dev, dev]), fmt)
        assert two[1] == data and two[2] == data and two[0] != data
    enc.close()
    if dims == (640, 368):      # the old kernel gives the same bytes (its own process: the switch is read when an encoder is made)
        code = ("import sys, torch, numpy as np; sys.path.insert(0, %r); from ultragrid_amd import codec as hip, lib as L, synth; "
                "e = hip.JpegEncoder(640, 368, 80, 100, subsampling=422); d = e.encode(torch.from_numpy(synth.s2_video('UYVY', 640, 368, salt=4)).cuda(), L.PF_UYVY); "
                "sys.stdout.buffer.write(d)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300, env={**os.environ, "UG_JPEG_WAVE_KERNEL": "1"})
        assert r.returncode == 0 and r.stdout == data, r.stderr[-500:]


@pytest.mark.gpu
def test_gpu_create_ex_refusals(hip):
    import ctypes as C
    from ultragrid_amd import lib as L
    l = L.load()
    enc = C.c_void_p()
    assert l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, 4, 422, 0, L.JPEG_NONINTERLEAVED, C.byref(enc)) == L.EUNSUPP     # one scan per component: 4:4:4
    assert l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, 4, 420, L.JPEG_CS_RGB, 0, C.byref(enc)) == L.EUNSUPP             # a 4:2:x stream is Y'CbCr
    assert l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, -1, 444, 0, 0, C.byref(enc)) == L.EINVAL
    assert l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, 4, 444, 7, 0, C.byref(enc)) == L.EINVAL and l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, 4, 444, 0, 4, C.byref(enc)) == L.EINVAL
    assert l.ug_hip_jpeg_encoder_create_ex(64, 64, 75, 4, 422, 0, L.JPEG_INPUT_UYVY, C.byref(enc)) == L.EUNSUPP          # a 4:2:x encoder takes UYVY anyway
