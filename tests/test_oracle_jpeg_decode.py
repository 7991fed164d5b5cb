"""CPU: the JPEG decode restatement (oracle/jpeg_decode_oracle.c) pinned to libjpeg-turbo (Pillow): component planes bit for bit wherever
libjpeg hands out untouched samples -- all planes of 4:4:4 YCbCr and R,G,B streams, the luma plane of 4:2:2 / 4:2:0 streams --, with and
without restart intervals, optimised Huffman tables, greyscale; plus the streams of this repository's own test writer."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def picture(w, h, seed=1, noise=4.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    return (rgb + rng.normal(0, noise, rgb.shape)).clip(0, 255).astype(np.uint8)


def pil_jpeg(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


CASES = [dict(quality=90, subsampling=0), dict(quality=75, subsampling=1), dict(quality=50, subsampling=2), dict(quality=95, subsampling=2, restart_marker_blocks=3),
         dict(quality=80, subsampling=0, optimize=True), dict(quality=30, subsampling=1, restart_marker_rows=1), dict(quality=100, subsampling=0, restart_marker_blocks=1)]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
@pytest.mark.parametrize("size", [(200, 120), (17, 9), (64, 64)])
def test_planes_equal_libjpeg(po, kw, size):
    w, h = size
    data = pil_jpeg(picture(w, h, seed=w), **kw)
    info, crop, _ = po.jpeg_decode_planes(data)
    assert (info["width"], info["height"], info["components"]) == (w, h, 3)
    ref = Image.open(io.BytesIO(data))
    ref.draft("YCbCr", None)           # raw YCbCr, no colour conversion
    ref = np.asarray(ref)
    assert np.array_equal(crop[0], ref[..., 0])
    if kw["subsampling"] == 0:          # no upsampling between libjpeg's planes and what Pillow returns
        assert np.array_equal(crop[1], ref[..., 1]) and np.array_equal(crop[2], ref[..., 2])


def test_greyscale_and_noise(po):
    g = picture(150, 70)[..., 1]
    data = pil_jpeg(g, quality=85)
    info, crop, _ = po.jpeg_decode_planes(data)
    assert info["components"] == 1 and np.array_equal(crop[0], np.asarray(Image.open(io.BytesIO(data))))
    noise = np.random.default_rng(5).integers(0, 256, (64, 96, 3), dtype=np.uint8)    # long codes, every coefficient present
    data = pil_jpeg(noise, quality=100, subsampling=0)
    _, crop, _ = po.jpeg_decode_planes(data)
    ref = Image.open(io.BytesIO(data))
    ref.draft("YCbCr", None)
    assert all(np.array_equal(crop[c], np.asarray(ref)[..., c]) for c in range(3))


@pytest.mark.parametrize("sub", [420, 422, 444])
def test_streams_of_the_repository_writer(po, sub):
    """The byte streams the encoder is held to (tests/jpeg_bitstream.py over the oracle's coefficients: JFIF YCbCr 4:2:x, Adobe R,G,B 4:4:4, restart
    intervals): decoded planes equal libjpeg's; for the R,G,B stream that is every plane of the picture Pillow returns."""
    from jpeg_bitstream import write_jpeg
    w, h = 176, 80
    rgb = picture(w, h, seed=3)
    ql, qc = po.jpeg_qtable(80, 0), po.jpeg_qtable(80, 1)
    if sub == 444:
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
    else:
        uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        mw = (w + 15) // 16
        y, u, v = po.uyvy_to_i420(uyvy, w, h) if sub == 420 else po.uyvy_to_i422(uyvy, w, h)
        mh, vy = ((h + 15) // 16, 2) if sub == 420 else ((h + 7) // 8, 1)
        coefs = [po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, vy * mh), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh),
                 po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh)]
    data = write_jpeg(w, h, ql, qc, *coefs, restart=3, sub=sub)
    info, crop, _ = po.jpeg_decode_planes(data)
    assert info["restart"] == 3 and (info["adobe"] == 0) == (sub == 444)
    # the entropy decoder alone hands back the very coefficients the stream was written from (bench.py's parity_check of the encoder leg rests on this)
    _, coded = po.jpeg_decode_coeffs(data)
    assert all(np.array_equal(a, b) for a, b in zip(coded, coefs))
    ref = Image.open(io.BytesIO(data))
    if sub == 444:
        assert ref.mode == "RGB"
        assert all(np.array_equal(crop[c], np.asarray(ref)[..., c]) for c in range(3))
        assert 10 * np.log10(255.0 ** 2 / np.mean((np.asarray(ref).astype(float) - rgb) ** 2)) > 34
    else:
        ref.draft("YCbCr", None)
        assert np.array_equal(crop[0], np.asarray(ref)[..., 0])


@pytest.mark.parametrize("ri", [0, 5])
def test_one_scan_per_component_equals_libjpeg(po, ri):
    """non-interleaved scans (GPUJPEG's default layout for RGB, gpujpeg.cpp:302): every plane equals what libjpeg decodes"""
    from jpeg_bitstream import write_jpeg_noninterleaved
    w, h = 150, 70
    rgb = picture(w, h, seed=4)
    ql = po.jpeg_qtable(85, 0)
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
    data = write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri)
    info, crop, _ = po.jpeg_decode_planes(data)
    assert info["scans"] == 3 and info["restart"] == ri
    assert all(np.array_equal(a, b) for a, b in zip(po.jpeg_decode_coeffs(data)[1], coefs))
    ref = np.asarray(Image.open(io.BytesIO(data)))
    assert all(np.array_equal(crop[c], ref[..., c]) for c in range(3))


def test_rejects_what_is_not_baseline(po):
    b = io.BytesIO()
    Image.fromarray(picture(64, 64)).save(b, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        po.jpeg_decode_planes(b.getvalue())
    with pytest.raises(ValueError):
        po.jpeg_decode_planes(b"\x00\x01\x02\x03")


def test_random_streams_equal_libjpeg(po):
    """300 random valid streams (tools/find_oracle_vs_libjpeg.py runs the same search over thousands): sizes from 1x1, qualities 1..100, all
    samplings and greyscale, optimised tables, restart intervals, smooth to pure-noise content."""
    checked = 0
    for seed in range(300):
        rng = np.random.default_rng(seed)
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        noise = [0.0, 2.0, 10.0, 60.0, 200.0][int(rng.integers(5))]
        img = (picture(w, h, seed=seed, noise=0.0).astype(float) + rng.normal(0, noise, (h, w, 3))).clip(0, 255).astype(np.uint8)
        grey = rng.random() < 0.1
        kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.integers(2)))
        if not grey:
            kw["subsampling"] = int(rng.integers(3))
        r = int(rng.integers(4))
        if r == 1:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        elif r == 2:
            kw["restart_marker_rows"] = int(rng.integers(1, 3))
        try:
            data = pil_jpeg(img[..., 1] if grey else img, **kw)
        except OSError:
            continue
        _, crop, _ = po.jpeg_decode_planes(data)
        ref = Image.open(io.BytesIO(data))
        if grey:
            assert np.array_equal(crop[0], np.asarray(ref)), (seed, kw)
        else:
            ref.draft("YCbCr", None)
            ref = np.asarray(ref)
            assert np.array_equal(crop[0], ref[..., 0]), (seed, kw)
            if kw["subsampling"] == 0:
                assert np.array_equal(crop[1], ref[..., 1]) and np.array_equal(crop[2], ref[..., 2]), (seed, kw)
        checked += 1
    assert checked > 250
