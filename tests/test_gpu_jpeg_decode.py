"""GPU parity: the HIP JPEG decoder (through the C ABI) against oracle/jpeg_decode_oracle.c (itself pinned to libjpeg-turbo,
tests/test_oracle_jpeg_decode.py): component planes bit for bit, then every output codec as the composition the header states."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_oracle_jpeg_decode import CASES, picture, pil_jpeg  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
@pytest.mark.parametrize("size", [(200, 120), (17, 9), (640, 360)])
def test_planes_equal_the_oracle(hip, po, kw, size):
    w, h = size
    data = pil_jpeg(picture(w, h, seed=w), **kw)
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    got = dec.planes(data)
    dec.close()
    assert len(got) == 3
    for c in range(3):
        assert np.array_equal(got[c].cpu().numpy(), crop[c]), c


def _own_stream(hip, src, fmt, w, h, q, ri, sub):
    import torch
    enc = hip.JpegEncoder(w, h, q, ri, subsampling=sub)
    data = enc.encode(torch.from_numpy(src).cuda(), fmt)
    enc.close()
    return data


@pytest.mark.parametrize("sub,ri", [(422, 4), (420, 2), (420, 7), (444, 4), (422, 1), (422, 40)])
def test_round_trip_of_the_repository_encoder(hip, po, sub, ri):
    """`-c jpeg` streams (4:2:2 / 4:2:0 YCbCr, R,G,B 4:4:4; restart intervals) decode to the oracle's planes; the outputs are what the header says:
    4:2:2 -> UYVY = the planes interleaved, RGB / RGBA = UltraGrid's own UYVY -> RGB of that, I420 from 4:2:0, R,G,B streams packed directly."""
    from ultragrid_amd import lib as L, synth
    w, h = 208, 88
    rgb = picture(w, h, seed=9, noise=2.0)
    if sub == 444:
        data = _own_stream(hip, np.ascontiguousarray(rgb).ravel(), L.PF_RGB, w, h, 85, ri, 444)
    else:
        data = _own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 85, ri, sub)
    info = hip.jpeg_read_info(data)
    assert info == dict(width=w, height=h, subsampling=sub, is_rgb=sub == 444, restart=ri)
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    for c, pl in enumerate(dec.planes(data)):
        assert np.array_equal(pl.cpu().numpy(), crop[c]), c
    if sub == 444:
        want_rgb = np.stack(crop, -1)
        assert np.array_equal(dec.decode(data, L.PF_RGB).cpu().numpy().reshape(h, w, 3), want_rgb)
        for sh in ((0, 8, 16), (16, 8, 0)):
            assert np.array_equal(dec.decode(data, L.PF_RGBA, sh).cpu().numpy(), po.convert_frame("RGB", "RGBA", want_rgb, w, h, sh))
        assert np.array_equal(dec.decode(data, L.PF_UYVY).cpu().numpy(), po.convert_frame("RGB", "UYVY", want_rgb, w, h))
        assert 10 * np.log10(255.0 ** 2 / np.mean((want_rgb.astype(float) - rgb) ** 2)) > 36
    else:
        y, u, v = crop
        want_uyvy = po.planar_to_uyvy(y, u, v, w, h, chroma=sub)
        assert np.array_equal(dec.decode(data, L.PF_UYVY).cpu().numpy(), want_uyvy)
        assert np.array_equal(dec.decode(data, L.PF_RGB).cpu().numpy(), po.convert_frame("UYVY", "RGB", want_uyvy, w, h))
        assert np.array_equal(dec.decode(data, L.PF_RGBA, (16, 8, 0)).cpu().numpy(), po.convert_frame("UYVY", "RGBA", want_uyvy, w, h, (16, 8, 0)))
        if sub == 420:
            assert np.array_equal(dec.decode(data, L.PF_I420).cpu().numpy(), np.concatenate([p.ravel() for p in crop]))
        src_uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        assert 10 * np.log10(255.0 ** 2 / np.mean((want_uyvy.reshape(h, -1)[:, 1::2].astype(float) - src_uyvy.reshape(h, -1)[:, 1::2]) ** 2)) > 38
    dec.close()


def test_yuv444_to_uyvy_averages_chroma_pairs(hip, po):
    """4:4:4 YCbCr -> UYVY: luma as it is, chroma of a pixel pair = (a + b) / 2 (as UltraGrid's own 4:4:4 -> 4:2:2 converters do)."""
    from ultragrid_amd import lib as L
    w, h = 100, 36
    data = pil_jpeg(picture(w, h, seed=2), quality=90, subsampling=0)
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_UYVY).cpu().numpy().reshape(h, w // 2, 4)
    y, u, v = (c.astype(int) for c in crop)
    assert np.array_equal(got[..., 1], y[:, 0::2]) and np.array_equal(got[..., 3], y[:, 1::2])
    assert np.array_equal(got[..., 0], (u[:, 0::2] + u[:, 1::2]) // 2) and np.array_equal(got[..., 2], (v[:, 0::2] + v[:, 1::2]) // 2)
    dec.close()


@pytest.mark.parametrize("ri", [0, 5])
def test_one_scan_per_component(hip, po, ri):
    """The layout GPUJPEG writes for RGB input by default (gpujpeg.cpp:302, interleaved = 0): three non-interleaved scans, each over its own
    block grid with its own restart segments (tests/jpeg_bitstream.py::write_jpeg_noninterleaved; libjpeg decodes it to the same planes,
    tests/test_oracle_jpeg_decode.py)."""
    from jpeg_bitstream import write_jpeg_noninterleaved
    from ultragrid_amd import lib as L
    w, h = 150, 70
    rgb = picture(w, h, seed=4)
    ql = po.jpeg_qtable(85, 0)
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
    data = write_jpeg_noninterleaved(w, h, ql, coefs, restart=ri)
    info, crop, _ = po.jpeg_decode_planes(data)
    assert info["scans"] == 3
    dec = hip.JpegDecoder()
    for c, pl in enumerate(dec.planes(data)):
        assert np.array_equal(pl.cpu().numpy(), crop[c]), c
    assert np.array_equal(dec.decode(data, L.PF_RGB).cpu().numpy().reshape(h, w, 3), np.stack(crop, -1))
    dec.close()


def test_frame_header_or_tables_between_scans_are_refused(hip, po):
    """ADVICE r2 (high, low): in a stream with one scan per component the second header parse walks past the first SOS.  A frame header there
    (new width / height / component count after the caller's size check, under scans already recorded) is refused, as libjpeg refuses it; so
    is a Huffman or quantisation table that an earlier scan used and a later segment redefines (one table set is uploaded per frame).  And
    the size the destination was made for is checked by the decoder itself."""
    import ctypes as C
    import torch
    from jpeg_bitstream import write_jpeg_noninterleaved
    from ultragrid_amd import lib as L
    w, h = 150, 70
    rgb = picture(w, h, seed=4)
    ql = po.jpeg_qtable(85, 0)
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
    data = bytearray(write_jpeg_noninterleaved(w, h, ql, coefs, restart=5))
    dec = hip.JpegDecoder()
    good = dec.decode(bytes(data), L.PF_RGB).cpu().numpy()
    sof = data.index(b"\xff\xc0")
    sof_seg = bytes(data[sof:sof + 2 + int.from_bytes(data[sof + 2:sof + 4], "big")])
    big = bytearray(sof_seg)
    big[5:7], big[7:9] = (4096).to_bytes(2, "big"), (4096).to_bytes(2, "big")    # 4096 x 4096 where the caller allocated 150 x 70
    dht = data.index(b"\xff\xc4")
    dht_seg = bytes(data[dht:dht + 2 + int.from_bytes(data[dht + 2:dht + 4], "big")])
    dqt = data.index(b"\xff\xdb")
    dqt_seg = bytes(data[dqt:dqt + 2 + int.from_bytes(data[dqt + 2:dqt + 4], "big")])
    second_sos = data.index(b"\xff\xda", data.index(b"\xff\xda") + 2)
    lib = L.load()
    dst = torch.full((w * h * 3 + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    for extra in (bytes(big), sof_seg, dht_seg, dqt_seg):
        bad = bytes(data[:second_sos]) + extra + bytes(data[second_sos:])
        rc = lib.ug_hip_jpeg_decoder_decode(dec._h, bad, len(bad), L.PF_RGB, dst.data_ptr(), 0, 0, 8, 16, None)
        torch.cuda.synchronize()
        assert rc == L.EUNSUPP, extra[:2]
        assert (dst[w * h * 3:] == 0xEE).all()
    # the size check of the decoder itself: a destination made for another picture size
    assert lib.ug_hip_jpeg_decoder_decode_sized(dec._h, bytes(data), len(data), w, h + 2, L.PF_RGB, dst.data_ptr(), 0, 0, 8, 16, None) == L.EINVAL
    assert lib.ug_hip_jpeg_decoder_decode_sized(dec._h, bytes(data), len(data), w, h, L.PF_RGB, dst.data_ptr(), 0, 0, 8, 16, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dst[: w * h * 3].cpu().numpy(), good)
    dec.close()


def test_greyscale_planes_ignore_the_sampling_factors(hip, po):
    """a one-component scan is not interleaved whatever its factors say (T.81 A.2.2): 2x2 decodes to the plane 1x1 decodes to."""
    rng = np.random.default_rng(5)
    b = io.BytesIO()
    Image.fromarray((rng.random((37, 50)) * 255).astype(np.uint8), "L").save(b, "JPEG", quality=85)
    d = bytearray(b.getvalue())
    want = np.asarray(Image.open(io.BytesIO(bytes(d))))
    dec = hip.JpegDecoder()
    one = dec.planes(bytes(d))[0].cpu().numpy()
    sof = d.index(b"\xff\xc0")
    d[sof + 11] = 0x22
    two = dec.planes(bytes(d))[0].cpu().numpy()
    dec.close()
    assert one.shape == (37, 50) and np.array_equal(one, two) and np.array_equal(one, want)


@pytest.mark.parametrize("dims,rst", [((50, 37), 0), ((640, 360), 3), ((1281, 721), 0)], ids=str)
def test_greyscale_streams_decode_to_every_output(hip, po, dims, rst):
    """A one-component stream (GPUJPEG_U8, video_decompress/gpujpeg.c:239-241) is a Y'CbCr picture whose chroma sits at its zero: UYVY = the plane with
    U = V = 128, I420 = the plane + two planes of 128, RGB / RGBA = that UYVY through the reference's conversion."""
    from ultragrid_amd import lib as L
    w, h = dims
    yy, xx = np.mgrid[0:h, 0:w]
    grey = (128 + 90 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + np.random.default_rng(w).normal(0, 4, (h, w))).clip(0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(grey, "L").save(b, "JPEG", quality=88, **({"restart_marker_blocks": rst} if rst else {}))
    data = b.getvalue()
    want = np.asarray(Image.open(io.BytesIO(data)))
    dec = hip.JpegDecoder()
    if w % 2 == 0:
        uyvy = dec.decode(data, L.PF_UYVY).cpu().numpy().reshape(h, w, 2)
        assert np.array_equal(uyvy[..., 1], want) and (uyvy[..., 0] == 128).all()
        assert np.array_equal(dec.decode(data, L.PF_RGB).cpu().numpy(), po.convert_frame("UYVY", "RGB", uyvy.ravel(), w, h))
        assert np.array_equal(dec.decode(data, L.PF_RGBA).cpu().numpy(), po.convert_frame("UYVY", "RGBA", uyvy.ravel(), w, h))
    i420 = dec.decode(data, L.PF_I420).cpu().numpy()
    dec.close()
    assert np.array_equal(i420[:w * h].reshape(h, w), want) and (i420[w * h:] == 128).all() and i420.size == w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)


def test_full_4k_frame_and_rejections(hip, po):
    import torch
    from ultragrid_amd import lib as L, synth
    w, h = 3840, 2160
    src = synth.s2_video("UYVY", w, h)
    data = _own_stream(hip, src, L.PF_UYVY, w, h, 75, 4, 422)
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_UYVY).cpu().numpy()
    assert np.array_equal(got, po.planar_to_uyvy(*crop, w, h, chroma=422))
    assert 10 * np.log10(255.0 ** 2 / np.mean((got.astype(float) - src) ** 2)) > 36
    b = io.BytesIO()
    Image.fromarray(picture(64, 64)).save(b, "JPEG", progressive=True)
    with pytest.raises(L.UgHipError) as e:
        dec.decode(b.getvalue(), L.PF_UYVY)
    assert e.value.rc == L.EUNSUPP
    with pytest.raises(L.UgHipError):
        dec.decode(data[: len(data) // 2][:100], L.PF_UYVY)      # headers cut off
    with pytest.raises(L.UgHipError):
        dec.decode(data, L.PF_V210)                              # no such output
    # a stream that lost its tail decodes what arrived (the rest: zero coefficients = mid grey), no crash
    cut = dec.decode(data[: len(data) // 2] + b"\xff\xd9", L.PF_UYVY).cpu().numpy()
    assert np.array_equal(cut[: w * 2 * 400], got[: w * 2 * 400])
    dec.close()


def _damage(data: bytes, rng, kind: str) -> bytes:
    """One kind of damage to the entropy-coded part of a stream (headers stay intact)."""
    b = bytearray(data)
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    end = len(b) - 2
    rst = [i for i in range(start, end) if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7]
    if kind == "cut":                      # the tail is lost, EOI appended
        at = int(rng.integers(start + 1, end))
        return bytes(b[:at]) + b"\xff\xd9"
    if kind == "cut_raw":                  # the tail is lost, nothing appended
        return bytes(b[: int(rng.integers(start + 1, end))])
    if kind == "drop_rst" and rst:         # a restart marker disappears: one segment fewer
        i = rst[int(rng.integers(len(rst)))]
        return bytes(b[:i] + b[i + 2:])
    if kind == "extra_rst":                # a restart marker too many
        i = int(rng.integers(start + 1, end))
        return bytes(b[:i] + b"\xff\xd5" + b[i:])
    if kind == "marker":                   # some other marker in the data
        i = int(rng.integers(start + 1, end))
        return bytes(b[:i] + b"\xff\xc4" + b[i:])
    for _ in range(int(rng.integers(1, 6))):   # "bytes": substitutions, with the values that matter to the syntax among them
        b[int(rng.integers(start, end))] = int(rng.choice([0x00, 0xFF, 0xD3, 0xC4, 0x5A, int(rng.integers(256))]))
    return bytes(b)


@pytest.mark.parametrize("kind", ["cut", "cut_raw", "drop_rst", "extra_rst", "marker", "bytes"])
def test_damaged_streams_decode_like_the_oracle(hip, po, kind):
    """Damage in the entropy-coded data never faults and gives the planes the sequential decoder gives: a segment ends at the first marker
    (zero bits from there), segment k starts behind the k-th restart marker whatever else stands in between, missing segments are empty."""
    from ultragrid_amd import lib as L
    w, h = 208, 88
    rgb = picture(w, h, seed=11, noise=3.0)
    data = _own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 85, 3, 422)
    rng = np.random.default_rng(sum(kind.encode()))     # the same damage in every run
    dec = hip.JpegDecoder()
    for trial in range(60):
        bad = _damage(data, rng, kind)
        try:
            info, crop, _ = po.jpeg_decode_planes(bad)
        except Exception:
            continue                        # the oracle refuses the stream: nothing to compare
        got = dec.planes(bad)
        for c in range(3):
            assert np.array_equal(got[c].cpu().numpy(), crop[c]), (kind, trial, c)
    # and the decoder is still good afterwards
    _, crop, _ = po.jpeg_decode_planes(data)
    for c, pl in enumerate(dec.planes(data)):
        assert np.array_equal(pl.cpu().numpy(), crop[c])
    dec.close()


# ---- scans without restart intervals: the self-synchronising parallel decode (jpeg_decode.hip, pass 2b; taken from 4 KiB of scan data up) ----
@pytest.mark.parametrize("rows", [0, 1, 9])
@pytest.mark.parametrize("mode,sub,dims,q,opt", [("RGB", 2, (1920, 1080), 75, False), ("RGB", 1, (1281, 723), 90, True), ("RGB", 0, (640, 360), 95, False), ("L", 0, (1000, 700), 85, True),
                                                  ("RGB", 2, (3840, 2160), 60, False), ("RGB", 0, (333, 129), 100, False)], ids=str)
def test_streams_without_restart_intervals(hip, po, mode, sub, dims, q, opt, rows):
    """another sender's stream (libjpeg: no restart markers -- or one per MCU row, or per 9 rows, as FFmpeg's slice threads write them; optimised tables or not): long
    segments, decoded by a lane per 1024 bits -- the planes are the oracle's = libjpeg's"""
    w, h = dims
    rgb = picture(w, h, seed=w, noise=6.0)
    b = io.BytesIO()
    Image.fromarray(rgb if mode == "RGB" else rgb[..., 1], mode).save(b, "JPEG", quality=q, optimize=opt, **({"subsampling": sub} if mode == "RGB" else {}),
                                                                      **({"restart_marker_rows": rows} if rows else {}))
    data = b.getvalue()
    assert (b"\xff\xdd" in data[:1200]) == bool(rows) and len(data) > 8192
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    for rep in range(2):        # (twice: the second call reuses the work buffers of the first)
        got = dec.planes(data)
        assert len(got) == len(crop) and all(np.array_equal(g.cpu().numpy(), c) for g, c in zip(got, crop)), rep
    dec.close()
    if mode == "L":
        assert np.array_equal(crop[0], np.asarray(Image.open(io.BytesIO(data))))


@pytest.mark.parametrize("nonint", [False, True])
def test_own_streams_without_restart_intervals(hip, po, nonint):
    """`-c jpeg:restart=0` streams, one interleaved scan and one scan per component (three one-segment scans): decoded in parallel, equal to the oracle and to the picture"""
    import torch
    from ultragrid_amd import lib as L
    w, h = 1280, 720
    rgb = picture(w, h, seed=3, noise=4.0)
    enc = hip.JpegEncoder(w, h, 85, 0, subsampling=444, flags=L.JPEG_NONINTERLEAVED if nonint else 0)
    data = enc.encode(torch.from_numpy(rgb.ravel()).cuda(), L.PF_RGB)
    enc.close()
    _, crop, _ = po.jpeg_decode_planes(data)
    dec = hip.JpegDecoder()
    got = dec.decode(data, L.PF_RGB).cpu().numpy().reshape(h, w, 3)
    dec.close()
    assert np.array_equal(got, np.stack(crop, -1))
    assert 10 * np.log10(255.0 ** 2 / np.mean((got.astype(float) - rgb) ** 2)) > 34


@pytest.mark.parametrize("kind,rows", [(k, r) for r in (0, 6) for k in ("cut", "cut_raw", "drop_rst", "extra_rst", "marker", "bytes", "tail") if r or k != "drop_rst"])
def test_damaged_streams_without_restart_intervals(hip, po, kind, rows):
    """Damage to a stream of one segment (rows = 0) or of a few long ones (a restart interval per 6 MCU rows: FFmpeg's slices): whatever the bits say, the parallel
    decode is the sequential decoder's chain of states (bit flips resynchronise or not -- the fixed point is the same); data that ends early, a missing or an extra
    marker send the scan the sequential way (zero bits to the end of the segment); bytes behind the last block are ignored"""
    w, h = 640, 360
    b = io.BytesIO()
    Image.fromarray(picture(w, h, seed=9, noise=5.0)).save(b, "JPEG", quality=88, subsampling=1, **({"restart_marker_rows": rows} if rows else {}))
    data = b.getvalue()
    rng = np.random.default_rng(sum(kind.encode()))
    dec = hip.JpegDecoder()
    compared = 0
    for trial in range(40):
        bad = data[:-2] + bytes(rng.integers(0, 255, int(rng.integers(1, 3000)), dtype=np.uint8).tolist()).replace(b"\xff", b"\xfe") + b"\xff\xd9" if kind == "tail" else _damage(data, rng, kind)
        try:
            _, crop, _ = po.jpeg_decode_planes(bad)
        except Exception:
            continue
        got = dec.planes(bad)
        for c in range(3):
            assert np.array_equal(got[c].cpu().numpy(), crop[c]), (kind, trial, c)
        compared += 1
    assert compared >= (0 if kind == "marker" else 20)      # (a table marker in the data: the oracle refuses most such streams)
    _, crop, _ = po.jpeg_decode_planes(data)
    assert all(np.array_equal(pl.cpu().numpy(), crop[c]) for c, pl in enumerate(dec.planes(data)))
    dec.close()


def test_mutated_headers_never_fault(hip, po):
    """Headers damaged at random (tables, frame and scan parameters, restart interval, lengths): every stream is either refused or decoded to
    something, the process survives, and the decoder still decodes a good stream correctly afterwards."""
    from ultragrid_amd import lib as L
    w, h = 208, 88
    rgb = picture(w, h, seed=5, noise=3.0)
    data = _own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 85, 3, 422)
    sos = data.index(b"\xff\xda")
    hdr_end = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    rng = np.random.default_rng(77)
    dec = hip.JpegDecoder()
    decoded = refused = 0
    for _ in range(400):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, hdr_end))] = int(rng.choice([0, 1, 2, 3, 0x11, 0x21, 0x22, 0x7F, 0xFF, int(rng.integers(256))]))
        try:
            dec.planes(bytes(b))
            decoded += 1
        except L.UgHipError:
            refused += 1
    assert decoded > 0 and refused > 0
    _, crop, _ = po.jpeg_decode_planes(data)
    for c, pl in enumerate(dec.planes(data)):
        assert np.array_equal(pl.cpu().numpy(), crop[c])
    dec.close()


def test_decoder_memory_is_flat_over_many_frames(hip, po):
    """A receiver decodes for hours: device memory in use does not grow while frames of alternating size and sampling keep coming (the work
    buffers grow to the largest frame once and are reused), and closing the decoder gives everything back."""
    import torch
    from ultragrid_amd import lib as L
    streams = []
    for (w, h, sub) in ((208, 88, 422), (640, 360, 420), (320, 200, 422)):
        rgb = picture(w, h, seed=w, noise=3.0)
        streams.append(_own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 80, 4, sub))
    torch.cuda.synchronize()
    free_at_start = torch.cuda.mem_get_info()[0]
    dec = hip.JpegDecoder()
    for i in range(30):                      # reach the steady state
        dec.decode(streams[i % 3], L.PF_UYVY)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free_steady = torch.cuda.mem_get_info()[0]
    for i in range(600):
        dec.decode(streams[i % 3], L.PF_UYVY)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert abs(torch.cuda.mem_get_info()[0] - free_steady) <= 4 << 20      # the allocator's granularity, not growth
    dec.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert free_at_start - torch.cuda.mem_get_info()[0] <= 8 << 20


@pytest.mark.parametrize("name", ["jpeg_damaged_a.jpg", "jpeg_damaged_b.jpg", "jpeg_damaged_c.jpg"])
def test_damaged_stream_fixtures(hip, po, name):
    """Streams found by the random search above on which an earlier version of the decoder and the oracle disagreed: substituted bytes that
    produce a bit pattern no Huffman code matches (the decoder then has to give up after 17 bits, as the MAXCODE walk and libjpeg do)."""
    bad = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), "rb").read()
    _, crop, _ = po.jpeg_decode_planes(bad)
    dec = hip.JpegDecoder()
    for c, pl in enumerate(dec.planes(bad)):
        assert np.array_equal(pl.cpu().numpy(), crop[c]), c
    dec.close()


def test_decoders_in_parallel_host_threads(hip, po):
    """One decoder object and one HIP stream per thread, four threads decoding different streams at once (ctypes drops the GIL during the
    calls): every result equals the oracle's -- the library keeps no shared mutable state between decoder objects."""
    import threading
    import torch
    from ultragrid_amd import lib as L
    jobs = []
    for k, (w, h, sub, ri) in enumerate(((208, 88, 422, 3), (320, 176, 420, 2), (160, 120, 422, 1), (640, 360, 420, 5))):
        rgb = picture(w, h, seed=20 + k, noise=3.0)
        data = _own_stream(hip, po.convert_frame("RGB", "UYVY", rgb, w, h), L.PF_UYVY, w, h, 85, ri, sub)
        _, crop, _ = po.jpeg_decode_planes(data)
        jobs.append((data, po.planar_to_uyvy(*crop, w, h, chroma=sub), w, h))
    errors = []

    def worker(job):
        data, want, w, h = job
        try:
            lib = L.load()
            st = torch.cuda.Stream()
            dec = hip.JpegDecoder()
            dst = torch.empty(2 * w * h, dtype=torch.uint8, device="cuda")
            for _ in range(60):
                rc = lib.ug_hip_jpeg_decoder_decode(dec._h, data, len(data), L.PF_UYVY, dst.data_ptr(), 0, 0, 8, 16, st.cuda_stream)
                assert rc == 0, L.last_error()
                st.synchronize()
                assert np.array_equal(dst.cpu().numpy(), want)
            dec.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
