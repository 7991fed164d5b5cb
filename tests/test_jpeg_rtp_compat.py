"""`-c jpeg` output against the reference's OWN RFC 2435 gate: src/transmit.c:1368 hands every JPEG frame to jpeg_get_rtp_hdr_data()
(src/utils/jpeg_reader.c:1106-1160) and sends nothing if that refuses the stream (check_rtp_compatibility :1060-1104: baseline, 8 bit,
3 interleaved components, at most two quantisation tables, sampling 2x1,1x1,1x1 -> type 0 or 2x2,1x1,1x1 -> type 1, +0x40 with restart
markers).  jpeg_reader.c is compiled as it lies into oracle/_ref/libugref.so (oracle/Makefile).

CPU: streams of tests/jpeg_bitstream.py, to which the product's stream is byte-equal (tests/test_gpu_jpeg.py::test_full_jpeg_stream_*).
GPU: the product encoder's and the `-c jpeg` module's real output."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RtpData(C.Structure):  # struct jpeg_rtp_data, src/utils/jpeg_reader.h:86-93
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("type", C.c_int), ("q", C.c_int), ("restart_interval", C.c_int),
                ("quantization_tables", C.POINTER(C.c_uint8) * 2), ("data", C.POINTER(C.c_uint8))]


def rtp_hdr(po, data: bytes):
    """(accepted, RtpData, offsets of the two table pointers and of the data pointer inside the stream)"""
    if not po.have_ref() or not hasattr(po.ref(), "jpeg_get_rtp_hdr_data"):
        pytest.skip("oracle/_ref/libugref.so (with src/utils/jpeg_reader.c) not built")
    fn = po.ref().jpeg_get_rtp_hdr_data
    fn.restype, fn.argtypes = C.c_bool, [C.c_void_p, C.c_int, C.POINTER(RtpData)]
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    d = RtpData()
    ok = bool(fn(C.addressof(buf), len(data), C.byref(d)))
    base = C.addressof(buf)
    offs = [C.cast(p, C.c_void_p).value - base if p else None for p in (d.quantization_tables[0], d.quantization_tables[1], d.data)]
    return ok, d, offs, buf


def check(po, data: bytes, w, h, sub, ri, ql, qc):
    from jpeg_bitstream import ZIGZAG
    ok, d, (t0, t1, dat), _keep = rtp_hdr(po, data)
    assert ok, "the reference's jpeg_get_rtp_hdr_data refuses the stream: transmit.c:1368 would not send it"
    assert (d.width, d.height) == (w, h)
    assert d.type == (1 if sub == 420 else 0) | (0x40 if ri else 0)         # RFC 2435 types 0 / 1, 64 / 65 with restart markers
    assert d.restart_interval == ri
    assert d.q == 255                                                       # no "quality = " comment: the tables travel in the RTP header
    # the table pointers point INTO the stream at the 64 zig-zag bytes of the luma and the (shared) chroma table
    zz = np.asarray(ZIGZAG)
    assert t0 is not None and t1 is not None
    assert np.array_equal(np.frombuffer(data, np.uint8, 64, t0), np.asarray(ql, np.uint8).ravel()[zz])
    assert np.array_equal(np.frombuffer(data, np.uint8, 64, t1), np.asarray(qc, np.uint8).ravel()[zz])
    # ... and the data pointer at the first entropy-coded byte: right behind the SOS header
    sos = data.index(b"\xff\xda")
    assert dat == sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")


def _picture(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    return rgb.clip(0, 255).astype(np.uint8)


def _coefs(po, uyvy, w, h, q, sub):
    """oracle coefficients of a UYVY frame as `-c jpeg` codes it (planes padded to whole MCUs by edge replication)"""
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    px = uyvy.reshape(h, w // 2, 4)
    y = np.stack([px[..., 1], px[..., 3]], -1).reshape(h, w)
    u, v = px[..., 0], px[..., 2]
    if sub == 420:
        a, b = u[0::2].astype(np.int32), u[1::2].astype(np.int32)
        u = ((a + b + 1) >> 1).astype(np.uint8)
        a, b = v[0::2].astype(np.int32), v[1::2].astype(np.int32)
        v = ((a + b + 1) >> 1).astype(np.uint8)
    mw, mh = (w + 15) // 16, (h + (15 if sub == 420 else 7)) // (16 if sub == 420 else 8)
    ybw, ybh = 2 * mw, (2 if sub == 420 else 1) * mh
    return ql, qc, [po.jpeg_fdct_quant_plane(np.ascontiguousarray(y), po.jpeg_divisors(ql), ybw, ybh),
                    po.jpeg_fdct_quant_plane(np.ascontiguousarray(u), po.jpeg_divisors(qc), mw, mh),
                    po.jpeg_fdct_quant_plane(np.ascontiguousarray(v), po.jpeg_divisors(qc), mw, mh)]


@pytest.mark.parametrize("sub,ri", [(422, 4), (420, 4), (422, 0), (420, 0), (420, 1), (422, 25)])
def test_writer_streams_pass_the_reference_rtp_gate(po, sub, ri):
    from jpeg_bitstream import write_jpeg
    w, h = 208, 96
    uyvy = po.convert_frame("RGB", "UYVY", _picture(w, h), w, h)
    ql, qc, coefs = _coefs(po, uyvy, w, h, 80, sub)
    data = write_jpeg(w, h, ql, qc, *coefs, restart=ri, sub=sub)
    check(po, data, w, h, sub, ri, ql, qc)


def test_rgb_444_streams_are_not_rtp_compatible_in_the_reference_either(po):
    """R,G,B 4:4:4 (what `-c jpeg` writes for RGB input, as gpujpeg.cpp:303-305 does) has no RFC 2435 type (jpeg_reader.c:1137-1146
    knows 2x1 and 2x2 luma sampling only), so the reference does not send such frames over RFC 2435 whoever wrote them.  Its reader in
    fact stops earlier: read_adobe_app14 (:803-851) takes SIX identifier bytes where the Adobe segment has five ("Adobe", then the
    16-bit version), reads the transform flag one byte late -- the 0xFF of the next marker -- and reports "Unsupported color transformation
    value '255'".  Recorded so that a change on either side is noticed."""
    from jpeg_bitstream import write_jpeg
    w, h = 64, 32
    rgb = _picture(w, h)
    ql = po.jpeg_qtable(80, 0)
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(rgb[..., c]), po.jpeg_divisors(ql), w // 8, h // 8) for c in range(3)]
    data = write_jpeg(w, h, ql, po.jpeg_qtable(80, 1), *coefs, restart=4, sub=444)
    ok, _, _, _ = rtp_hdr(po, data)
    assert not ok


@pytest.mark.gpu
@pytest.mark.parametrize("sub,ri,dims", [(422, 4, (1920, 1080)), (420, 4, (1920, 1080)), (422, 1, (200, 120)), (420, 7, (200, 120))])
def test_product_streams_pass_the_reference_rtp_gate(hip, po, sub, ri, dims):
    import torch
    w, h = dims
    uyvy = po.convert_frame("RGB", "UYVY", _picture(w, h), w, h)
    enc = hip.JpegEncoder(w, h, 80, ri, subsampling=sub)
    data = enc.encode(torch.from_numpy(uyvy).cuda())
    enc.close()
    check(po, bytes(data), w, h, sub, ri, po.jpeg_qtable(80, 0), po.jpeg_qtable(80, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,sub,ri", [("jpeg:q=80:restart=4", 422, 4), ("jpeg:q=80:restart=4:subsampling=420", 420, 4), ("jpeg", 422, None)])
def test_module_output_passes_the_reference_rtp_gate(tmp_path, po, cfg, sub, ri):
    """the frame compress_pop() returns from `-c jpeg` inside the reference's own compress framework (oracle/_ref/ug_harness)"""
    harness = os.path.join(ROOT, "oracle", "_ref", "ug_harness")
    if not os.path.exists(harness):
        pytest.skip("oracle/_ref/ug_harness not built")
    w, h = 192, 96
    uyvy = po.convert_frame("RGB", "UYVY", _picture(w, h), w, h)
    raw, jpg = tmp_path / "in.raw", tmp_path / "f.jpg"
    uyvy.tofile(raw)
    r = subprocess.run([harness, cfg, "UYVY", str(w), str(h), str(raw), str(jpg)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    data = jpg.read_bytes()
    info = hip_info(data)
    q = 80 if "q=80" in cfg else None
    ok, d, _, _ = rtp_hdr(po, data)
    assert ok and (d.width, d.height) == (w, h) and d.restart_interval == info["restart"]
    assert d.type == (1 if sub == 420 else 0) | (0x40 if info["restart"] else 0)
    if q is not None:
        check(po, data, w, h, sub, ri, po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1))


def hip_info(data):
    from ultragrid_amd import codec
    return codec.jpeg_read_info(data)
