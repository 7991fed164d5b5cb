"""Drop-in boundary: UltraGrid's own compress framework + registry (compiled from the reference into
oracle/_ref/ug_harness together with our module object) drives `-c dxt` end to end.
The harness binary is built in the container (needs /root/reference headers) and travels to the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from ultragrid_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ug_harness")
needs_harness = pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ug_harness not built (needs /root/reference)")


def _run(args, **kw):
    return subprocess.run([HARNESS] + [str(a) for a in args], capture_output=True, text=True, timeout=30, **kw)


@needs_harness
def test_module_registers_as_dxt():
    r = _run(["list"])
    assert r.returncode == 0
    assert "dxt" in r.stdout.split()  # REGISTER_MODULE(dxt, ...) -> lib_common.cpp registry


@needs_harness
def test_init_conventions(tmp_path):
    raw = tmp_path / "in.raw"
    np.zeros(64 * 16 * 2, np.uint8).tofile(raw)
    r = _run(["dxt:help", "UYVY", 64, 16, raw, tmp_path / "o.bin"])
    assert "usage" in r.stdout and "rc=1" in r.stderr            # INIT_NOERR -> compress_init returns 1 (video_compress.cpp:277-279)
    r = _run(["dxt:bogus", "UYVY", 64, 16, raw, tmp_path / "o.bin"])
    assert r.returncode == 2 and "unknown option" in (r.stdout + r.stderr)  # NULL -> error (video_compress.cpp:271-276)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["UYVY", "v210", "RGB", "RGBA", "YUYV", "BGR"])
@pytest.mark.parametrize("cfg", ["dxt:DXT5", "dxt:DXT1", "dxt"])
def test_compress_frame_through_reference_framework(tmp_path, po, codec, cfg):
    w, h = 192, 64
    src = synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=5)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    src.tofile(raw)
    r = _run([cfg, codec, w, h, raw, out])
    assert r.returncode == 0, r.stdout + r.stderr
    oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1   # default = DXT1 (cuda_dxt.cpp:104)
    assert ("DXT5" if cfg.endswith("DXT5") else "DXT1") in r.stdout
    if codec == "YUYV":
        want = po.dxt_encode(po.IN_UYVY, oid, po.convert_frame("YUYV", "UYVY", src, w, h), w, h)
    elif codec == "BGR":
        want = po.dxt_encode(po.IN_RGB, oid, po.convert_frame("BGR", "RGB", src, w, h), w, h)
    else:
        want = po.dxt_encode({"UYVY": po.IN_UYVY, "v210": po.IN_V210, "RGB": po.IN_RGB, "RGBA": po.IN_RGBA}[codec], oid, src, w, h)
    got = np.fromfile(out, np.uint8)
    assert np.array_equal(got, want)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["UYVY", "YUYV", "v210", "RGB"])
def test_dxt1_yuv_through_reference_framework(tmp_path, po, codec):
    """-c dxt:DXT1_YUV (the RTDXT option, dxt_glsl.cpp:104-110,233-234): DXT1 blocks over the Y,Cb,Cr samples of the UYVY form of the
    frame (everything else is converted to UYVY with the pixfmt_conv.c arithmetic first); output codec DXT1_YUV."""
    w, h = 192, 64
    src = synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=9)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    src.tofile(raw)
    r = _run(["dxt:DXT1_YUV", codec, w, h, raw, out])
    assert r.returncode == 0 and "DXT1_YUV" in r.stdout, r.stdout + r.stderr
    uyvy = src if codec == "UYVY" else po.convert_frame(codec, "UYVY", src, w, h)
    assert np.array_equal(np.fromfile(out, np.uint8), po.dxt_encode(po.IN_UYVY_RAW, po.OUT_DXT1, uyvy, w, h))


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,codec", [("dxt:DXT5", "UYVY"), ("dxt:DXT5", "YUYV"), ("dxt:DXT1", "RGB"), ("jpeg:q=80:restart=4", "UYVY"), ("jpeg:q=80:restart=4", "v210")])
def test_device_resident_frames(tmp_path, po, cfg, codec):
    """a frame whose tile data already lives in device memory (types.h:295-298 mem_location; the harness also poisons its
    host copy) is encoded in place, without the upload -- same bytes as the host-frame path."""
    w, h, tiles = 192, 64, 2
    frames = [synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=20 + t) for t in range(tiles)]
    raw = tmp_path / "in.raw"
    np.concatenate(frames).tofile(raw)
    outs = []
    for mode in ([], ["dev"]):
        out = tmp_path / f"out{len(mode)}.bin"
        r = _run([cfg, codec, w, h, raw, out, tiles] + mode)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] and len(outs[0]) > 0
    if cfg.startswith("dxt"):
        oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1
        pin = {"UYVY": po.IN_UYVY, "YUYV": po.IN_UYVY, "RGB": po.IN_RGB}[codec]
        want = [po.dxt_encode(pin, oid, po.convert_frame("YUYV", "UYVY", f, w, h) if codec == "YUYV" else f, w, h) for f in frames]
        assert outs[1] == b"".join(x.tobytes() for x in want)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["dxt:DXT5:dev=0,0,0", "dxt:DXT1:dev=0:workers=1", "dxt:DXT5", "jpeg:q=70:restart=3:dev=0,0:workers=2"])
def test_frames_sharded_over_workers_in_order(tmp_path, po, cfg):
    """frames are dealt to the first idle worker (one per listed device; here one GPU listed several times, so
    several frames are in flight on it) and come back in push order with their sequence numbers, each identical to the
    single-frame result -- the reference's GPUJPEG scheme (gpujpeg.cpp:643-722) behind the asynchronous frame API."""
    w, h, n = 256, 64, 9
    frames = [synth.s1_random("UYVY", w, h, salt=40 + i) for i in range(n)]
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    np.concatenate(frames).tofile(raw)
    r = _run([cfg, "UYVY", w, h, raw, out, 1, "host", n])
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"frames={n}" in r.stdout and "seq=" + ",".join(str(i) for i in range(n)) + "," in r.stdout, r.stdout
    data = out.read_bytes()
    if cfg.startswith("dxt"):
        oid = po.OUT_DXT5YCOCG if "DXT5" in cfg else po.OUT_DXT1
        want = b"".join(po.dxt_encode(po.IN_UYVY, oid, f, w, h).tobytes() for f in frames)
        assert data == want
    else:
        singles = []
        for i, f in enumerate(frames):
            one_in, one_out = tmp_path / f"f{i}.raw", tmp_path / f"f{i}.jpg"
            f.tofile(one_in)
            assert _run(["jpeg:q=70:restart=3", "UYVY", w, h, one_in, one_out]).returncode == 0
            singles.append(one_out.read_bytes())
        assert data == b"".join(singles)


@needs_harness
@pytest.mark.gpu
def test_tiled_4k_fanout(tmp_path, po):
    """4 tiles ("tiled 4K", types.h:340-343): the framework fans tiles out to worker threads, one module state
    each (video_compress.cpp:441-490); every tile must match the oracle."""
    w, h, tiles = 960, 540 // 4 * 4, 4
    frames = [synth.s2_video("UYVY", w, h, salt=t) for t in range(tiles)]
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    np.concatenate(frames).tofile(raw)
    r = _run(["dxt:DXT5", "UYVY", w, h, raw, out, tiles])
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, np.uint8).reshape(tiles, -1)
    for t in range(tiles):
        assert np.array_equal(got[t], po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, frames[t], w, h)), t


def _ref_best_and_decode(po, codec, candidates, src, w, h):
    """What the reference's compress modules do on the CPU before upload (cuda_dxt.cpp:152-158,206-220): get_best_decoder_from(codec,
    candidates) of the COMPILED reference picks the target codec, its line decoder converts line by line (dst_len =
    vc_get_linesize(width, target), shifts 0/8/16).  Returns (target name, converted frame)."""
    import ctypes as C
    # RGBA: the -msse4.1 build's vc_copylineRGBAtoRGB never advances its source in the SSSE3 tail loop (pixfmt_conv.c:832-838) and
    # replicates one pixel over the last 4-7 pixels of a line; the reference's portable build is the pin there
    r = po.ref(scalar=codec == "RGBA")
    r.get_codec_from_name.argtypes = [C.c_char_p]
    r.get_codec_name.restype = C.c_char_p
    r.get_best_decoder_from.restype = C.c_void_p
    r.get_best_decoder_from.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    ci = r.get_codec_from_name(codec.encode())
    cand = (C.c_int * (len(candidates) + 1))(*[r.get_codec_from_name(c.encode()) for c in candidates], 0)
    out = C.c_int(0)
    fn = r.get_best_decoder_from(ci, cand, C.byref(out))
    if not fn:
        return None, None
    sls, dls = r.vc_get_linesize(w, ci), r.vc_get_linesize(w, out.value)
    pad = np.concatenate([np.ascontiguousarray(src, np.uint8).ravel(), np.zeros(64, np.uint8)])
    dst = np.zeros(dls * h + 64, np.uint8)
    dec = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)(fn)
    for y in range(h):
        dec(dst.ctypes.data + y * dls, pad.ctypes.data + y * sls, dls, 0, 8, 16)
    return r.get_codec_name(out.value).decode(), dst[: dls * h]


def _random_frame(po, codec, w, h, salt):
    import ctypes as C
    r = po.ref()
    r.get_codec_from_name.argtypes = [C.c_char_p]
    n = r.vc_get_linesize(w, r.get_codec_from_name(codec.encode())) * h
    buf = np.random.default_rng(1000 + salt).integers(0, 256, n, dtype=np.uint8)
    if codec in ("v210", "DVS10"):
        buf = (buf.view(np.uint32) & 0x3FFFFFFF).view(np.uint8)
    return buf


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec,w,h", [("R10k", 192, 64), ("R12L", 192, 64), ("RG48", 192, 64), ("Y216", 192, 64), ("Y416", 192, 64),
                                       ("VUYA", 192, 64), ("DVS10", 192, 64), ("R12L", 200, 8), ("Y216", 100, 12), ("R10k", 52, 4),
                                       ("v210", 1280, 720), ("v210", 2048, 1080), ("v210", 52, 8), ("RGBA", 200, 16), ("BGR", 100, 8)])
@pytest.mark.parametrize("cfg", ["dxt:DXT5", "dxt:DXT1"])
def test_every_codec_the_reference_module_takes(tmp_path, po, codec, w, h, cfg):
    """`-c dxt` takes whatever cuda_dxt.cpp takes -- every codec with a decoder to RGB or UYVY
    (get_best_decoder_from, cuda_dxt.cpp:152-158, pixfmt_conv.c:3126-3172), v210 at widths that are not multiples of 12 included
    (1280x720, 2048x1080) -- and its output is bit-equal to the reference's own sequence: the COMPILED reference's choice of target
    codec and its line decoder, then the DXT oracle on the result."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libugref.so not built")
    src = _random_frame(po, codec, w, h, salt=w + h)
    target, conv = _ref_best_and_decode(po, codec, ["RGB", "UYVY"], src, w, h)
    assert target in ("RGB", "UYVY"), target
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    src.tofile(raw)
    r = subprocess.run([HARNESS, cfg, codec, str(w), str(h), str(raw), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1
    want = po.dxt_encode(po.IN_RGB if target == "RGB" else po.IN_UYVY, oid, conv, w, h, threads=0)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["Y216", "R10k", "v210"])
def test_dxt1_yuv_from_wide_codecs(tmp_path, po, codec):
    """-c dxt:DXT1_YUV: UYVY is the only encoder input (dxt_glsl.cpp:104-110); every other codec goes through its decoder to UYVY."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libugref.so not built")
    w, h = 200, 16
    src = _random_frame(po, codec, w, h, salt=3)
    target, conv = _ref_best_and_decode(po, codec, ["UYVY"], src, w, h)
    assert target == "UYVY"
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    src.tofile(raw)
    r = _run(["dxt:DXT1_YUV", codec, w, h, raw, out])
    assert r.returncode == 0 and "DXT1_YUV" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(out, np.uint8), po.dxt_encode(po.IN_UYVY_RAW, po.OUT_DXT1, conv, w, h))


@needs_harness
@pytest.mark.gpu
def test_tie_rule_option(tmp_path, po):
    """ties=even (default) / ties=away select UG_DXT_TIES_*: the S3 colour bars in RGB sit on round() ties in every white block."""
    w, h = 192, 64
    src = synth.s3_bars("RGB", w, h)
    raw = tmp_path / "in.raw"
    src.tofile(raw)
    outs = {}
    for opt, ties in (("", "even"), (":ties=even", "even"), (":ties=away", "away")):
        out = tmp_path / f"o{len(outs)}.bin"
        r = _run(["dxt:DXT5" + opt, "RGB", w, h, raw, out])
        assert r.returncode == 0, r.stdout + r.stderr
        outs[opt] = out.read_bytes()
        assert outs[opt] == po.dxt_encode(po.IN_RGB, po.OUT_DXT5YCOCG, src, w, h, ties=ties).tobytes(), opt
    assert outs[""] != outs[":ties=away"]
    assert _run(["dxt:ties=sometimes", "UYVY", w, h, raw, tmp_path / "x"]).returncode == 2


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec,nbytes", [("I420", 64 * 16 * 3 // 2), ("DXT1", 64 * 16 // 2), ("H.264", 64 * 16)])
def test_only_what_the_reference_refuses_is_refused(tmp_path, po, codec, nbytes):
    """Planar and compressed codecs have no decoder to RGB / UYVY: get_best_decoder_from() returns NULL in the reference, which prints
    "Unsupported codec" and drops the frame (cuda_dxt.cpp:154-158) -- same here, and nothing is faked on the CPU."""
    if po.have_ref():
        assert _ref_best_and_decode(po, codec, ["RGB", "UYVY"], np.zeros(nbytes, np.uint8), 64, 16)[0] is None
    raw = tmp_path / "in.raw"
    np.zeros(max(nbytes, 64 * 16 * 4), np.uint8).tofile(raw)
    r = _run(["dxt:DXT5", codec, 64, 16, raw, tmp_path / "o.bin"])
    assert r.returncode in (1, 3) and (r.returncode == 1 or "Unsupported codec" in (r.stdout + r.stderr)), r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------------
# receiver side: the reference's src/video_decompress.c selects our C decompress module by priority
# ---------------------------------------------------------------------------------------------------------
DEC_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ug_dec_harness")
needs_dec_harness = pytest.mark.skipif(not os.path.exists(DEC_HARNESS), reason="oracle/_ref/ug_dec_harness not built")


@needs_dec_harness
def test_decompress_module_registers():
    r = subprocess.run([DEC_HARNESS, "list"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "dxt_mi355x" in r.stdout.split()


@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("out", ["RGBA", "RGB", "UYVY"])
@pytest.mark.parametrize("comp", ["DXT1", "DXT1_YUV", "DXT5"])
def test_decompress_through_reference_framework(tmp_path, po, comp, out):
    w, h = 192, 64
    oid = {"DXT1": po.OUT_DXT1, "DXT1_YUV": po.OUT_DXT1_YUV, "DXT5": po.OUT_DXT5YCOCG}[comp]
    blocks = po.dxt_encode(po.IN_UYVY_RAW if comp == "DXT1_YUV" else po.IN_UYVY, po.OUT_DXT1 if comp == "DXT1_YUV" else oid, synth.s2_video("UYVY", w, h), w, h)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    blocks.tofile(src)
    r = subprocess.run([DEC_HARNESS, comp, out, str(w), str(h), str(src), str(dst)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(dst, np.uint8), po.dxt_decode(oid, out, blocks, w, h))
    # display pitch larger than the packed line (dxt_glsl.c:163-186 path)
    ls = po.linesize(w, out)
    pitch = ls + 64
    r = subprocess.run([DEC_HARNESS, comp, out, str(w), str(h), str(src), str(dst), str(pitch)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(dst, np.uint8).reshape(h, pitch)[:, :ls]
    assert np.array_equal(got.ravel(), po.dxt_decode(oid, out, blocks, w, h))


@needs_dec_harness
@pytest.mark.gpu
def test_short_frame_decodes_what_arrived_and_black_for_the_rest(tmp_path, po):
    """the module accepts corrupted (short) frames; the blocks that did not arrive must decode as all-zero blocks, not as
    whatever the freshly allocated device buffer held."""
    w, h = 192, 64
    blocks = po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, synth.s2_video("UYVY", w, h), w, h)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    blocks.tofile(src)
    got_len = (blocks.size // 3) // 16 * 16
    ls = po.linesize(w, "RGBA")
    r = subprocess.run([DEC_HARNESS, "DXT5", "RGBA", str(w), str(h), str(src), str(dst), str(ls), str(got_len)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0, r.stdout + r.stderr
    partial = blocks.copy()
    partial[got_len:] = 0
    assert np.array_equal(np.fromfile(dst, np.uint8), po.dxt_decode(po.OUT_DXT5YCOCG, "RGBA", partial, w, h))


SHARDER_TEST = os.path.join(ROOT, "oracle", "_ref", "ug_sharder_test")


@pytest.mark.skipif(not os.path.exists(SHARDER_TEST), reason="oracle/_ref/ug_sharder_test not built")
@pytest.mark.parametrize("workers,frames", [(1, 40), (4, 300), (8, 500)])
def test_frame_sharder_cpu(workers, frames):
    """mi355x::frame_sharder with a fake tile encoder (random delays, injected failures, tiled frames): in-order delivery, failed
    frames skipped, all workers used, metadata kept, the poison pill ends the stream (module/ug_sharder_test.cpp).  No GPU."""
    r = subprocess.run([SHARDER_TEST, str(workers), str(frames)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(SHARDER_TEST), reason="oracle/_ref/ug_sharder_test not built")
@pytest.mark.parametrize("workers,frames,batch", [(1, 100, 4), (2, 300, 8), (3, 500, 16)])
def test_frame_sharder_batching_cpu(workers, frames, batch):
    """batch=<n>: a busy worker queues up to n frames and a batch encoder takes what has queued up; the same properties hold, batches
    do form, tiled frames and mixed rounds fall back to one at a time, and a frame OBJECT pushed twice in a row (two sequence numbers)
    comes out twice, in order -- the sequence number travels with the queue entry, not in the shared frame."""
    r = subprocess.run([SHARDER_TEST, str(workers), str(frames), str(batch)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK") and "batch_calls=0" not in r.stdout, r.stdout + r.stderr


@needs_harness
def test_jpeg_module_registers():
    r = _run(["list"])
    assert "jpeg" in r.stdout.split()   # the name the reference uses as the hidden alias of its GPUJPEG module (gpujpeg.cpp:791-792)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("sub", [None, 420])
@pytest.mark.parametrize("codec", ["UYVY", "v210", "RGB", "YUYV", "RGBA", "I420"])
def test_jpeg_through_reference_framework(tmp_path, po, codec, sub):
    """-c jpeg:q=80:restart=4[:subsampling=420] through compress_init/compress_frame/compress_pop: the stream is what the test
    writer produces from the oracle's coefficients and libjpeg decodes it.  Without the option the module codes the input's own
    sampling like the reference (gpujpeg.cpp:295-305): 4:2:2 for UYVY/YUYV/v210, R,G,B 4:4:4 for RGB/RGBA, 4:2:0 for I420; with
    subsampling=420 RGB-family input goes through the pixfmt_conv.c RGB->UYVY arithmetic first."""
    import io
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from jpeg_bitstream import write_jpeg
    w, h = 192, 96
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    uyvy0 = po.convert_frame("RGB", "UYVY", rgb, w, h)
    rgba = po.convert_frame("RGB", "RGBA", rgb, w, h)
    i420 = np.concatenate([p.ravel() for p in po.uyvy_to_i420(uyvy0, w, h)])
    src = {"UYVY": uyvy0, "RGB": rgb.ravel(), "RGBA": rgba, "YUYV": po.convert_frame("UYVY", "YUYV", uyvy0, w, h),
           "v210": po.convert_frame("UYVY", "v210", uyvy0, w, h), "I420": i420}[codec]
    raw, out = tmp_path / "in.raw", tmp_path / "out.jpg"
    np.ascontiguousarray(src).tofile(raw)
    r = _run(["jpeg:q=80:restart=4" + (f":subsampling={sub}" if sub else ""), codec, w, h, raw, out])
    assert r.returncode == 0 and "JPEG" in r.stdout, r.stdout + r.stderr
    data = out.read_bytes()
    ql, qc = po.jpeg_qtable(80, 0), po.jpeg_qtable(80, 1)
    eff = sub or {"RGB": 444, "RGBA": 444, "I420": 420}.get(codec, 422)
    if eff == 444:
        comp = rgb if codec == "RGB" else po.convert_frame("RGBA", "RGB", rgba, w, h).reshape(h, w, 3)
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(comp[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        from jpeg_bitstream import write_jpeg_noninterleaved
        assert data == write_jpeg_noninterleaved(w, h, ql, coefs, restart=4)        # RGB input: one scan per component unless `:interleaved` (gpujpeg.cpp:303)
        out_i = tmp_path / "out_i.jpg"
        assert _run(["jpeg:q=80:restart=4:interleaved" + (f":subsampling={sub}" if sub else ""), codec, w, h, raw, out_i]).returncode == 0
        assert out_i.read_bytes() == write_jpeg(w, h, ql, qc, *coefs, restart=4, sub=444)
        img = Image.open(io.BytesIO(data))
        assert img.mode == "RGB"
        assert 10 * np.log10(255.0 ** 2 / np.mean((np.asarray(img).astype(float) - rgb.astype(float)) ** 2)) > 36
        return
    uyvy = {"v210": lambda: po.convert_frame("v210", "UYVY", src, w, h), "RGBA": lambda: po.convert_frame("RGBA", "UYVY", rgba, w, h)}.get(codec, lambda: uyvy0)()
    mw = (w + 15) // 16
    if eff == 420:
        y, u, v = po.uyvy_to_i420(uyvy, w, h)
        mh, vy = (h + 15) // 16, 2
    else:
        y, u, v = po.uyvy_to_i422(uyvy, w, h)
        mh, vy = (h + 7) // 8, 1
    want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, vy * mh),
                      po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh), po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh),
                      restart=4, sub=eff)
    assert data == want
    img = Image.open(io.BytesIO(data))
    img.draft("YCbCr", None)
    dec = np.asarray(img)
    assert 10 * np.log10(255.0 ** 2 / np.mean((dec[..., 0].astype(float) - y.astype(float)) ** 2)) > 40


@needs_harness
@pytest.mark.gpu
def test_jpeg_module_rejects_impossible_subsampling(tmp_path):
    raw = tmp_path / "in.raw"
    np.zeros(64 * 64 * 2, np.uint8).tofile(raw)
    assert _run(["jpeg:subsampling=411", "UYVY", 64, 64, raw, tmp_path / "o"]).returncode != 0
    np.zeros(63 * 64 + 2 * 32 * 32, np.uint8).tofile(raw)
    assert _run(["jpeg:subsampling=422", "I420", 63, 64, raw, tmp_path / "o"]).returncode != 0   # planar input with an option goes through UYVY: pixel pairs


@needs_harness
@pytest.mark.gpu
def test_tiles_dealt_over_device_list(tmp_path, po):
    """dev=<list> with a tiled frame: the frame goes to one worker, which encodes its tiles with per-tile encoder states on its
    device (one GPU here, listed twice: the code path is the multi-GPU one, the data path has no inter-device traffic)."""
    w, h, tiles = 384, 128, 4
    frames = [synth.s1_random("UYVY", w, h, salt=10 + t) for t in range(tiles)]
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    np.concatenate(frames).tofile(raw)
    r = _run(["dxt:DXT5:dev=0,0", "UYVY", w, h, raw, out, tiles])
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, np.uint8).reshape(tiles, -1)
    for t in range(tiles):
        assert np.array_equal(got[t], po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, frames[t], w, h)), t
    r = _run(["dxt:DXT5:dev=7", "UYVY", w, h, raw, out, tiles])   # no such device on a 1-GPU box: refused at init
    assert r.returncode == 2 and "cannot use HIP device 7" in (r.stdout + r.stderr)


@needs_harness
@pytest.mark.gpu
def test_reference_jpeg_fixture_end_to_end(tmp_path):
    """test/gpujpeg_test.cpp:68-106, the only fixture the reference holds at this boundary: a 1920x1080 RGB frame of all-127 bytes through
    compress_init / compress_frame / compress_pop with the module's default parameters, decoded again (there: by the gpujpeg decompress
    module; here: by libjpeg through Pillow, an independent decoder), max |diff| <= 1 over every byte."""
    import io
    from PIL import Image
    w, h = 1920, 1080
    raw, out = tmp_path / "in.raw", tmp_path / "out.jpg"
    np.full(w * h * 3, 127, np.uint8).tofile(raw)
    r = subprocess.run([HARNESS, "jpeg", "RGB", str(w), str(h), str(raw), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "JPEG" in r.stdout, r.stdout + r.stderr
    img = Image.open(io.BytesIO(out.read_bytes()))
    assert img.size == (w, h) and img.mode == "RGB"
    dec = np.asarray(img).astype(int)
    assert np.abs(dec - 127).max() <= 1


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["RGB", "UYVY"])
@pytest.mark.parametrize("kind", ["S2", "S3"])
def test_jpeg_default_parameters_on_real_content(tmp_path, po, kind, codec):
    """The same path on S2 (low-pass video noise) and S3 (colour bars + ramp) content at 1080p with the module defaults (q=75): decodes
    with libjpeg to a PSNR floor against what went in (RGB: the frame itself; UYVY: its luma plane)."""
    import io
    from PIL import Image
    w, h = 1920, 1080
    src = synth.frame(kind, codec, w, h)
    raw, out = tmp_path / "in.raw", tmp_path / "out.jpg"
    src.tofile(raw)
    r = subprocess.run([HARNESS, "jpeg", codec, str(w), str(h), str(raw), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    img = Image.open(io.BytesIO(out.read_bytes()))
    if codec == "RGB":
        ref = src.reshape(h, w, 3).astype(float)
        dec = np.asarray(img).astype(float)
    else:
        img.draft("YCbCr", None)
        dec = np.asarray(img)[..., 0].astype(float)
        ref = src.reshape(h, w, 2)[..., 1].astype(float)
    psnr = 10 * np.log10(255.0 ** 2 / max(np.mean((dec - ref) ** 2), 1e-9))
    assert psnr > (30 if kind == "S2" else 36), psnr   # S2 carries N(0,2) per-pixel noise that q=75 removes


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["R10k", "R12L", "RG48", "Y216", "Y416", "VUYA", "DVS10", "BGR"])
def test_jpeg_every_codec_the_reference_module_takes(tmp_path, po, codec):
    """-c jpeg vs gpujpeg.cpp:227-236,262-272,592-608: the frame is decoded to what get_best_decoder_from(codec, {UYVY, RGB, RGBA}) of the
    COMPILED reference ranks first, with its line decoder, and coded with that codec's own subsampling (4:2:2 for UYVY, R,G,B 4:4:4 for
    RGB / RGBA).  The stream must be what the test writer makes of the oracle's coefficients of exactly those samples."""
    import sys
    if not po.have_ref():
        pytest.skip("oracle/_ref/libugref.so not built")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from jpeg_bitstream import write_jpeg
    w, h = 192, 96
    # smooth content in the codec's own format: start from an RGB picture, bring it into `codec` with the reference's converters
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    src = _to_codec(po, rgb, codec, w, h)
    target, conv = _ref_best_and_decode(po, codec, ["UYVY", "RGB", "RGBA"], src, w, h)
    assert target in ("UYVY", "RGB", "RGBA")
    raw, out = tmp_path / "in.raw", tmp_path / "out.jpg"
    src.tofile(raw)
    r = _run(["jpeg:q=80:restart=4", codec, w, h, raw, out])
    assert r.returncode == 0 and "JPEG" in r.stdout, r.stdout + r.stderr
    ql, qc = po.jpeg_qtable(80, 0), po.jpeg_qtable(80, 1)
    if target == "UYVY":
        y, u, v = po.uyvy_to_i422(conv, w, h)
        mw, mh = (w + 15) // 16, (h + 7) // 8
        want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, mh), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh),
                          po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh), restart=4, sub=422)
    else:
        comp = conv.reshape(h, w, 4 if target == "RGBA" else 3)
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(comp[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        from jpeg_bitstream import write_jpeg_noninterleaved
        want = write_jpeg_noninterleaved(w, h, ql, coefs, restart=4)
    assert out.read_bytes() == want


def _to_codec(po, rgb, codec, w, h):
    """an RGB picture in `codec`, through the compiled reference's own converters (whatever chain decoders[] offers)"""
    import ctypes as C
    r = po.ref()
    r.get_codec_from_name.argtypes = [C.c_char_p]
    r.get_decoder_from_to.restype = C.c_void_p

    def conv(a, i, o):
        ci, co = r.get_codec_from_name(i.encode()), r.get_codec_from_name(o.encode())
        fn = r.get_decoder_from_to(ci, co)
        assert fn, (i, o)
        sls, dls = r.vc_get_linesize(w, ci), r.vc_get_linesize(w, co)
        pad = np.concatenate([np.ascontiguousarray(a, np.uint8).ravel(), np.zeros(64, np.uint8)])
        dst = np.zeros(dls * h + 64, np.uint8)
        dec = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)(fn)
        for y in range(h):
            dec(dst.ctypes.data + y * dls, pad.ctypes.data + y * sls, r.vc_get_size(w, co), 0, 8, 16)
        return dst[: dls * h]
    chain = {"R10k": ["RGBA", "R10k"], "R12L": ["R12L"], "RG48": ["RG48"], "Y216": ["UYVY", "Y216"], "Y416": ["UYVY", "Y416"], "VUYA": ["RGBA", "VUYA"],
             "DVS10": ["UYVY", "v210"], "BGR": None}[codec]
    if chain is None:
        return np.ascontiguousarray(rgb[..., ::-1]).ravel()
    cur, name = rgb.ravel(), "RGB"
    for nxt in chain:
        cur = conv(cur, name, nxt)
        name = nxt
    return cur   # DVS10: v210 bytes are a valid DVS10 frame as far as the decoder is concerned (same 6 px / 16 B grouping)


@needs_harness
@pytest.mark.gpu
def test_jpeg_option_forms_of_the_reference_module(tmp_path):
    """gpujpeg.cpp:376-418: `<quality>[:<restart>]` positionally, quality= / restart=, interleaved, the internal colour space the module
    would use anyway (RGB for RGB input, Y709 for 4:2:x input); what needs a colour conversion or alpha is refused / warned about."""
    w, h = 64, 32
    raw = tmp_path / "in.raw"
    synth.s2_video("UYVY", w, h).tofile(raw)
    outs = []
    for cfg in ("jpeg:q=60:restart=3", "jpeg:60:3", "jpeg:quality=60:restart=3:interleaved:Y709", "jpeg:qual=60:r=3:inter:sub=422"):   # (<k>=<v> with any prefix of the key: IS_KEY_PREFIX, utils/macros.h:162-164)
        out = tmp_path / f"o{len(outs)}.jpg"
        r = _run([cfg, "UYVY", w, h, raw, out])
        assert r.returncode == 0, cfg + r.stdout + r.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2] == outs[3]
    out = tmp_path / "nori.jpg"                                                         # restart interval 0 = none (the value goes to GPUJPEG as it is, gpujpeg.cpp:345)
    r = _run(["jpeg:60:0", "UYVY", w, h, raw, out])
    assert r.returncode == 0, r.stdout + r.stderr
    data = out.read_bytes()
    assert b"\xff\xdd" not in data[:700] and not any(bytes([0xff, 0xd0 + k]) in data[600:] for k in range(8))
    import io
    from PIL import Image
    a, b = (np.asarray(Image.open(io.BytesIO(d)).convert("L")).astype(float) for d in (data, outs[0]))
    assert np.array_equal(a, b)                                                          # the same coefficients, with and without restart markers
    if os.path.exists(DEC_HARNESS):                                                      # and the receiving module reads both to the same picture
        back = []
        for name, d in (("nori", data), ("ri3", outs[0])):
            (tmp_path / f"{name}.jpg").write_bytes(d)
            r = subprocess.run([DEC_HARNESS, "JPEG", "UYVY", str(w), str(h), str(tmp_path / f"{name}.jpg"), str(tmp_path / f"{name}.raw")], capture_output=True, text=True, timeout=60)
            assert r.returncode == 0, r.stdout + r.stderr
            back.append(np.fromfile(tmp_path / f"{name}.raw", np.uint8)[:2 * w * h])
        assert np.array_equal(back[0], back[1])
    assert _run(["jpeg:RGB", "UYVY", w, h, raw, tmp_path / "x"]).returncode == 3       # configure fails (R, G, B components are coded 4:4:4 only; `:subsampling=444:RGB` is taken): frame dropped
    r = _run(["jpeg:alpha", "UYVY", w, h, raw, tmp_path / "x"])
    assert r.returncode == 0 and "Requested alpha encode but input codec is unsupported pixel format" in (r.stdout + r.stderr)      # gpujpeg.cpp:327-328
    rgba = tmp_path / "rgba.raw"
    synth.s1_random("RGBA", w, h).tofile(rgba)
    r = _run(["jpeg:alpha", "RGBA", w, h, rgba, tmp_path / "x"])
    assert r.returncode == 3 and "fourth component" in (r.stdout + r.stderr)            # refused, not silently dropped
    assert _run(["jpeg", "RGBA", w, h, rgba, tmp_path / "x"]).returncode == 0


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("opt,cs", [("", 4), (":Y709", 4), (":Y601", 2), (":Y601full", 3), (":RGB", 1)])
@pytest.mark.parametrize("codec", ["UYVY", "v210"])
def test_jpeg_444_from_a_422_source(tmp_path, po, codec, opt, cs):
    """`-c jpeg:subsampling=444` on 4:2:2 input (gpujpeg.cpp:297-302: the option overrides the codec's own subsampling): every pixel with its pair's chroma,
    one interleaved scan (the input is not RGB, :303), coded in BT.709 as it comes, in BT.601, or as R, G, B with `:RGB`; bytes == the test writer's over the
    oracle's 3 B/px picture; the product's decompress module gives the source back within the codec's loss"""
    from jpeg_bitstream import write_jpeg
    w, h = 208, 72
    uyvy = synth.s2_video("UYVY", w, h, salt=3)
    src = uyvy if codec == "UYVY" else po.convert_frame("UYVY", "v210", uyvy, w, h)
    as_uyvy = uyvy if codec == "UYVY" else po.convert_frame("v210", "UYVY", src, w, h)
    raw, out = tmp_path / "in.raw", tmp_path / "o.jpg"
    np.ascontiguousarray(src).tofile(raw)
    r = _run([f"jpeg:q=85:restart=4:subsampling=444{opt}", codec, w, h, raw, out])
    assert r.returncode == 0, r.stdout + r.stderr
    pic = po.jpeg_colour_convert("UYVY444", 4, cs, as_uyvy, w, h).reshape(h, w, 3)
    ql, qc = po.jpeg_qtable(85, 0), po.jpeg_qtable(85, 1)
    coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(pic[..., c]), po.jpeg_divisors(ql if c == 0 or cs == 1 else qc), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
    assert out.read_bytes() == write_jpeg(w, h, ql, qc, *coefs, restart=4, sub=444, ycc=cs != 1)     # (R, G, B: table 0 for every component)
    if os.path.exists(DEC_HARNESS) and cs in (4, 1):
        back = tmp_path / "back.raw"
        r = subprocess.run([DEC_HARNESS, "JPEG", "UYVY", str(w), str(h), str(out), str(back)], capture_output=True, text=True, timeout=30)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(back, np.uint8)[:2 * w * h].reshape(h, w, 2)[..., 1].astype(float)
        assert 10 * np.log10(255.0 ** 2 / np.mean((got - as_uyvy.reshape(h, w, 2)[..., 1]) ** 2)) > 32


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["subsampling=422", "subsampling=444", "Y601", "Y601full:subsampling=420", "subsampling=444:RGB", "subsampling=420:Y709"])
def test_jpeg_planar_input_with_options(tmp_path, po, cfg):
    """I420 input is handed over as it is (gpujpeg.cpp:335) unless an option asks for another subsampling or colour space -- GPUJPEG then resamples / converts in its
    preprocessor; here the picture goes through the reference's i420_8_to_uyvy shuffle (both lines of a pair take the chroma line) and on as UYVY input: the
    stream == the one UYVY input of those samples gives; with nothing to change (`subsampling=420:Y709`) == the planar path's"""
    w, h = 208, 80
    uyvy0 = synth.s2_video("UYVY", w, h, salt=5)
    planes = po.uyvy_to_i420(uyvy0, w, h)
    i420 = np.concatenate([p.ravel() for p in planes])
    as_uyvy = np.empty((h, w // 2, 4), np.uint8)
    as_uyvy[..., 1], as_uyvy[..., 3] = planes[0][:, 0::2], planes[0][:, 1::2]
    as_uyvy[..., 0], as_uyvy[..., 2] = np.repeat(planes[1], 2, axis=0)[:h], np.repeat(planes[2], 2, axis=0)[:h]
    a, b, oa, ob = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "a.jpg", tmp_path / "b.jpg"
    i420.tofile(a)
    as_uyvy.tofile(b)
    r = _run([f"jpeg:q=85:restart=4:{cfg}", "I420", w, h, a, oa])
    assert r.returncode == 0, r.stdout + r.stderr
    if cfg == "subsampling=420:Y709":
        assert _run(["jpeg:q=85:restart=4", "I420", w, h, a, ob]).returncode == 0
    else:
        assert _run([f"jpeg:q=85:restart=4:{cfg}" + ("" if "subsampling" in cfg else ":subsampling=420"), "UYVY", w, h, b, ob]).returncode == 0     # (I420's own subsampling where none is asked for)
    assert oa.read_bytes() == ob.read_bytes() and len(oa.read_bytes()) > 1000
    import io
    from PIL import Image
    Image.open(io.BytesIO(oa.read_bytes())).load()


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("opt,cs", [("Y601", 2), ("Y601full", 3), ("Y709", 4)])
@pytest.mark.parametrize("codec", ["RGB", "RGBA", "UYVY", "v210"])
def test_jpeg_internal_colour_space_options(tmp_path, po, codec, opt, cs):
    """`-c jpeg:Y601 | Y601full | Y709` (gpujpeg.cpp:398-403): RGB-family input is coded as Y'CbCr 4:4:4 of that space (one scan per component, or one
    with `:interleaved`), 4:2:x input as BT.601 where asked; bytes == the test writer's over the oracle's colour stage + FDCT; Pillow reads them all."""
    import io
    from PIL import Image
    from jpeg_bitstream import write_jpeg, write_jpeg_noninterleaved
    w, h = 200, 72
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    ql, qc = po.jpeg_qtable(85, 0), po.jpeg_qtable(85, 1)
    raw = tmp_path / "in.raw"
    if codec in ("RGB", "RGBA"):
        src = rgb.ravel() if codec == "RGB" else po.convert_frame("RGB", "RGBA", rgb, w, h)
        np.ascontiguousarray(src).tofile(raw)
        picture = rgb if codec == "RGB" else po.convert_frame("RGBA", "RGB", src, w, h).reshape(h, w, 3)
        ycc = po.jpeg_colour_convert("RGB", 1, cs, picture, w, h).reshape(h, w, 3)
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(ycc[..., c]), po.jpeg_divisors(ql if c == 0 else qc), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        for cfg, want in ((f"jpeg:q=85:restart=4:{opt}", write_jpeg_noninterleaved(w, h, ql, coefs, restart=4, qt_chroma=qc)),
                          (f"jpeg:q=85:restart=4:{opt}:interleaved", write_jpeg(w, h, ql, qc, *coefs, restart=4, sub=444, ycc=True))):
            out = tmp_path / "o.jpg"
            r = _run([cfg, codec, w, h, raw, out])
            assert r.returncode == 0, cfg + r.stdout + r.stderr
            assert out.read_bytes() == want, cfg
            img = np.asarray(Image.open(io.BytesIO(want)).convert("RGB")).astype(float)
            assert 10 * np.log10(255.0 ** 2 / np.mean((img - rgb) ** 2)) > (34 if opt == "Y601full" else 14)
    else:
        uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        src = uyvy if codec == "UYVY" else po.convert_frame("UYVY", "v210", uyvy, w, h)
        np.ascontiguousarray(src).tofile(raw)
        as_uyvy = uyvy if codec == "UYVY" else po.convert_frame("v210", "UYVY", src, w, h)
        conv = as_uyvy if opt == "Y709" else po.jpeg_colour_convert("UYVY", 4, cs, as_uyvy, w, h)
        y, u, v = po.uyvy_to_i422(conv, w, h)
        mw, mh = (w + 15) // 16, (h + 7) // 8
        want = write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, mh), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh),
                          po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh), restart=4, sub=422)
        out = tmp_path / "o.jpg"
        r = _run([f"jpeg:q=85:restart=4:{opt}", codec, w, h, raw, out])
        assert r.returncode == 0, r.stdout + r.stderr
        assert out.read_bytes() == want
        Image.open(io.BytesIO(want)).load()


@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("out", ["UYVY", "RGB", "RGBA"])
def test_greyscale_jpeg_through_the_decompress_framework(tmp_path, po, out):
    """a one-component JPEG (another sender's; GPUJPEG_U8 in gpujpeg.c:239-241) through decompress_init_multi / reconfigure / decompress_frame: the probe
    answers, the picture is the luma plane with neutral chroma"""
    import io
    from PIL import Image
    w, h = 320, 176
    yy, xx = np.mgrid[0:h, 0:w]
    grey = (128 + 100 * np.sin(xx / 23.0) * np.cos(yy / 13.0)).clip(0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(grey, "L").save(b, "JPEG", quality=90, restart_marker_blocks=4)
    src, dst = tmp_path / "g.jpg", tmp_path / "o.raw"
    src.write_bytes(b.getvalue())
    r = subprocess.run([DEC_HARNESS, "JPEG", out, str(w), str(h), str(src), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    luma = np.asarray(Image.open(io.BytesIO(b.getvalue())))
    uyvy = np.empty((h, w, 2), np.uint8)
    uyvy[..., 0], uyvy[..., 1] = 128, luma
    want = uyvy.ravel() if out == "UYVY" else po.convert_frame("UYVY", out, uyvy.ravel(), w, h)
    assert np.array_equal(np.fromfile(dst, np.uint8)[:want.size], want)


@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("rows", [0, 8])
def test_third_party_jpeg_with_long_segments_through_the_decompress_framework(tmp_path, po, rows):
    """a libjpeg / FFmpeg-style stream (no restart intervals, or one per 8 MCU rows) through decompress_frame: the parallel decode of long segments inside the module"""
    import io
    from PIL import Image
    w, h = 1280, 720
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1)
    rgb = (rgb + np.random.default_rng(2).normal(0, 5, rgb.shape)).clip(0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=80, subsampling=1, **({"restart_marker_rows": rows} if rows else {}))
    src, dst = tmp_path / "t.jpg", tmp_path / "o.raw"
    src.write_bytes(b.getvalue())
    r = subprocess.run([DEC_HARNESS, "JPEG", "UYVY", str(w), str(h), str(src), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    _, crop, _ = po.jpeg_decode_planes(b.getvalue())
    assert np.array_equal(np.fromfile(dst, np.uint8)[:2 * w * h], po.planar_to_uyvy(*crop, w, h, chroma=422))


@needs_dec_harness
def test_jpeg_decompress_module_registers():
    r = subprocess.run([DEC_HARNESS, "list"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "jpeg_mi355x" in r.stdout.split()


@needs_harness
@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec,cfg,out", [("UYVY", "jpeg:q=85:restart=4", "UYVY"), ("UYVY", "jpeg:q=85:restart=4", "RGB"), ("UYVY", "jpeg:q=85:restart=2:subsampling=420", "I420"),
                                           ("RGB", "jpeg:q=85:restart=4", "RGBA"), ("RGB", "jpeg:q=85:restart=4", "UYVY"), ("v210", "jpeg:q=90", "UYVY")])
def test_sender_to_receiver_through_both_reference_frameworks(tmp_path, po, codec, cfg, out):
    """The whole `-c jpeg` story inside UltraGrid's own frameworks: compress_init / compress_frame / compress_pop (video_compress.cpp) on the
    sending side, then decompress_init_multi / reconfigure / decompress_frame (video_decompress.c: probe first, as the receiver does) on the
    other; the module's output is the decode oracle's planes put together the way the header of the decoder states, and close to what went in."""
    w, h = 192, 96
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
    src = {"UYVY": uyvy, "RGB": rgb.ravel(), "v210": po.convert_frame("UYVY", "v210", uyvy, w, h)}[codec]
    raw, jpg, dec = tmp_path / "in.raw", tmp_path / "f.jpg", tmp_path / "out.raw"
    np.ascontiguousarray(src).tofile(raw)
    assert _run([cfg, codec, w, h, raw, jpg]).returncode == 0
    data = jpg.read_bytes()
    ls = w * h if out == "I420" else po.linesize(w, out)
    pitch = ls if out == "I420" else ls + (64 if out == "RGB" else 0)      # one case with a display pitch
    r = subprocess.run([DEC_HARNESS, "JPEG", out, str(w), str(h), str(jpg), str(dec), str(pitch)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    is_rgb = codec == "RGB"
    sub = 4200 if "420" in cfg else (4440 if is_rgb else 4220)
    assert f"depth=8 subsampling={sub} rgb={int(is_rgb)}" in r.stdout, r.stdout
    _, crop, _ = po.jpeg_decode_planes(data)
    got = np.fromfile(dec, np.uint8)
    if out != "I420":
        got = got.reshape(h, pitch)[:, :ls].ravel()
    if is_rgb:
        packed = np.stack(crop, -1)
        want = {"RGBA": lambda: po.convert_frame("RGB", "RGBA", packed, w, h), "UYVY": lambda: po.convert_frame("RGB", "UYVY", packed, w, h)}[out]()
        ref_in = {"RGBA": lambda: po.convert_frame("RGB", "RGBA", rgb, w, h), "UYVY": lambda: uyvy}[out]()
    else:
        chroma = 420 if "420" in cfg else 422
        as_uyvy = po.planar_to_uyvy(*crop, w, h, chroma=chroma)
        want = {"UYVY": lambda: as_uyvy, "RGB": lambda: po.convert_frame("UYVY", "RGB", as_uyvy, w, h), "I420": lambda: np.concatenate([p.ravel() for p in crop])}[out]()
        ref_in = {"UYVY": lambda: uyvy, "RGB": lambda: po.convert_frame("UYVY", "RGB", uyvy, w, h), "I420": lambda: np.concatenate([p.ravel() for p in po.uyvy_to_i420(uyvy, w, h)])}[out]()
    assert np.array_equal(got, want)
    assert 10 * np.log10(255.0 ** 2 / np.mean((got.astype(float) - ref_in.astype(float)) ** 2)) > 33


@needs_dec_harness
def test_jpeg_to_dxt_transcoder_registers():
    r = subprocess.run([DEC_HARNESS, "list"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "jpeg_to_dxt_mi355x" in r.stdout.split()


@needs_harness
@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec,cfg", [("UYVY", "jpeg:q=85:restart=4"), ("RGB", "jpeg:q=85:restart=4"), ("UYVY", "jpeg:q=90:restart=2:subsampling=420")])
@pytest.mark.parametrize("out", ["DXT1", "DXT5"])
def test_jpeg_to_dxt_transcoder(tmp_path, po, codec, cfg, out):
    """JPEG -> DXT1 / DXT5 inside the receiver's framework (decompress_init_multi picks the transcoder for a DXT display codec, priority 900 as
    gpujpeg_to_dxt.cpp:368-373): decoded to packed RGB on the device, block-compressed bottom-up (negative height) with the CUDA kernels'
    rounding -- the DXT oracle run on the decode oracle's picture."""
    w, h = 192, 96
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    src = rgb.ravel() if codec == "RGB" else po.convert_frame("RGB", "UYVY", rgb, w, h)
    raw, jpg, dec = tmp_path / "in.raw", tmp_path / "f.jpg", tmp_path / "out.dxt"
    np.ascontiguousarray(src).tofile(raw)
    assert _run([cfg, codec, w, h, raw, jpg]).returncode == 0
    r = subprocess.run([DEC_HARNESS, "JPEG", out, str(w), str(h), str(jpg), str(dec)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    _, crop, _ = po.jpeg_decode_planes(jpg.read_bytes())
    if codec == "RGB":
        picture = np.stack(crop, -1).ravel()
    else:
        picture = po.convert_frame("UYVY", "RGB", po.planar_to_uyvy(*crop, w, h, chroma=420 if "420" in cfg else 422), w, h)
    want = po.dxt_encode(po.IN_RGB, po.OUT_DXT1 if out == "DXT1" else po.OUT_DXT5YCOCG, picture, w, -h, ties="away")
    got = np.fromfile(dec, np.uint8)
    assert np.array_equal(got, want)
    # and it is a picture: decoded again (flipped back) it is close to what went in
    back = po.dxt_decode(po.OUT_DXT1 if out == "DXT1" else po.OUT_DXT5YCOCG, "RGB", got, w, h).reshape(h, w, 3)[::-1]
    assert 10 * np.log10(255.0 ** 2 / np.mean((back.astype(float) - rgb.astype(float)) ** 2)) > 28


REF_UNIT_TEST = os.path.join(ROOT, "oracle", "_ref", "ug_ref_unit_test")


@pytest.mark.skipif(not os.path.exists(REF_UNIT_TEST), reason="oracle/_ref/ug_ref_unit_test not built")
@pytest.mark.gpu
def test_the_reference_s_own_gpujpeg_unit_test_passes():
    """test/gpujpeg_test.cpp of the reference, compiled unmodified from the reference tree and linked with the reference's compress and
    decompress frameworks, run against this repository's modules under the names it asks for (`GPUJPEG:check`, `GPUJPEG`,
    `--param decompress=gpujpeg`): compress a 1920x1080 RGB frame, decompress it to RGB, every byte within 1 of the input."""
    r = subprocess.run([REF_UNIT_TEST], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "gpujpeg_test_simple: PASSED" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(REF_UNIT_TEST), reason="oracle/_ref/ug_ref_unit_test not built")
def test_the_reference_unit_test_skips_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([REF_UNIT_TEST], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "SKIPPED" in r.stdout, r.stdout + r.stderr


# ---- the PLUGIN route of the boundary (VERDICT r2 missing #4): modules that arrive by dlopen only ---------------------------------------
PLUGIN_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ug_plugin_harness")
PLUGINS = ["ultragrid_vcompress_dxt.so", "ultragrid_vcompress_jpeg.so", "ultragrid_vdecompress_dxt_mi355x.so", "ultragrid_vdecompress_jpeg_mi355x.so",
           "ultragrid_vdecompress_jpeg_to_dxt_mi355x.so"]
needs_plugin_harness = pytest.mark.skipif(not os.path.exists(PLUGIN_HARNESS), reason="oracle/_ref/ug_plugin_harness not built (needs /root/reference)")


def _installation(tmp_path):
    """<tmp>/bin/ug_plugin_harness + <tmp>/lib/ultragrid/ultragrid_*.so: the layout open_all() searches relative to argv[0]
    (lib_common.cpp:192-196).  Copies, not links; the kernel library is found through LD_LIBRARY_PATH as in an installation."""
    import shutil
    (tmp_path / "bin").mkdir()
    (tmp_path / "lib" / "ultragrid").mkdir(parents=True)
    shutil.copy(PLUGIN_HARNESS, tmp_path / "bin" / "ug_plugin_harness")
    for p in PLUGINS:
        shutil.copy(os.path.join(ROOT, "oracle", "_ref", p), tmp_path / "lib" / "ultragrid" / p)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "ultragrid_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))

    def run(*args):
        return subprocess.run([str(tmp_path / "bin" / "ug_plugin_harness")] + [str(a) for a in args], capture_output=True, text=True, timeout=60, env=env)
    return run


@needs_plugin_harness
def test_plugins_load_through_open_all_and_register(tmp_path):
    """lib_common.cpp:186-223 (-DBUILD_LIBRARIES): glob + dlopen(RTLD_NOW | RTLD_GLOBAL) of every ultragrid_*.so; RTLD_NOW means every
    undefined symbol of a module resolved against the -rdynamic executable, and the REGISTER_MODULE constructors ran inside dlopen.  The
    harness binary links no module object: without the plugins it knows no compression at all."""
    run = _installation(tmp_path)
    r = run("list")
    assert r.returncode == 0 and "PLUGINS opened=5" in r.stdout, r.stdout + r.stderr
    names = r.stdout.split()
    for n in ("dxt", "jpeg", "gpujpeg", "dxt_mi355x", "jpeg_mi355x", "jpeg_to_dxt_mi355x"):
        assert n in names, (n, r.stdout)
    assert "opening warning" not in r.stdout + r.stderr
    for p in PLUGINS:
        (tmp_path / "lib" / "ultragrid" / p).unlink()
    r = run("list")
    assert r.returncode == 4 and "PLUGINS opened=0" in r.stdout


@needs_plugin_harness
@pytest.mark.gpu
def test_frames_through_modules_that_arrived_by_dlopen(tmp_path, po):
    """One frame each way through modules that exist in the process only because open_all() dlopen'ed them: compress_init("dxt:DXT5") ->
    bit-equal to the oracle; decompress_init_multi(DXT5 -> RGBA) on those blocks -> bit-equal to the decoder oracle; `-c jpeg` then the JPEG
    decompress plugin (probe first) -> the decode oracle's planes."""
    run = _installation(tmp_path)
    w, h = 192, 64
    src = synth.s1_random("UYVY", w, h, salt=21)
    raw, dxt, rgba = tmp_path / "in.raw", tmp_path / "out.dxt", tmp_path / "back.rgba"
    src.tofile(raw)
    r = run("compress", "dxt:DXT5", "UYVY", w, h, raw, dxt)
    assert r.returncode == 0 and "PLUGINS opened=5" in r.stdout and "OK compress codec=DXT5" in r.stdout, r.stdout + r.stderr
    blocks = np.fromfile(dxt, np.uint8)
    assert np.array_equal(blocks, po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, src, w, h))
    r = run("decompress", "DXT5", "RGBA", w, h, dxt, rgba)
    assert r.returncode == 0 and "OK decompress DXT5 -> RGBA" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(rgba, np.uint8), po.dxt_decode(po.OUT_DXT5YCOCG, "RGBA", blocks, w, h))
    jpg, back = tmp_path / "f.jpg", tmp_path / "back.uyvy"
    r = run("compress", "jpeg:q=85:restart=4", "UYVY", w, h, raw, jpg)
    assert r.returncode == 0 and "OK compress codec=JPEG" in r.stdout, r.stdout + r.stderr
    r = run("decompress", "JPEG", "UYVY", w, h, jpg, back)
    assert r.returncode == 0, r.stdout + r.stderr
    _, crop, _ = po.jpeg_decode_planes(jpg.read_bytes())
    assert np.array_equal(np.fromfile(back, np.uint8), po.planar_to_uyvy(*crop, w, h, chroma=422))


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,codec", [("jpeg:q=80:restart=4", "UYVY"), ("jpeg:q=85:restart=2:subsampling=420", "UYVY"), ("jpeg:q=80:restart=4", "RGB"), ("jpeg:q=80", "v210")])
def test_batched_workers_deliver_the_same_streams_in_order(tmp_path, po, cfg, codec):
    """`batch=<n>`: a busy worker queues up to n frames and hands them to ug_hip_jpeg_encoder_encode_batch in one go.
    14 distinct frames pushed back to back through the reference's compress framework: the streams that come out -- and their order -- are
    byte for byte those of the one-frame-per-call module (batch=1, what the reference does), with one worker (batches certainly form)
    and with the default two."""
    w, h, n = 192, 96, 14
    frames = []
    for f in range(n):
        yy, xx = np.mgrid[0:h, 0:w]
        rgb = np.stack([128 + 100 * np.sin(xx / (20.0 + f)) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / (21.0 + f)), 128 + 80 * np.sin(yy / 9.0 + f)], -1).clip(0, 255).astype(np.uint8)
        uyvy = po.convert_frame("RGB", "UYVY", rgb, w, h)
        frames.append({"UYVY": uyvy, "RGB": rgb.ravel(), "v210": po.convert_frame("UYVY", "v210", uyvy, w, h)}[codec])
    raw = tmp_path / "in.raw"
    np.concatenate([np.ascontiguousarray(x).ravel() for x in frames]).tofile(raw)
    outs = {}
    for tag, extra in (("one", ":batch=1:workers=1"), ("b4w1", ":batch=4:workers=1"), ("b16w2", ":batch=16"), ("b3w2", ":batch=3:workers=2")):
        out = tmp_path / f"{tag}.bin"
        r = _run([cfg + extra, codec, w, h, raw, out, 1, "host", n])
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"frames={n}" in r.stdout and "seq=" + ",".join(str(i) for i in range(n)) + "," in r.stdout, r.stdout
        outs[tag] = out.read_bytes()
    assert len(set(outs.values())) == 1, {k: len(v) for k, v in outs.items()}
    assert outs["one"].count(b"\xff\xd8\xff") >= n           # n JPEG streams one after the other
    assert _run([cfg + ":batch=17", codec, w, h, raw, tmp_path / "x", 1, "host", 1]).returncode == 2   # refused at init


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,codec", [("dxt:DXT5", "UYVY"), ("dxt:DXT1", "RGB"), ("dxt:DXT5", "YUYV"), ("dxt:DXT1_YUV", "v210")])
def test_batched_dxt_workers_deliver_the_same_blocks_in_order(tmp_path, po, cfg, codec):
    """`-c dxt:...:batch=<n>`: queued frames go through ONE ug_hip_dxt_encode_batch_ex launch (and the device-side conversions first, where
    the input needs them); blocks and order equal the one-frame module's."""
    w, h, n = 192, 64, 13
    frames = [synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=40 + f) for f in range(n)]
    raw = tmp_path / "in.raw"
    np.concatenate(frames).tofile(raw)
    outs = {}
    for tag, extra in (("one", ":batch=1:workers=1"), ("b4w1", ":batch=4:workers=1"), ("b16w2", ":batch=16")):
        out = tmp_path / f"{tag}.bin"
        r = _run([cfg + extra, codec, w, h, raw, out, 1, "host", n])
        assert r.returncode == 0 and "seq=" + ",".join(str(i) for i in range(n)) + "," in r.stdout, r.stdout + r.stderr
        outs[tag] = np.fromfile(out, np.uint8)
    assert np.array_equal(outs["one"], outs["b4w1"]) and np.array_equal(outs["one"], outs["b16w2"])
    per = outs["one"].size // n
    if codec == "UYVY":
        assert np.array_equal(outs["one"][:per], po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, frames[0], w, h))


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["dxt:DXT5", "jpeg:q=80"])
def test_workers_run_on_the_gpu_s_numa_node(tmp_path, po, cfg):
    """NUMA placement in the PRODUCT.  Every worker thread of the frame sharder binds itself to the CPUs of its GPU's
    NUMA node (ug_hip_bind_thread_to_device) before its encoder state -- and with it the pinned frame pool -- exists; the worker reports
    where it may run (UG_MI355X_NUMA_REPORT) and the test checks that against sysfs.  On a box whose platform names no node (numa_node =
    -1) the worker is left alone; numa=0 switches the binding off; the encoded bytes are the same either way."""
    w, h, n = 192, 64, 6
    frames = [synth.s1_random("UYVY", w, h, salt=70 + f) for f in range(n)]
    raw = tmp_path / "in.raw"
    np.concatenate(frames).tofile(raw)
    env = dict(os.environ, UG_MI355X_NUMA_REPORT="1")
    outs = {}
    for tag, extra in (("numa", ""), ("off", ":numa=0")):
        out = tmp_path / f"{tag}.bin"
        r = _run([cfg + extra, "UYVY", w, h, raw, out, 1, "host", n], env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = out.read_bytes()
        lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("NUMA worker")]
        if tag == "off":
            assert not lines
            continue
        assert len(lines) == 2, r.stdout                      # two workers per device by default
        for l in lines:
            f = dict(kv.split("=") for kv in l.split()[2:])
            assert f["dev"] == "0" and f["rc"] == "0"
            node, bound, aff = int(f["node"]), int(f["bound"]), _cpulist(f["affinity"])
            p = f"/sys/devices/system/node/node{node}/cpulist"
            if node >= 0 and os.path.exists(p):
                want = _cpulist(open(p).read()) & os.sched_getaffinity(0)
                if want:
                    assert aff == want and bound == len(want), (l, sorted(want))
                else:
                    assert bound == 0
            else:                                             # the platform does not say: left where it was
                assert bound == 0 and aff == os.sched_getaffinity(0)
    assert outs["numa"] == outs["off"]


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,codec,w,h", [("dxt:DXT5", "UYVY", 192, 64), ("dxt:DXT1", "RGB", 192, 64), ("dxt:DXT1", "RGBA", 200, 36), ("dxt:DXT5", "v210", 192, 64),
                                           ("dxt:DXT5", "UYVY", 1924, 36), ("dxt:DXT1_YUV", "UYVY", 192, 64), ("dxt:DXT5", "YUYV", 200, 44)])
def test_interlaced_input_is_blended_before_encoding(tmp_path, po, cfg, codec, w, h):
    """on INTERLACED_MERGED input RTDXT -- the module `-c dxt` replaces -- blends the lines of the decoded frame (vc_deinterlace,
    video_codec.c:597-664) before encoding and announces the stream as progressive (dxt_glsl.cpp:195-201,291-293).  The module does the same on
    the device: blocks == encoder(oracle blend(line-decoded frame)), through the one-frame path, the batch path and from a device-resident frame;
    deinterlace=no leaves the frame as it is (and interlaced); widths whose decoded line is no multiple of 16 bytes included."""
    src = synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=11)
    raw = tmp_path / "in.raw"
    np.concatenate([src, src]).tofile(raw)
    env = dict(os.environ, UG_HARNESS_INTERLACING="merged")
    yuv_out = cfg.endswith("DXT1_YUV")
    target = "UYVY" if codec in ("UYVY", "YUYV", "v210") or yuv_out else "RGB"
    decoded = src if codec == target else po.convert_frame(codec, target, src, w, h)
    ls = {"UYVY": 2 * w, "RGB": 3 * w}[target]
    oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1
    pin = {"UYVY": po.IN_UYVY_RAW if yuv_out else po.IN_UYVY, "RGB": po.IN_RGB}[target]
    want = po.dxt_encode(pin, oid, po.deinterlace_blend(decoded, ls, h), w, h)
    plain = po.dxt_encode(pin, oid, decoded, w, h)
    assert not np.array_equal(want, plain)
    for extra, mem, n in (("", "host", 1), (":batch=4:workers=1", "host", 2), ("", "dev", 1)):
        out = tmp_path / "out.bin"
        r = _run([cfg + extra, codec, w, h, raw, out, 1, mem, n], env=env)
        assert r.returncode == 0 and "interlacing=p " in r.stdout, r.stdout + r.stderr
        got = np.fromfile(out, np.uint8)
        assert np.array_equal(got[: want.size], want), (extra, mem)
        if n == 2:
            assert np.array_equal(got[want.size:], want)
    out = tmp_path / "out_no.bin"
    r = _run([cfg + ":deinterlace=no", codec, w, h, raw, out, 1, "host", 1], env=env)
    assert r.returncode == 0 and "interlacing=i " in r.stdout and np.array_equal(np.fromfile(out, np.uint8), plain), r.stdout + r.stderr
    r = _run([cfg, codec, w, h, raw, out, 1, "host", 1])                                   # a progressive source: nothing changes
    assert r.returncode == 0 and np.array_equal(np.fromfile(out, np.uint8), plain)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,codec,w,h", [("dxt:DXT5", "UYVY", 1920, 1080), ("dxt:DXT5", "v210", 3840, 2160), ("dxt:DXT1", "RGB", 1280, 720), ("dxt:DXT1", "RGBA", 200, 36),
                                           ("dxt:DXT1_YUV", "v210", 96, 48), ("dxt:DXT5", "YUYV", 200, 44), ("dxt:DXT1", "UYVY", 1924, 1084), ("dxt:DXT5", "R10k", 192, 64)])
def test_row_bands_give_the_bytes_of_the_whole_frame(tmp_path, po, cfg, codec, w, h):
    """VERDICT r4 next #6 / SURVEY.md 8(e) "tile-level split of a single frame": bands=<k> uploads, encodes and downloads ONE frame as k row bands that
    overlap each other.  Blocks and line converters are band-independent, so the bytes must be those of bands=1 -- which are the oracle's -- for every k,
    with pinned and pageable source frames, for fused and pre-converted inputs, for heights that are no multiple of 16 (1084: the last band takes the
    rest), for pictures with fewer 16-line units than bands, and for interlaced input (de-interlaced whole: bands do not apply)."""
    src = _random_frame(po, codec, w, h, 5) if codec in ("R10k",) else synth.s1_random(codec if codec != "YUYV" else "UYVY", w, h, salt=3)
    raw = tmp_path / "in.raw"
    np.concatenate([src, src]).tofile(raw)
    outs = {}
    for k in (1, 2, 4, 7, 16):
        for pinned in (False, True):
            out = tmp_path / f"o{k}{int(pinned)}.bin"
            env = dict(os.environ, **({"UG_HARNESS_PINNED": "1"} if pinned else {}))
            r = _run([f"{cfg}:bands={k}:workers=1", codec, w, h, raw, out, 1, "host", 2], env=env)
            assert r.returncode == 0, r.stdout + r.stderr
            outs[(k, pinned)] = out.read_bytes()
    assert len(set(outs.values())) == 1
    if codec in ("UYVY", "v210", "RGB", "RGBA") and not cfg.endswith("DXT1_YUV"):
        oid = po.OUT_DXT5YCOCG if cfg.endswith("DXT5") else po.OUT_DXT1
        want = po.dxt_encode({"UYVY": po.IN_UYVY, "v210": po.IN_V210, "RGB": po.IN_RGB, "RGBA": po.IN_RGBA}[codec], oid, src, w, h).tobytes()
        assert outs[(4, True)] == want + want
    assert _run(["dxt:bands=0", codec, w, h, raw, tmp_path / "x"]).returncode == 2 and _run(["dxt:bands=17", codec, w, h, raw, tmp_path / "x"]).returncode == 2
    if codec == "UYVY" and h == 1080:
        env = dict(os.environ, UG_HARNESS_INTERLACING="merged")
        a, b = tmp_path / "ia.bin", tmp_path / "ib.bin"
        assert _run([cfg + ":bands=4", codec, w, h, raw, a, 1, "host", 1], env=env).returncode == 0 and _run([cfg, codec, w, h, raw, b, 1, "host", 1], env=env).returncode == 0
        assert a.read_bytes() == b.read_bytes() != outs[(1, False)][: len(b.read_bytes())]


@needs_dec_harness
@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["ug_dec_harness", "ug_dec_harness_asan", "ug_dec_harness_tsan"])
def test_receiver_tile_fanout_decodes_all_tiles_at_once(tmp_path, po, binary):
    """The receive side's concurrency convention (rtp/video_decoders.cpp:590-612,676-690): one decompress state per tile from ONE decompress_init_multi(..., n),
    decompress_frame of all tiles at the same time on worker threads -- four states of each module on one GPU, sharing the process-wide copy lanes, 20 frames
    each: every output equals the single-state result (which the other tests pin to the oracle).  Plain, under ASan + UBSan and under ThreadSanitizer."""
    import shutil
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{binary} not built")
    w, h = 640, 368
    uyvy = synth.s2_video("UYVY", w, h)
    raw, jpg, dxt5, dxt1 = (tmp_path / n for n in ("in.raw", "f.jpg", "f.dxt5", "f.dxt1"))
    uyvy.tofile(raw)
    assert _run(["jpeg:q=85:restart=4", "UYVY", w, h, raw, jpg]).returncode == 0
    po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, uyvy, w, h).tofile(dxt5)
    po.dxt_encode(po.IN_UYVY, po.OUT_DXT1, uyvy, w, h).tofile(dxt1)
    env = dict(os.environ, UG_DEC_TILES="4", ASAN_OPTIONS="detect_leaks=0 exitcode=66 protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1 exitcode=66 suppressions=" + os.path.join(ROOT, "ultragrid_amd", "module", "tsan_gpu.supp"))
    pre = ["setarch", "x86_64", "-R"] if binary.endswith("_tsan") and shutil.which("setarch") else []
    for comp, out, f in (("DXT5", "RGBA", dxt5), ("DXT5", "UYVY", dxt5), ("DXT1", "RGB", dxt1), ("JPEG", "UYVY", jpg), ("JPEG", "RGBA", jpg), ("JPEG", "DXT1", jpg), ("JPEG", "DXT5", jpg)):
        r = subprocess.run(pre + [exe, comp, out, str(w), str(h), str(f), str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=300, env=env)
        text = r.stdout + r.stderr
        if "unexpected memory mapping" in text or "ReserveShadowMemoryRange failed" in text or "Shadow memory range interleaves" in text:
            pytest.skip("this box's address-space layout cannot host the sanitizer runtime")
        assert "Sanitizer" not in text and "runtime error" not in text, text[-5000:]
        assert r.returncode == 0 and "TILES n=4 rounds=20 OK" in r.stdout, (comp, out, text[-2000:])


# ---------------------------------------------------------------------------------------------------------------------------------------
# Round 6: frame sizes that are not multiples of 4 through the modules (dxt_glsl.cpp:150-160 takes any tile size), the receivers' device
# choice (--param mi355x-device / -D, mi355x_receiver.h), their band pipeline and the 2-D copy of a display pitch
# ---------------------------------------------------------------------------------------------------------------------------------------
@needs_dec_harness
@pytest.mark.parametrize("param,want", [("", "DEVICES n=1: 0 | states: 0 0 0 0 0 | bands=0"), ("mi355x-device=2", "DEVICES n=1: 2 | states: 2 2 2 2 2 | bands=0"),
                                        ("mi355x-device=0:1:3", "DEVICES n=3: 0 1 3 | states: 0 1 3 0 1 | bands=0"),
                                        ("decompress=dxt_mi355x,mi355x-device=1+1,mi355x-bands=8", "DEVICES n=2: 1 1 | states: 1 1 1 1 1 | bands=8"),
                                        ("mi355x-bands=0", "DEVICES n=1: 0 | states: 0 0 0 0 0 | bands=1"), ("mi355x-bands=99", "DEVICES n=1: 0 | states: 0 0 0 0 0 | bands=16")])
def test_receiver_device_parameter_parsing(param, want):
    """CPU: what the decompress modules derive from `--param mi355x-device=<n>[:<n>...]` (':' or '+' between the numbers -- ',' separates --param
    entries, host.cpp:1098-1100) and `mi355x-bands=<k>`; the states of a process take the listed devices in turn."""
    r = subprocess.run([DEC_HARNESS, "devices"], capture_output=True, text=True, timeout=30, env={**os.environ, "UG_PARAM": param})
    assert r.returncode == 0 and r.stdout.strip() == want, r.stdout + r.stderr


@needs_dec_harness
@pytest.mark.parametrize("param", ["mi355x-device=0;1", "mi355x-device=", "mi355x-device=0:-1", "mi355x-device=x", "mi355x-device=0:", "mi355x-device=4096"])
def test_receiver_device_parameter_malformed(param):
    r = subprocess.run([DEC_HARNESS, "devices"], capture_output=True, text=True, timeout=30, env={**os.environ, "UG_PARAM": param})
    assert r.returncode == 4 and "DEVICES BAD" in r.stdout, r.stdout + r.stderr


_PIN = {"UYVY": "IN_UYVY", "v210": "IN_V210", "RGB": "IN_RGB", "RGBA": "IN_RGBA"}


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("codec,w,h", [("UYVY", 198, 70), ("UYVY", 1366, 38), ("v210", 198, 70), ("v210", 1366, 10), ("RGB", 199, 33), ("RGB", 1366, 7), ("RGBA", 197, 34),
                                       ("BGR", 50, 21), ("YUYV", 14, 9)])
@pytest.mark.parametrize("cfg", ["dxt:DXT5", "dxt:DXT1", "dxt:DXT5:bands=3"])
def test_compress_frame_sizes_that_are_not_multiples_of_4(tmp_path, po, codec, w, h, cfg):
    """-c dxt takes what -c RTDXT takes (VERDICT r5 "What's missing" #3): the stream holds (w+3)/4 x (h+3)/4 blocks, bytes == the oracle's (which
    is pinned to the executed reference shaders at such sizes, tests/test_oracle_dxt.py); bands=<k> cuts such frames too (16-line edges)."""
    base = {"YUYV": "UYVY", "BGR": "RGB"}.get(codec, codec)
    src = synth.s1_random(base, w, h, salt=w)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    src.tofile(raw)
    r = _run([cfg, codec, w, h, raw, out])
    assert r.returncode == 0, r.stdout + r.stderr
    oid = po.OUT_DXT5YCOCG if "DXT5" in cfg else po.OUT_DXT1
    if codec == "YUYV":
        want = po.dxt_encode(po.IN_UYVY, oid, po.convert_frame("YUYV", "UYVY", src, w, h), w, h)
    elif codec == "BGR":
        want = po.dxt_encode(po.IN_RGB, oid, po.convert_frame("BGR", "RGB", src, w, h), w, h)
    else:
        want = po.dxt_encode(getattr(po, _PIN[codec]), oid, src, w, h)
    got = np.fromfile(out, np.uint8)
    assert got.size == want.size == po.dxt_size(oid, w, h) and np.array_equal(got, want)


@needs_harness
@pytest.mark.gpu
def test_odd_width_422_is_refused_by_the_module(tmp_path):
    raw = tmp_path / "in.raw"
    np.zeros(16 * 8 * 2, np.uint8).tofile(raw)
    r = _run(["dxt:DXT5", "UYVY", 15, 8, raw, tmp_path / "o.bin"])
    assert r.returncode != 0 and "pixel pairs" in (r.stdout + r.stderr)


def _psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / max(1e-9, np.mean((a.astype(float) - b.astype(float)) ** 2)))


@needs_harness
@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(720, 486), (2048, 858)])
@pytest.mark.parametrize("comp", ["DXT5", "DXT1"])
def test_ntsc_and_2k_scope_round_trip_through_both_frameworks(tmp_path, po, w, h, comp):
    """720x486 (NTSC) and 2048x858 (2K scope) UYVY: ug_harness (-c dxt) -> ug_dec_harness (dxt_mi355x), the reference's compress and
    decompress frameworks on either side; bytes == the oracles', and the picture is as good as the same content cut to multiples of 4."""
    src = synth.s2_video("UYVY", w, h, salt=2)
    raw, dxt, dec = tmp_path / "in.raw", tmp_path / "f.dxt", tmp_path / "out.raw"
    src.tofile(raw)
    assert _run([f"dxt:{comp}", "UYVY", w, h, raw, dxt]).returncode == 0
    oid = po.OUT_DXT5YCOCG if comp == "DXT5" else po.OUT_DXT1
    blocks = np.fromfile(dxt, np.uint8)
    assert np.array_equal(blocks, po.dxt_encode(po.IN_UYVY, oid, src, w, h, threads=0))
    r = subprocess.run([DEC_HARNESS, comp, "UYVY", str(w), str(h), str(dxt), str(dec)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    back = np.fromfile(dec, np.uint8)
    assert np.array_equal(back, po.dxt_decode(oid, "UYVY", blocks, w, h))
    wa, ha = w // 4 * 4, h // 4 * 4
    crop = np.ascontiguousarray(src.reshape(h, 2 * w)[:ha, :2 * wa])
    back_a = po.dxt_decode(oid, "UYVY", po.dxt_encode(po.IN_UYVY, oid, crop, wa, ha, threads=0), wa, ha)
    assert _psnr(back, src) >= _psnr(back_a, crop.ravel()) - 0.05


@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("param", ["", "mi355x-bands=1", "mi355x-bands=7", "mi355x-device=0,mi355x-bands=16"])
@pytest.mark.parametrize("comp,out,w,h", [("DXT5", "RGBA", 198, 70), ("DXT1", "UYVY", 1366, 166), ("DXT5", "RGB", 199, 133), ("DXT1_YUV", "RGBA", 64, 486), ("DXT5", "UYVY", 256, 130)])
def test_decompress_unaligned_sizes_bands_and_pitch(tmp_path, po, comp, out, w, h, param):
    """dxt_mi355x at sizes that are not multiples of 4, with the band pipeline at several band counts and with a display pitch: bytes == the
    decode oracle's whatever the cut (arbitrary block contents: every decoder path)"""
    oid = {"DXT1": po.OUT_DXT1, "DXT1_YUV": po.OUT_DXT1_YUV, "DXT5": po.OUT_DXT5YCOCG}[comp]
    blocks = np.random.default_rng(w + h).integers(0, 256, po.dxt_size(po.OUT_DXT1 if comp != "DXT5" else oid, w, h), dtype=np.uint8)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    blocks.tofile(src)
    want = po.dxt_decode(oid, out, blocks, w, h)
    ls = po.linesize(w, out)
    for pitch in (ls, ls + 64, ls + 4):
        r = subprocess.run([DEC_HARNESS, comp, out, str(w), str(h), str(src), str(dst), str(pitch)], capture_output=True, text=True, timeout=30,
                           env={**os.environ, "UG_PARAM": param})
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(dst, np.uint8).reshape(h, pitch)
        assert np.array_equal(got[:, :ls].ravel(), want), (pitch, param)
        assert not got[:, ls:].any()          # the gap between the lines is not touched (the harness hands over a zeroed buffer)


@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("comp,out", [("DXT5", "RGBA"), ("DXT1", "UYVY")])
def test_decompress_4k_with_a_display_pitch(tmp_path, po, comp, out):
    """the pitched path at 4K (VERDICT r5 "What's weak" #3a): one 2-D copy per band instead of 2160 copies; bytes == the packed result"""
    w, h = 3840, 2160
    oid = po.OUT_DXT5YCOCG if comp == "DXT5" else po.OUT_DXT1
    one = po.dxt_encode(po.IN_UYVY, oid, synth.s2_video("UYVY", w, 240, salt=1), w, 240, threads=0)
    blocks = np.tile(one, h // 240)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    blocks.tofile(src)
    ls = po.linesize(w, out)
    packed = None
    for pitch in (ls, ls + 64):
        r = subprocess.run([DEC_HARNESS, comp, out, str(w), str(h), str(src), str(dst), str(pitch)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(dst, np.uint8).reshape(h, pitch)[:, :ls]
        if packed is None:
            packed = got.copy()
            band = po.dxt_decode(oid, out, one, w, 240).reshape(240, ls)
            assert np.array_equal(packed[:240], band) and np.array_equal(packed[-240:], band)
        else:
            assert np.array_equal(got, packed)


@needs_harness
@needs_dec_harness
@pytest.mark.gpu
@pytest.mark.parametrize("param,delay", [("", 0), ("mi355x-device=0", 0), ("mi355x-device=0:0", 1), ("mi355x-device=0+0+0", 2)])
@pytest.mark.parametrize("w,h,out", [(192, 96, "DXT5"), (198, 102, "DXT1"), (198, 102, "DXT5")])
def test_transcoder_frame_rotation_over_listed_devices(tmp_path, po, param, delay, w, h, out):
    """jpeg_to_dxt_mi355x with the GPU listed once, twice, three times (gpujpeg_to_dxt.cpp:187-212,305-328: one worker per listed device, frame k
    comes out when frame k + N - 1 goes in): N - 1 calls answer DECODER_NO_FRAME first, then every frame comes out, in order, with the bytes of
    the undelayed module -- 40 more frames are pushed through and compared.  Also at a frame size that is not a multiple of 4."""
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([128 + 100 * np.sin(xx / 20.0) * np.cos(yy / 15.0), 128 + 90 * np.cos(xx / 33.0 + yy / 21.0), 128 + 80 * np.sin(yy / 9.0)], -1).clip(0, 255).astype(np.uint8)
    raw, jpg, dec = tmp_path / "in.raw", tmp_path / "f.jpg", tmp_path / "out.dxt"
    po.convert_frame("RGB", "UYVY", rgb, w, h).tofile(raw)
    assert _run(["jpeg:q=85:restart=4", "UYVY", w, h, raw, jpg]).returncode == 0
    r = subprocess.run([DEC_HARNESS, "JPEG", out, str(w), str(h), str(jpg), str(dec), str(w // (2 if out == "DXT1" else 1))], capture_output=True, text=True, timeout=60,
                       env={**os.environ, "UG_PARAM": param, "UG_DEC_REPEAT": "40"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert (f"DELAY frames={delay}" in r.stdout) == (delay > 0), r.stdout
    _, crop, _ = po.jpeg_decode_planes(jpg.read_bytes())
    picture = po.convert_frame("UYVY", "RGB", po.planar_to_uyvy(*crop, w, h, chroma=422), w, h)
    oid = po.OUT_DXT1 if out == "DXT1" else po.OUT_DXT5YCOCG
    want = po.dxt_encode(po.IN_RGB, oid, picture, w, -h, ties="away")
    got = np.fromfile(dec, np.uint8)
    assert got.size == po.dxt_size(oid, w, h) and np.array_equal(got, want)


@needs_dec_harness
@pytest.mark.gpu
def test_receiver_refuses_an_unusable_device(tmp_path):
    src = tmp_path / "in.bin"
    np.zeros(64 * 16, np.uint8).tofile(src)
    r = subprocess.run([DEC_HARNESS, "DXT5", "RGBA", "64", "16", str(src), str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=30,
                       env={**os.environ, "UG_PARAM": "mi355x-device=63"})
    assert r.returncode != 0 and "device 63" in (r.stdout + r.stderr)


@needs_dec_harness
@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["ug_dec_harness_asan", "ug_dec_harness_tsan"])
def test_receiver_threads_under_sanitizers(tmp_path, po, binary):
    """The two places where a receiver module has threads of its own, under ASan + UBSan and under ThreadSanitizer, 30 frames each: dxt_mi355x's band
    pipeline (the downloader thread beside the caller's, 8 bands) and jpeg_to_dxt_mi355x's frame rotation (one worker per listed device, the GPU
    listed three times); every frame equals the first, which equals the plain binary's."""
    import shutil
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{binary} not built")
    w, h = 640, 368
    uyvy = synth.s2_video("UYVY", w, h, salt=3)
    raw, jpg, dxt5 = (tmp_path / n for n in ("in.raw", "f.jpg", "f.dxt5"))
    uyvy.tofile(raw)
    assert _run(["jpeg:q=85:restart=4", "UYVY", w, h, raw, jpg]).returncode == 0
    po.dxt_encode(po.IN_UYVY, po.OUT_DXT5YCOCG, uyvy, w, h).tofile(dxt5)
    env = dict(os.environ, UG_DEC_REPEAT="30", ASAN_OPTIONS="detect_leaks=0 exitcode=66 protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1 exitcode=66 suppressions=" + os.path.join(ROOT, "ultragrid_amd", "module", "tsan_gpu.supp"))
    pre = ["setarch", "x86_64", "-R"] if binary.endswith("_tsan") and shutil.which("setarch") else []
    for comp, out, f, param, pitch in (("DXT5", "UYVY", dxt5, "mi355x-bands=8", 2 * w + 64), ("DXT5", "RGBA", dxt5, "mi355x-bands=3", 4 * w), ("JPEG", "DXT5", jpg, "mi355x-device=0:0:0", w)):
        plain, san = tmp_path / "plain.raw", tmp_path / "san.raw"
        r0 = subprocess.run([DEC_HARNESS, comp, out, str(w), str(h), str(f), str(plain), str(pitch)], capture_output=True, text=True, timeout=60)
        assert r0.returncode == 0, r0.stdout + r0.stderr
        r = subprocess.run(pre + [exe, comp, out, str(w), str(h), str(f), str(san), str(pitch)], capture_output=True, text=True, timeout=300, env=dict(env, UG_PARAM=param))
        text = r.stdout + r.stderr
        if "unexpected memory mapping" in text or "ReserveShadowMemoryRange failed" in text or "Shadow memory range interleaves" in text:
            pytest.skip("this box's address-space layout cannot host the sanitizer runtime")
        assert "Sanitizer" not in text and "runtime error" not in text, text[-5000:]
        assert r.returncode == 0 and "THROUGHPUT frames=30" in r.stdout, (comp, out, text[-2000:])
        assert plain.read_bytes() == san.read_bytes(), (comp, out)
