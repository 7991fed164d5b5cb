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


@needs_harness
@pytest.mark.gpu
def test_unsupported_input_is_refused_not_faked(tmp_path):
    raw = tmp_path / "in.raw"
    np.zeros(64 * 16 * 6, np.uint8).tofile(raw)
    r = _run(["dxt:DXT5", "RG48", 64, 16, raw, tmp_path / "o.bin"])
    assert r.returncode == 3 and "Unsupported codec" in (r.stdout + r.stderr)  # frame dropped, no CPU fallback


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


SHARDER_TEST = os.path.join(ROOT, "oracle", "_ref", "ug_sharder_test")


@pytest.mark.skipif(not os.path.exists(SHARDER_TEST), reason="oracle/_ref/ug_sharder_test not built")
@pytest.mark.parametrize("workers,frames", [(1, 40), (4, 300), (8, 500)])
def test_frame_sharder_cpu(workers, frames):
    """mi355x::frame_sharder with a fake tile encoder (random delays, injected failures, tiled frames): in-order delivery, failed
    frames skipped, all workers used, metadata kept, the poison pill ends the stream (module/ug_sharder_test.cpp).  No GPU."""
    r = subprocess.run([SHARDER_TEST, str(workers), str(frames)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


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
        assert data == write_jpeg(w, h, ql, qc, *coefs, restart=4, sub=444)
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
    assert _run(["jpeg:subsampling=444", "UYVY", 64, 64, raw, tmp_path / "o"]).returncode != 0   # 4:4:4 needs RGB-family input
    assert _run(["jpeg:subsampling=411", "UYVY", 64, 64, raw, tmp_path / "o"]).returncode != 0


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
