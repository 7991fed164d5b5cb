"""vc_deinterlace (src/video_codec.c:597-664): the oracle's restatement against the compiled reference (CPU), the HIP kernel against the
oracle (GPU), the `-c dxt` module on INTERLACED_MERGED input through the reference's compress framework (GPU, tests/test_module_harness.py)."""
import ctypes as C

import numpy as np
import pytest

from ultragrid_amd import synth

# line size (bytes), lines: whole 16-byte columns, line sizes that are no multiple of 16 (the last column reaches into the next line),
# odd and tiny heights (below 5 lines nothing happens), real frame sizes
GEOMETRIES = [(64, 10), (3840, 1080), (5760, 1080), (200, 9), (24, 7), (40, 12), (16, 4), (16, 5), (136, 33), (7680, 64), (2 * 1924, 35), (3 * 1924, 36), (52, 6)]


def test_oracle_equals_the_compiled_reference(po):
    if not po.have_ref():
        pytest.skip("oracle/_ref/libugref.so not built")
    rng = np.random.default_rng(20260925)
    for ls, lines in GEOMETRIES:
        f = rng.integers(0, 256, ls * lines + 64, dtype=np.uint8)
        assert np.array_equal(po.deinterlace_blend(f, ls, lines), po.ref_deinterlace(f, ls, lines)), (ls, lines)
    # an unaligned start (the reference switches to its movdqu body: same arithmetic)
    f = rng.integers(0, 256, 3840 * 100 + 80, dtype=np.uint8)
    buf = np.zeros(f.size + 16, np.uint8)
    for off in (4, 7):
        view = buf[off: off + f.size]
        view[:] = f
        r = po.ref()
        r.vc_deinterlace.restype = None
        r.vc_deinterlace.argtypes = [C.c_void_p, C.c_long, C.c_int]
        r.vc_deinterlace(view.ctypes.data, 3840, 100)
        assert np.array_equal(view, po.deinterlace_blend(f, 3840, 100))


def test_filter_properties(po):
    """what the blend is: flat pictures stay flat, line 0 and the last lines stay as they are, a single bright line is spread downwards only"""
    ls, lines = 64, 20
    flat = np.full(ls * lines, 93, np.uint8)
    assert np.array_equal(po.deinterlace_blend(flat, ls, lines), flat)
    f = np.random.default_rng(2).integers(0, 256, ls * lines, dtype=np.uint8)
    out = po.deinterlace_blend(f, ls, lines).reshape(lines, ls)
    assert np.array_equal(out[0], f.reshape(lines, ls)[0]) and np.array_equal(out[lines - 3:], f.reshape(lines, ls)[lines - 3:])
    spike = np.zeros((lines, ls), np.uint8)
    spike[6] = 255
    out = po.deinterlace_blend(spike.ravel(), ls, lines).reshape(lines, ls)
    assert not out[:5].any() and out[5].any() and out[6].any() and out[9].any()


@pytest.mark.gpu
def test_gpu_deinterlace_bit_exact(hip, po):
    import torch
    from ultragrid_amd import lib as L
    l = L.load()
    rng = np.random.default_rng(7)
    for ls, lines in GEOMETRIES + [(7680, 2160)]:
        f = rng.integers(0, 256, ls * lines, dtype=np.uint8)
        dev = torch.from_numpy(np.concatenate([f, np.full(64, 0xA5, np.uint8)])).cuda()
        assert l.ug_hip_deinterlace_blend(dev.data_ptr(), ls, lines, None) == 0, L.last_error()
        got = dev.cpu().numpy()
        assert np.array_equal(got[: f.size], po.deinterlace_blend(f, ls, lines)), (ls, lines)
        assert (got[f.size:] == 0xA5).all()                                  # nothing behind the frame is touched
    # the kernel cuts a column into segments that run side by side and chains their summaries (csrc/deinterlace.hip): every way the steps
    # of a picture can be dealt to 16 waves of <= 34 steps, one round and several, on content that sits on the thresholds of the summary
    # (two-valued pictures, nearly flat ones) as well as noise
    for lines in list(range(5, 84)) + [1087, 1091, 1092, 1093, 1095, 2179, 2183, 2185, 3300]:
        for ls in ((68, 37) if lines < 84 else (80,)):
            kind = (lines + ls) % 3
            if kind == 0:
                f = rng.integers(0, 256, ls * lines, dtype=np.uint8)
            elif kind == 1:
                f = (rng.integers(0, 2, ls * lines, dtype=np.uint8) * 255).astype(np.uint8)
            else:
                f = np.clip(rng.integers(0, 256) + rng.integers(-1, 2, ls * lines), 0, 255).astype(np.uint8)
            dev = torch.from_numpy(f.copy()).cuda()
            assert l.ug_hip_deinterlace_blend(dev.data_ptr(), ls, lines, None) == 0, L.last_error()
            assert np.array_equal(dev.cpu().numpy(), po.deinterlace_blend(f, ls, lines)), (ls, lines, kind)
    # video content, three frames per launch, frames 4096 bytes further apart than they are long
    w, h, n = 1920, 1080, 3
    fr = [synth.s2_video("UYVY", w, h, salt=s) for s in range(n)]
    stride = 2 * w * h + 4096
    buf = np.full(n * stride, 0x5A, np.uint8)
    for i, x in enumerate(fr):
        buf[i * stride: i * stride + x.size] = x
    dev = torch.from_numpy(buf).cuda()
    assert l.ug_hip_deinterlace_blend_batch(dev.data_ptr(), 2 * w, h, n, stride, None) == 0
    got = dev.cpu().numpy()
    for i, x in enumerate(fr):
        assert np.array_equal(got[i * stride: i * stride + x.size], po.deinterlace_blend(x, 2 * w, h))
        assert (got[i * stride + x.size: (i + 1) * stride] == 0x5A).all()
    assert l.ug_hip_deinterlace_blend(None, 64, 10, None) == L.EINVAL and l.ug_hip_deinterlace_blend(dev.data_ptr(), 0, 10, None) == L.EINVAL
    # a line shorter than one 16-byte column is refused (the reference's columns overlap themselves there); below 5 lines nothing happens at any size
    assert l.ug_hip_deinterlace_blend(dev.data_ptr(), 15, 10, None) == L.EINVAL and l.ug_hip_deinterlace_blend(dev.data_ptr(), 8, 4, None) == 0
