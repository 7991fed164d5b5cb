"""Sanitizer targets (the reference ships none; "builder should add ASan/UBSan/TSan to its own harness"), built by
`make -C ultragrid_amd/module sanitize` into oracle/_ref/ and run here in the CPU suite:
  ug_sharder_test_tsan   the frame sharder (mi355x_frame_sharder.h) + the reference's video_frame / vf_split objects under ThreadSanitizer
  ug_jpeg_parser_asan    the host half of csrc/jpeg_decode.hip -- the header parser that meets network bytes -- under ASan + UBSan
  ug_dec_harness_asan    the three plain-C decompress shims behind the reference's video_decompress.c under ASan + UBSan"""
import io
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
TSAN, PARSER, DEC = (os.path.join(REF, n) for n in ("ug_sharder_test_tsan", "ug_jpeg_parser_asan", "ug_dec_harness_asan"))


def _clean(r):
    out = r.stdout + r.stderr
    assert "Sanitizer" not in out and "runtime error" not in out, out[-4000:]
    return out


@pytest.mark.skipif(not os.path.exists(TSAN), reason="oracle/_ref/ug_sharder_test_tsan not built (needs /root/reference)")
@pytest.mark.parametrize("workers,frames,batch", [(4, 300, 1), (8, 300, 4), (2, 200, 8)])
def test_frame_sharder_under_thread_sanitizer(workers, frames, batch):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66 second_deadlock_stack=1")
    r = subprocess.run([TSAN, str(workers), str(frames), str(batch)], capture_output=True, text=True, timeout=300, env=env)
    out = _clean(r)
    assert r.returncode == 0 and out.startswith("OK"), out[-3000:]


@pytest.mark.skipif(not os.path.exists(PARSER), reason="oracle/_ref/ug_jpeg_parser_asan not built")
def test_jpeg_header_parser_under_address_and_ub_sanitizer(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(20260925)
    img = (rng.random((48, 64, 3)) * 255).astype(np.uint8)
    seeds = []
    for i, kw in enumerate((dict(quality=80, subsampling=1), dict(quality=90, subsampling=2, restart_marker_blocks=2), dict(quality=70, subsampling=0, optimize=True))):
        p = tmp_path / f"seed{i}.jpg"
        Image.fromarray(img).save(p, "JPEG", **kw)
        seeds.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0 abort_on_error=0 exitcode=66", UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    r = subprocess.run([PARSER, "60000", *seeds], capture_output=True, text=True, timeout=600, env=env)
    out = _clean(r)
    assert r.returncode == 0 and "OK iterations=60000" in out, out[-3000:]


@pytest.mark.skipif(not os.path.exists(DEC), reason="oracle/_ref/ug_dec_harness_asan not built (needs /root/reference)")
def test_decompress_shims_under_address_and_ub_sanitizer(tmp_path):
    """Without a GPU the shims can register, report their priorities, be initialised and refuse to decode -- every line of that under ASan + UBSan;
    with one (the GPU box runs the CPU suite of the driver's fresh checkout too) frames really pass through them."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0 exitcode=66 protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    r = subprocess.run([DEC, "list"], capture_output=True, text=True, timeout=120, env=env)
    out = _clean(r)
    assert r.returncode == 0 and "dxt_mi355x" in out and "jpeg_mi355x" in out, out
    w, h = 64, 16
    blocks = np.random.default_rng(1).integers(0, 256, w * h, dtype=np.uint8)
    src, dst = tmp_path / "in.bin", tmp_path / "out.raw"
    blocks.tofile(src)
    for comp, out_codec in (("DXT5", "RGBA"), ("DXT1", "UYVY")):
        data = blocks if comp == "DXT5" else blocks[: w * h // 2]
        data.tofile(src)
        r = subprocess.run([DEC, comp, out_codec, str(w), str(h), str(src), str(dst)], capture_output=True, text=True, timeout=120, env=env)
        _clean(r)                                        # whatever the exit code (no GPU here: the module refuses), no sanitizer report
    b = io.BytesIO()
    from PIL import Image
    Image.fromarray((np.random.default_rng(2).random((h, w, 3)) * 255).astype(np.uint8)).save(b, "JPEG", quality=80, subsampling=1, restart_marker_blocks=1)
    src.write_bytes(b.getvalue())
    r = subprocess.run([DEC, "JPEG", "UYVY", str(w), str(h), str(src), str(dst)], capture_output=True, text=True, timeout=120, env=env)
    _clean(r)
