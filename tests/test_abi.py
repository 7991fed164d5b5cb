"""CPU: the C-ABI library loads and exports every symbol include/ug_mi355x.h declares (no compute)."""
import ctypes as C
import os
import re

from ultragrid_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "ug_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(ug_hip_[a-z0-9_]+)\s*\(", hdr))


def test_header_and_binding_agree():
    assert _declared() == set(lib.SYMBOLS), _declared() ^ set(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    l = lib.load()  # raises if the .so or a symbol is missing
    for name in _declared():
        assert getattr(l, name) is not None
    assert l.ug_hip_abi_version() == 3


def test_no_torch_or_cxx_types_in_the_abi():
    hdr = open(os.path.join(ROOT, "include", "ug_mi355x.h")).read()
    assert "std::" not in hdr and "torch" not in hdr and "at::" not in hdr
    assert 'extern "C"' in hdr


def test_error_paths_without_a_gpu():
    l = lib.load()
    # argument validation happens before any device call (cuda_dxt.cu:745 semantics: -1)
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, 16, 16, 18, 4, 0, None) == lib.EINVAL       # width % 4
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, 16, 16, 16, 6, 0, None) == lib.EINVAL       # height % 4
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, 8, 16, 16, 4, 0, None) == lib.EINVAL        # src alignment
    assert l.ug_hip_dxt_encode(lib.PF_RGB, lib.DXT1, None, 16, 16, 4, 0, None) == lib.EINVAL     # NULL
    assert l.ug_hip_dxt_encode(lib.PF_RG48, lib.DXT1, 16, 16, 16, 4, 0, None) == lib.EUNSUPP
    assert l.ug_hip_dxt_encode(lib.PF_V210, lib.DXT1, 16, 16, 16, 4, 40, None) == lib.EINVAL     # v210 pitch % 16
    assert l.ug_hip_pixfmt_convert(lib.PF_RGB, lib.PF_V210, 16, 16, 8, 8, 0, 0, 0, 8, 16, None) == lib.EUNSUPP
    assert b"unsupported" in l.ug_hip_last_error_string()
    assert l.ug_hip_pixfmt_supported(lib.PF_V210, lib.PF_UYVY) == 1 and l.ug_hip_pixfmt_supported(lib.PF_RGB, lib.PF_V210) == 0
    assert l.ug_hip_dxt_size(lib.DXT1, 1920, 1080) == 1036800 and l.ug_hip_dxt_size(lib.DXT5_YCOCG, 3840, -2160) == 8294400
    # linesizes == vc_get_linesize (video_codec.c:507-521; SURVEY.md 8 geometry table)
    assert l.ug_hip_linesize(lib.PF_V210, 1920) == 5120 and l.ug_hip_linesize(lib.PF_V210, 7680) == 20480
    assert l.ug_hip_linesize(lib.PF_UYVY, 3841) == 7684 and l.ug_hip_linesize(lib.PF_RGB, 1920) == 5760
    n = C.c_int(0)
    rc = l.ug_hip_device_count(C.byref(n))
    assert rc in (lib.SUCCESS, lib.ERUNTIME)  # no GPU here: error code + message, never a crash


def test_jpeg_tables_match_oracle(po):
    l = lib.load()
    import numpy as np
    for q in (1, 35, 50, 75, 100):
        for comp in (0, 1):
            t = (C.c_uint8 * 64)()
            d = (C.c_float * 64)()
            l.ug_hip_jpeg_qtable(q, comp, t)
            l.ug_hip_jpeg_divisors(t, d)
            assert np.array_equal(np.frombuffer(t, np.uint8), po.jpeg_qtable(q, comp))
            assert np.array_equal(np.frombuffer(d, np.float32).view(np.uint32), po.jpeg_divisors(po.jpeg_qtable(q, comp)).view(np.uint32))


def test_product_does_not_reference_the_oracle():
    """The product path must not import / link anything under oracle/."""
    pkg = os.path.join(ROOT, "ultragrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and '"oracle.h"' not in txt, os.path.join(dirpath, f)
