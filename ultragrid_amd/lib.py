"""ctypes binding of libug_mi355x.so (include/ug_mi355x.h).

This is the test / bench driver's view of the C ABI; UltraGrid itself binds the same
symbols from C++ (ultragrid_amd/module/).  There is NO fallback: if the HIP library is
missing or a symbol is absent, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UG_MI355X_LIB: developer override used for A/B runs of kernel variants (always a HIP build of this library)
LIB_PATH = os.environ.get("UG_MI355X_LIB") or os.path.join(_HERE, "libug_mi355x.so")

# ug_pixfmt_t
PF_NONE, PF_RGBA, PF_UYVY, PF_YUYV, PF_RGB, PF_BGR, PF_V210, PF_RG48, PF_YUV444, PF_UYVY_RAW, PF_I420 = range(11)
PF_R10K, PF_R12L, PF_Y216, PF_Y416, PF_VUYA, PF_DVS10 = range(11, 17)
PF_NAMES = {"RGBA": PF_RGBA, "UYVY": PF_UYVY, "YUYV": PF_YUYV, "RGB": PF_RGB, "BGR": PF_BGR, "v210": PF_V210,
            "RG48": PF_RG48, "YUV444": PF_YUV444, "UYVY_RAW": PF_UYVY_RAW, "R10k": PF_R10K, "R12L": PF_R12L, "Y216": PF_Y216,
            "Y416": PF_Y416, "VUYA": PF_VUYA, "DVS10": PF_DVS10}
# ug_dxt_t
DXT1, DXT1_YUV, DXT5_YCOCG = 1, 2, 6
# UG_DXT_TIES_*
TIES_EVEN, TIES_AWAY = 0, 1
JPEG_CS_ASIS, JPEG_CS_RGB, JPEG_CS_YCBCR_BT601, JPEG_CS_YCBCR_BT601_256LVLS, JPEG_CS_YCBCR_BT709 = 0, 1, 2, 3, 4
JPEG_NONINTERLEAVED = 1
JPEG_INPUT_UYVY = 2
COPY_NO_WAIT, COPY_NO_JOIN = 1, 2
ABI_VERSION = 5

SUCCESS, EINVAL, EUNSUPP, ERUNTIME = 0, -1, -2, -3

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t

# every symbol include/ug_mi355x.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ug_hip_abi_version": (_i, []),
    "ug_hip_device_count": (_i, [C.POINTER(_i)]),
    "ug_hip_pointer_is_device": (_i, [_vp]),
    "ug_hip_pointer_device": (_i, [_vp]),
    "ug_hip_set_device": (_i, [_i]),
    "ug_hip_malloc": (_i, [C.POINTER(_vp), _sz]),
    "ug_hip_free": (_i, [_vp]),
    "ug_hip_malloc_host": (_i, [C.POINTER(_vp), _sz]),
    "ug_hip_free_host": (_i, [_vp]),
    "ug_hip_memcpy": (_i, [_vp, _vp, _sz, _i]),
    "ug_hip_memcpy_async": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ug_hip_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "ug_hip_event_create": (_i, [C.POINTER(C.c_void_p)]),
    "ug_hip_event_destroy": (_i, [_vp]),
    "ug_hip_event_record": (_i, [_vp, _vp]),
    "ug_hip_stream_wait_event": (_i, [_vp, _vp]),
    "ug_hip_memcpy_2d_async": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _i, _vp]),
    "ug_hip_upload_ordered": (_i, [_i, _vp, _vp, _sz, _i, _vp]),
    "ug_hip_download_ordered": (_i, [_i, _vp, _vp, _sz, _vp]),
    "ug_hip_upload_ordered_ex": (_i, [_i, _vp, _vp, _sz, _i, _vp, _i]),
    "ug_hip_download_ordered_ex": (_i, [_i, _vp, _vp, _sz, _vp, _i]),
    "ug_hip_download_2d_ordered_ex": (_i, [_i, _vp, _sz, _vp, _sz, _sz, _sz, _vp, _i]),
    "ug_hip_stream_create": (_i, [C.POINTER(_vp)]),
    "ug_hip_stream_destroy": (_i, [_vp]),
    "ug_hip_stream_sync": (_i, [_vp]),
    "ug_hip_last_error_string": (C.c_char_p, []),
    "ug_hip_device_numa_node": (_i, [_i, C.POINTER(_i)]),
    "ug_hip_bind_thread_to_device": (_i, [_i, C.POINTER(_i)]),
    "ug_hip_numa_node_of_pci": (_i, [C.c_char_p, C.c_char_p, C.POINTER(_i)]),
    "ug_hip_bind_thread_to_numa_node": (_i, [_i, C.c_char_p, C.POINTER(_i)]),
    "ug_hip_time_dxt_encode": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _i, _vp, C.POINTER(C.c_float)]),
    "ug_hip_dxt_encode": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "ug_hip_dxt_encode_batch": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _vp]),
    "ug_hip_dxt_encode_batch_ex": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _i, _vp]),
    "ug_hip_dxt_size": (_sz, [_i, _i, _i]),
    "ug_hip_rgb_to_dxt1": (_i, [_vp, _vp, _i, _i, _vp]),
    "ug_hip_yuv_to_dxt1": (_i, [_vp, _vp, _i, _i, _vp]),
    "ug_hip_rgb_to_dxt6": (_i, [_vp, _vp, _i, _i, _vp]),
    "ug_hip_yuv_to_dxt6": (_i, [_vp, _vp, _i, _i, _vp]),
    "ug_hip_yuv422_to_yuv444": (_i, [_vp, _vp, _i, _vp]),
    "ug_hip_dxt_decode": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ug_hip_dxt_decode_ex": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ug_hip_pixfmt_supported": (_i, [_i, _i]),
    "ug_hip_pixfmt_convert": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ug_hip_pixfmt_convert_batch": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _sz, _sz, _vp]),
    "ug_hip_linesize": (_i, [_i, _i]),
    "ug_hip_deinterlace_blend": (_i, [_vp, _sz, _i, _vp]),
    "ug_hip_deinterlace_blend_batch": (_i, [_vp, _sz, _i, _i, _sz, _vp]),
    "ug_hip_uyvy_to_i420": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_v210_to_p010le": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_yuv420p_to_uyvy": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_yuv422p_to_uyvy": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_yuv422p10le_to_v210": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_uyvy_to_i422": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_selftest_dxt_decode": (_i, [C.POINTER(C.c_uint), _vp]),
    "ug_hip_dxt_decode_debug": (_i, [_i, _vp]),
    "ug_hip_selftest_dxt_encode": (_i, [C.POINTER(C.c_uint), _vp]),
    "ug_hip_dxt_encode_stats": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "ug_hip_uyvy_to_nv12": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_pixfmt_best": (_i, [_i, C.POINTER(_i), C.POINTER(_i)]),
    "ug_hip_pixfmt_line_func": (_i, [C.c_char_p, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ug_hip_from_planar": (_i, [C.c_char_p, _vp, _vp]),
    "ug_hip_to_planar": (_i, [C.c_char_p, _vp, _vp]),
    "ug_hip_from_planar_supported": (_i, [C.c_char_p]),
    "ug_hip_to_planar_supported": (_i, [C.c_char_p]),
    "ug_hip_uv_to_av": (_i, [C.c_char_p, C.c_char_p, _vp, _vp, _vp]),
    "ug_hip_av_to_uv": (_i, [C.c_char_p, C.c_char_p, _vp, _i, _vp, C.POINTER(_i), _vp]),
    "ug_hip_uv_to_av_supported": (_i, [C.c_char_p, C.c_char_p]),
    "ug_hip_av_to_uv_supported": (_i, [C.c_char_p, C.c_char_p]),
    "ug_hip_color_coeffs": (_i, [_i, _i, C.POINTER(_i)]),
    "ug_hip_compute_color_coeffs": (_i, [C.c_double, C.c_double, _i, C.POINTER(_i)]),
    "ug_hip_jpeg_qtable": (None, [_i, _i, _vp]),
    "ug_hip_jpeg_divisors": (None, [_vp, _vp]),
    "ug_hip_jpeg_fdct_quant_plane": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ug_hip_uyvy_to_jpeg420_coeffs": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ug_hip_uyvy_to_jpeg422_coeffs": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ug_hip_uyvy_to_jpeg42x_coeffs_batch": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _sz, _sz, _sz, _vp]),
    "ug_hip_jpeg_encoder_create": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "ug_hip_jpeg_encoder_create_sub": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ug_hip_jpeg_encoder_create_ex": (_i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ug_hip_jpeg_colour_matrix": (_i, [_i, _i, C.POINTER(C.c_float)]),
    "ug_hip_jpeg_colour_convert": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ug_hip_jpeg_encoder_destroy": (None, [_vp]),
    "ug_hip_jpeg_encoder_max_size": (_sz, [_vp]),
    "ug_hip_jpeg_decoder_create": (_i, [C.POINTER(_vp)]),
    "ug_hip_jpeg_decoder_destroy": (None, [_vp]),
    "ug_hip_jpeg_read_info": (_i, [_vp, _sz, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "ug_hip_jpeg_decoder_decode": (_i, [_vp, _vp, _sz, _i, _vp, _i, _i, _i, _i, _vp]),
    "ug_hip_jpeg_decoder_decode_sized": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "ug_hip_jpeg_decoder_plane": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "ug_hip_jpeg_encoder_encode": (_i, [_vp, _i, _vp, _i, _vp, _sz, C.POINTER(_sz), _vp]),
    "ug_hip_jpeg_encoder_encode_batch": (_i, [_vp, _i, _i, _vp, _i, _sz, _vp, _sz, _sz, C.POINTER(_sz), _vp]),
}


class UgHipError(RuntimeError):
    def __init__(self, rc: int, what: str):
        self.rc = rc
        super().__init__(f"{what}: rc={rc}: {last_error()}")


_lib = None


def load() -> C.CDLL:
    """Load the HIP library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C ultragrid_amd/csrc` (there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if lib.ug_hip_abi_version() != ABI_VERSION:
            raise ImportError("libug_mi355x.so ABI version mismatch")
        _lib = lib
    return _lib


def last_error() -> str:
    return load().ug_hip_last_error_string().decode()


def check(rc: int, what: str) -> None:
    if rc != SUCCESS:
        raise UgHipError(rc, what)
