"""Test-only: the image's libjpeg-turbo (IJG API 8, no headers installed) driven through ctypes with its FLOAT DCT (JDCT_FLOAT) -- the
published implementation of the algorithm oracle/jpeg_oracle.c restates (jfdctflt.c / jcdctmgr.c; on x86-64 their SSE2 forms).  The
fields of struct jpeg_compress_struct that have to be set by hand are addressed by offset (jpeglib.h of libjpeg-turbo 2.1.x with
JPEG_LIB_VERSION 80, LP64); the library confirms the struct's size itself (jpeg_CreateCompress refuses any other) and compress() checks
the defaults it finds at the neighbouring offsets before it trusts them."""
import ctypes as C

import numpy as np

SIZE = 584                      # sizeof(struct jpeg_compress_struct), checked by jpeg_CreateCompress
OFF = dict(image_width=48, image_height=52, input_components=56, in_color_space=60, data_precision=88, num_components=92,
           do_fancy_downsampling=304, smoothing_factor=308, dct_method=312, restart_interval=316, restart_in_rows=320)
JCS_GRAYSCALE, JCS_RGB = 1, 2
JDCT_ISLOW, JDCT_FLOAT = 0, 2


def load():
    for name in ("libjpeg.so.8", "/usr/lib/x86_64-linux-gnu/libjpeg.so.8"):
        try:
            lj = C.CDLL(name)
            lj.jpeg_fdct_float, lj.jpeg_quality_scaling, lj.jpeg_CreateCompress, lj.jpeg_mem_dest, lj.jpeg_set_colorspace
        except (OSError, AttributeError):
            continue
        lj.jpeg_std_error.restype = C.c_void_p
        lj.jpeg_std_error.argtypes = [C.c_void_p]
        lj.jpeg_CreateCompress.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        lj.jpeg_mem_dest.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ulong)]
        lj.jpeg_set_defaults.argtypes = [C.c_void_p]
        lj.jpeg_set_colorspace.argtypes = [C.c_void_p, C.c_int]
        lj.jpeg_set_quality.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lj.jpeg_start_compress.argtypes = [C.c_void_p, C.c_int]
        lj.jpeg_write_scanlines.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        lj.jpeg_write_scanlines.restype = C.c_uint
        lj.jpeg_finish_compress.argtypes = [C.c_void_p]
        lj.jpeg_destroy_compress.argtypes = [C.c_void_p]
        lj.jpeg_fdct_float.restype = None
        lj.jpeg_fdct_float.argtypes = [C.c_void_p]
        lj.jpeg_quality_scaling.restype = C.c_int
        lj.jpeg_quality_scaling.argtypes = [C.c_int]
        return lj
    return None


def compress(lj, image: np.ndarray, quality: int, dct: int = JDCT_FLOAT, restart: int = 0) -> bytes:
    """image: (h, w) grey or (h, w, 3) R,G,B kept as R,G,B components (no colour transform, 1x1 sampling, table 0: what GPUJPEG is asked
    for with RGB input, gpujpeg.cpp:303-305); standard Huffman tables; `restart` MCUs per restart interval.  Returns the JPEG stream."""
    image = np.ascontiguousarray(image, np.uint8)
    h, w = image.shape[:2]
    nc = 1 if image.ndim == 2 else 3
    err = C.create_string_buffer(1024)
    cinfo = C.create_string_buffer(SIZE)
    C.cast(cinfo, C.POINTER(C.c_void_p))[0] = lj.jpeg_std_error(err)
    lj.jpeg_CreateCompress(cinfo, 80, SIZE)
    outp, outn = C.c_void_p(0), C.c_ulong(0)
    lj.jpeg_mem_dest(cinfo, C.byref(outp), C.byref(outn))
    f = C.cast(cinfo, C.POINTER(C.c_int))
    f[OFF["image_width"] // 4], f[OFF["image_height"] // 4] = w, h
    f[OFF["input_components"] // 4], f[OFF["in_color_space"] // 4] = nc, JCS_GRAYSCALE if nc == 1 else JCS_RGB
    lj.jpeg_set_defaults(cinfo)
    if nc == 3:
        lj.jpeg_set_colorspace(cinfo, JCS_RGB)
    # the layout this module assumes, confirmed on the defaults: 8-bit precision, the component count, fancy downsampling on, no smoothing,
    # JDCT_ISLOW, no restart interval
    seen = tuple(f[OFF[k] // 4] for k in ("data_precision", "num_components", "do_fancy_downsampling", "smoothing_factor", "dct_method", "restart_interval", "restart_in_rows"))
    assert seen == (8, nc, 1, 0, JDCT_ISLOW, 0, 0), seen
    lj.jpeg_set_quality(cinfo, quality, 1)
    f[OFF["dct_method"] // 4] = dct
    f[OFF["restart_interval"] // 4] = restart
    lj.jpeg_start_compress(cinfo, 1)
    rows = (C.c_void_p * h)(*[image[y].ctypes.data for y in range(h)])
    done = 0
    while done < h:
        done += lj.jpeg_write_scanlines(cinfo, C.byref(rows, done * 8), h - done)
    lj.jpeg_finish_compress(cinfo)
    data = C.string_at(outp.value, outn.value)
    lj.jpeg_destroy_compress(cinfo)
    C.CDLL(None).free(outp)
    return data


def scan_bytes(stream: bytes) -> bytes:
    """the entropy-coded bytes of a one-scan stream: behind the SOS header, in front of EOI"""
    i = stream.index(b"\xff\xda")
    n = int.from_bytes(stream[i + 2:i + 4], "big")
    assert stream[-2:] == b"\xff\xd9"
    return stream[i + 2 + n:-2]


def compress_planes(lj, y: np.ndarray, cb: np.ndarray, cr: np.ndarray, width: int, height: int, sub: int, quality: int, restart: int = 0,
                    dct: int = JDCT_FLOAT) -> bytes:
    """YCbCr 4:2:0 (sub = 420) or 4:2:2 (422) from planes that are ALREADY subsampled (raw_data_in: libjpeg's own downsampler is out of the
    way, jpeg_write_raw_data takes the component planes as they are): Y of (height, width), Cb / Cr of the subsampled size.  Rows and
    columns up to the next whole MCU are edge-replicated here, as jpeg_write_raw_data leaves that to its caller."""
    assert sub in (420, 422)
    vs = 2 if sub == 420 else 1
    mcu_w, mcu_h = 16, 8 * vs
    mw, mh = (width + mcu_w - 1) // mcu_w, (height + mcu_h - 1) // mcu_h

    def pad(p, rows, cols):
        p = np.ascontiguousarray(p, np.uint8)
        return np.ascontiguousarray(np.pad(p, ((0, rows - p.shape[0]), (0, cols - p.shape[1])), mode="edge"))
    planes = [pad(y, mh * mcu_h, mw * mcu_w), pad(cb, mh * 8, mw * 8), pad(cr, mh * 8, mw * 8)]
    err = C.create_string_buffer(1024)
    cinfo = C.create_string_buffer(SIZE)
    C.cast(cinfo, C.POINTER(C.c_void_p))[0] = lj.jpeg_std_error(err)
    lj.jpeg_CreateCompress(cinfo, 80, SIZE)
    outp, outn = C.c_void_p(0), C.c_ulong(0)
    lj.jpeg_mem_dest(cinfo, C.byref(outp), C.byref(outn))
    f = C.cast(cinfo, C.POINTER(C.c_int))
    f[OFF["image_width"] // 4], f[OFF["image_height"] // 4] = width, height
    f[OFF["input_components"] // 4], f[OFF["in_color_space"] // 4] = 3, 3   # JCS_YCbCr
    lj.jpeg_set_defaults(cinfo)   # YCbCr in -> YCbCr out, 2x2 / 1x1 / 1x1, tables 0 / 1 / 1
    seen = tuple(f[OFF[k] // 4] for k in ("data_precision", "num_components", "do_fancy_downsampling", "smoothing_factor", "dct_method", "restart_interval", "restart_in_rows"))
    assert seen == (8, 3, 1, 0, JDCT_ISLOW, 0, 0), seen
    RAW_DATA_IN = 288
    assert f[RAW_DATA_IN // 4] == 0
    # jpeg_component_info[0]: component_id, component_index, h_samp_factor, v_samp_factor, quant_tbl_no, dc_tbl_no, ac_tbl_no (ints)
    comp = C.cast(C.cast(C.byref(cinfo, 104), C.POINTER(C.c_void_p))[0], C.POINTER(C.c_int))
    assert (comp[0], comp[2], comp[3], comp[4], comp[5], comp[6]) == (1, 2, 2, 0, 0, 0), tuple(comp[i] for i in range(7))   # (the index is filled in later)
    comp[3] = vs
    lj.jpeg_set_quality(cinfo, quality, 1)
    f[OFF["dct_method"] // 4] = dct
    f[OFF["restart_interval"] // 4] = restart
    f[RAW_DATA_IN // 4] = 1
    f[OFF["do_fancy_downsampling"] // 4] = 0   # (required with raw data since IJG v7)
    lj.jpeg_start_compress(cinfo, 1)
    lj.jpeg_write_raw_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lj.jpeg_write_raw_data.restype = C.c_uint
    for r in range(mh):
        rows = [(C.c_void_p * (mcu_h if c == 0 else 8))(*[planes[c][r * (mcu_h if c == 0 else 8) + i].ctypes.data for i in range(mcu_h if c == 0 else 8)]) for c in range(3)]
        image = (C.c_void_p * 3)(*[C.addressof(x) for x in rows])
        assert lj.jpeg_write_raw_data(cinfo, image, mcu_h) == mcu_h
    lj.jpeg_finish_compress(cinfo)
    data = C.string_at(outp.value, outn.value)
    lj.jpeg_destroy_compress(cinfo)
    C.CDLL(None).free(outp)
    return data
