"""Thin torch-tensor front end over the C ABI (test / bench driver).

torch is used only for device memory and streams; every compute call goes through
libug_mi355x.so (ultragrid_amd/lib.py).  Function names follow the reference interface
they stand in for (cuda_dxt.h, pixfmt_conv.h decoder_t loop, to_planar.h).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _u8(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError("device tensor required (the product path has no CPU fallback)")
    if t.dtype != torch.uint8 or not t.is_contiguous():
        raise ValueError("contiguous uint8 tensor required")
    return t


def dxt_size(out: int, w: int, h: int) -> int:
    return L.load().ug_hip_dxt_size(out, w, h)


def dxt_encode(in_fmt: int, out_fmt: int, src: torch.Tensor, w: int, h: int, pitch: int = 0,
               dst: torch.Tensor | None = None, ties: int | None = None) -> torch.Tensor:
    """Fused unpack + colour conversion + block encode of one image (cuda_{rgb,yuv}_to_dxt{1,6},
    cuda_dxt.h:30-89, plus the decoder_t pre-pass).  h < 0 reads the source bottom-up.  ties: L.TIES_EVEN / L.TIES_AWAY
    (None = the library default, ug_hip_dxt_encode)."""
    src = _u8(src)
    if dst is None:
        dst = torch.empty(dxt_size(out_fmt, w, h), dtype=torch.uint8, device=src.device)
    if ties is None:
        rc = L.load().ug_hip_dxt_encode(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, pitch, _stream())
    else:
        rc = L.load().ug_hip_dxt_encode_batch_ex(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, pitch, 1, 0, 0, ties, _stream())
    L.check(rc, "ug_hip_dxt_encode")
    return dst


def dxt_encode_batch(in_fmt: int, out_fmt: int, src: torch.Tensor, w: int, h: int, frames: int,
                     src_frame_stride: int, dst: torch.Tensor | None = None, pitch: int = 0, ties: int | None = None) -> torch.Tensor:
    src = _u8(src)
    per = dxt_size(out_fmt, w, h)
    if dst is None:
        dst = torch.empty(per * frames, dtype=torch.uint8, device=src.device)
    if ties is None:
        rc = L.load().ug_hip_dxt_encode_batch(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, pitch, frames,
                                              src_frame_stride, per, _stream())
    else:
        rc = L.load().ug_hip_dxt_encode_batch_ex(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, pitch, frames,
                                                 src_frame_stride, per, ties, _stream())
    L.check(rc, "ug_hip_dxt_encode_batch")
    return dst


def time_dxt_encode(in_fmt: int, out_fmt: int, src: torch.Tensor, dst: torch.Tensor, w: int, h: int, frames: int,
                    src_frame_stride: int, dst_frame_stride: int, iters: int) -> float:
    """ms per launch, hipEvents on the current stream (ug_hip_time_dxt_encode)."""
    import ctypes as C
    ms = C.c_float(0)
    rc = L.load().ug_hip_time_dxt_encode(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, 0, frames,
                                         src_frame_stride, dst_frame_stride, iters, _stream(), C.byref(ms))
    L.check(rc, "ug_hip_time_dxt_encode")
    return ms.value


def dxt_decode(in_fmt: int, out_fmt: int, blocks: torch.Tensor, w: int, h: int, shifts=(0, 8, 16), ties: int | None = None) -> torch.Tensor:
    """DXT1 / DXT5-YCoCg -> RGB / BGR / RGBA / UYVY (dxt_decoder_decompress behind video_decompress/dxt_glsl.c:142-189)."""
    blocks = _u8(blocks)
    dst = torch.empty(linesize(out_fmt, w) * h, dtype=torch.uint8, device=blocks.device)
    if ties is None:
        rc = L.load().ug_hip_dxt_decode(in_fmt, out_fmt, blocks.data_ptr(), dst.data_ptr(), w, h, 0, *shifts, _stream())
    else:
        rc = L.load().ug_hip_dxt_decode_ex(in_fmt, out_fmt, blocks.data_ptr(), dst.data_ptr(), w, h, 0, *shifts, ties, _stream())
    L.check(rc, "ug_hip_dxt_decode")
    return dst


def yuv422_to_yuv444(src: torch.Tensor, pix_count: int) -> torch.Tensor:
    src = _u8(src)
    dst = torch.empty(pix_count * 3, dtype=torch.uint8, device=src.device)
    L.check(L.load().ug_hip_yuv422_to_yuv444(src.data_ptr(), dst.data_ptr(), pix_count, _stream()),
            "ug_hip_yuv422_to_yuv444")
    return dst


def linesize(fmt: int, w: int) -> int:
    n = L.load().ug_hip_linesize(fmt, w)
    if n <= 0:
        raise L.UgHipError(n, f"ug_hip_linesize({fmt}, {w})")
    return n


def pixfmt_convert(in_fmt: int, out_fmt: int, src: torch.Tensor, w: int, h: int, shifts=(0, 8, 16)) -> torch.Tensor:
    """Whole-frame decoder_t conversion on the device (pixfmt_conv.c:3041-3125)."""
    src = _u8(src)
    dst = torch.zeros(linesize(out_fmt, w) * h, dtype=torch.uint8, device=src.device)
    rc = L.load().ug_hip_pixfmt_convert(in_fmt, out_fmt, src.data_ptr(), dst.data_ptr(), w, h, 0, 0, *shifts, _stream())
    L.check(rc, "ug_hip_pixfmt_convert")
    return dst


def uyvy_to_i420(src: torch.Tensor, w: int, h: int):
    src = _u8(src)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    y = torch.zeros((h, w), dtype=torch.uint8, device=src.device)
    u = torch.zeros((ch, cw), dtype=torch.uint8, device=src.device)
    v = torch.zeros((ch, cw), dtype=torch.uint8, device=src.device)
    rc = L.load().ug_hip_uyvy_to_i420(src.data_ptr(), 0, y.data_ptr(), w, u.data_ptr(), cw, v.data_ptr(), cw, w, h,
                                      _stream())
    L.check(rc, "ug_hip_uyvy_to_i420")
    return y, u, v


def v210_to_p010le(src: torch.Tensor, w: int, h: int, y_pad: int = 0, uv_pad: int = 0, fill: int = 0):
    """v210_to_p010le (to_planar.c:64-155), any geometry; planes of h x (w + y_pad) and ceil(h / 2) x (w + uv_pad) int16 samples
    (the pads are line padding the reference's whole-group writes reach), pre-filled with `fill`."""
    src = _u8(src)
    y = torch.full((h, w + y_pad), fill, dtype=torch.int16, device=src.device)
    uv = torch.full(((h + 1) // 2, w + uv_pad), fill, dtype=torch.int16, device=src.device)
    rc = L.load().ug_hip_v210_to_p010le(src.data_ptr(), 0, y.data_ptr(), 2 * (w + y_pad), uv.data_ptr(), 2 * (w + uv_pad), w, h, _stream())
    L.check(rc, "ug_hip_v210_to_p010le")
    return y, uv


def planar_to_uyvy(y: torch.Tensor, cb: torch.Tensor, cr: torch.Tensor, w: int, h: int, chroma: int = 420) -> torch.Tensor:
    """yuv420p_to_uyvy / yuv422p_to_uyvy (from_planar.c:583-683, 391-423): 2-D uint8 plane tensors -> packed UYVY."""
    out = torch.zeros(linesize(L.PF_UYVY, w) * h, dtype=torch.uint8, device=y.device)
    fn = L.load().ug_hip_yuv420p_to_uyvy if chroma == 420 else L.load().ug_hip_yuv422p_to_uyvy
    rc = fn(y.data_ptr(), y.stride(0), cb.data_ptr(), cb.stride(0), cr.data_ptr(), cr.stride(0), out.data_ptr(), 0, w, h, _stream())
    L.check(rc, "ug_hip_yuv42xp_to_uyvy")
    return out


def yuv422p10le_to_v210(y: torch.Tensor, cb: torch.Tensor, cr: torch.Tensor, w: int, h: int) -> torch.Tensor:
    """from_planar.c:296-333: int16/uint16 plane tensors (10-bit samples) -> v210."""
    out = torch.zeros(linesize(L.PF_V210, w) * h, dtype=torch.uint8, device=y.device)
    rc = L.load().ug_hip_yuv422p10le_to_v210(y.data_ptr(), 2 * y.stride(0), cb.data_ptr(), 2 * cb.stride(0), cr.data_ptr(), 2 * cr.stride(0),
                                             out.data_ptr(), 0, w, h, _stream())
    L.check(rc, "ug_hip_yuv422p10le_to_v210")
    return out


def uyvy_to_i422(src: torch.Tensor, w: int, h: int):
    src = _u8(src)
    cw = (w + 1) // 2
    y = torch.zeros((h, w), dtype=torch.uint8, device=src.device)
    u = torch.zeros((h, cw), dtype=torch.uint8, device=src.device)
    v = torch.zeros((h, cw), dtype=torch.uint8, device=src.device)
    rc = L.load().ug_hip_uyvy_to_i422(src.data_ptr(), 0, y.data_ptr(), w, u.data_ptr(), cw, v.data_ptr(), cw, w, h, _stream())
    L.check(rc, "ug_hip_uyvy_to_i422")
    return y, u, v


def uyvy_to_nv12(src: torch.Tensor, w: int, h: int, src_pitch: int = 0):
    src = _u8(src)
    cw = (w + 1) // 2
    y = torch.zeros((h, w), dtype=torch.uint8, device=src.device)
    c = torch.zeros(((h + 1) // 2, 2 * cw), dtype=torch.uint8, device=src.device)
    rc = L.load().ug_hip_uyvy_to_nv12(src.data_ptr(), src_pitch, y.data_ptr(), w, c.data_ptr(), 2 * cw, w, h, _stream())
    L.check(rc, "ug_hip_uyvy_to_nv12")
    return y, c


class FromPlanarData(C.Structure):
    """struct ug_from_planar_data (include/ug_mi355x.h) == struct from_planar_data (from_planar.h:58-70)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p), ("out_pitch", C.c_uint),
                ("in_data", C.c_void_p * 4), ("in_linesize", C.c_uint * 4), ("in_depth", C.c_int), ("log2_chroma_h", C.c_int),
                ("rgb_shift", C.c_int * 3)]


class ToPlanarData(C.Structure):
    """struct ug_to_planar_data == struct to_planar_data (to_planar.h:53-59)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("out_data", C.c_void_p * 4), ("out_linesize", C.c_uint * 4),
                ("in_data", C.c_void_p)]


def from_planar(func: str, planes, w: int, h: int, out: torch.Tensor, out_pitch: int, in_depth: int = 0, rgb_shift=(0, 8, 16)) -> torch.Tensor:
    """`func` = a decode_planar_func_t name of from_planar.h; planes = 2-D device tensors (uint8, or int16/uint16 holding the samples);
    `out` = preallocated uint8 device tensor.  The line sizes are the tensors' row strides."""
    d = FromPlanarData()
    d.width, d.height = w, h
    d.out_data, d.out_pitch = out.data_ptr(), out_pitch
    for i, p in enumerate(planes):
        d.in_data[i] = p.data_ptr()
        d.in_linesize[i] = p.stride(0) * p.element_size()
    d.in_depth = in_depth
    d.rgb_shift[0], d.rgb_shift[1], d.rgb_shift[2] = rgb_shift
    rc = L.load().ug_hip_from_planar(func.encode(), C.byref(d), _stream())
    L.check(rc, f"ug_hip_from_planar({func})")
    return out


def to_planar(func: str, src: torch.Tensor, w: int, h: int, planes) -> list:
    """`func` = a decode_buffer_func_t name of to_planar.h; planes = preallocated 2-D device tensors"""
    d = ToPlanarData()
    d.width, d.height = w, h
    for i, p in enumerate(planes):
        d.out_data[i] = p.data_ptr()
        d.out_linesize[i] = p.stride(0) * p.element_size()
    d.in_data = src.data_ptr()
    rc = L.load().ug_hip_to_planar(func.encode(), C.byref(d), _stream())
    L.check(rc, f"ug_hip_to_planar({func})")
    return list(planes)


class AvFrame(C.Structure):
    """struct ug_av_frame (include/ug_mi355x.h): the AVFrame fields the lavc converters read"""
    _fields_ = [("data", C.c_void_p * 4), ("linesize", C.c_int * 4), ("width", C.c_int), ("height", C.c_int),
                ("colorspace", C.c_int), ("color_range", C.c_int)]


def _av_frame(planes, w: int, h: int, colorspace: int = 2, color_range: int = 1) -> AvFrame:
    f = AvFrame()
    for i, p in enumerate(planes):
        f.data[i] = p.data_ptr()
        f.linesize[i] = p.stride(0) * p.element_size()
    f.width, f.height, f.colorspace, f.color_range = w, h, colorspace, color_range
    return f


def uv_to_av(uv_codec: str, av_pixfmt: str, src: torch.Tensor, w: int, h: int, planes) -> list:
    """to_lavc_vid_conv(): `planes` = preallocated 2-D device tensors (rows x bytes-or-samples) of the output frame"""
    f = _av_frame(planes, w, h)
    rc = L.load().ug_hip_uv_to_av(uv_codec.encode(), av_pixfmt.encode(), src.data_ptr(), C.byref(f), _stream())
    L.check(rc, f"ug_hip_uv_to_av({uv_codec}, {av_pixfmt})")
    return list(planes)


def av_to_uv(av_pixfmt: str, uv_codec: str, planes, w: int, h: int, dst: torch.Tensor, pitch: int, rgb_shift=(0, 8, 16),
             colorspace: int = 2, color_range: int = 1) -> torch.Tensor:
    """av_to_uv_convert(): decoder frame (2-D device tensors) -> packed UltraGrid buffer `dst`"""
    f = _av_frame(planes, w, h, colorspace, color_range)
    sh = (C.c_int * 3)(*rgb_shift)
    rc = L.load().ug_hip_av_to_uv(av_pixfmt.encode(), uv_codec.encode(), dst.data_ptr(), pitch, C.byref(f), sh, _stream())
    L.check(rc, f"ug_hip_av_to_uv({av_pixfmt}, {uv_codec})")
    return dst


def jpeg_divisors_device(quality: int, device) -> torch.Tensor:
    """128 fp32 divisors (luma, chroma) in device memory."""
    import ctypes as C
    import numpy as np
    out = np.zeros(128, np.float32)
    for comp in (0, 1):
        q = (C.c_uint8 * 64)()
        d = (C.c_float * 64)()
        L.load().ug_hip_jpeg_qtable(quality, comp, q)
        L.load().ug_hip_jpeg_divisors(q, d)
        out[64 * comp: 64 * comp + 64] = np.frombuffer(d, np.float32)
    return torch.from_numpy(out).to(device)


def jpeg_fdct_quant_plane(plane: torch.Tensor, div: torch.Tensor, blocks_w: int = 0, blocks_h: int = 0,
                          want_coef: bool = False):
    h, w = plane.shape
    bw = blocks_w or (w + 7) // 8
    bh = blocks_h or (h + 7) // 8
    out = torch.zeros((bw * bh, 64), dtype=torch.int16, device=plane.device)
    coef = torch.zeros((bw * bh, 64), dtype=torch.float32, device=plane.device) if want_coef else None
    rc = L.load().ug_hip_jpeg_fdct_quant_plane(plane.data_ptr(), plane.stride(0), w, h, bw, bh, div.data_ptr(),
                                               out.data_ptr(), coef.data_ptr() if want_coef else None, _stream())
    L.check(rc, "ug_hip_jpeg_fdct_quant_plane")
    return (out, coef) if want_coef else out


def uyvy_to_jpeg_coeffs(src: torch.Tensor, w: int, h: int, div: torch.Tensor, subsampling: int = 420):
    """Fused UYVY -> planar 4:2:0 / 4:2:2 -> FDCT + quantise (ug_hip_uyvy_to_jpeg42x_coeffs)."""
    src = _u8(src)
    mw = (w + 15) // 16
    mh, ybl = ((h + 15) // 16, 4) if subsampling == 420 else ((h + 7) // 8, 2)
    oy = torch.zeros((ybl * mw * mh, 64), dtype=torch.int16, device=src.device)
    ocb = torch.zeros((mw * mh, 64), dtype=torch.int16, device=src.device)
    ocr = torch.zeros((mw * mh, 64), dtype=torch.int16, device=src.device)
    fn = {420: L.load().ug_hip_uyvy_to_jpeg420_coeffs, 422: L.load().ug_hip_uyvy_to_jpeg422_coeffs}[subsampling]
    rc = fn(src.data_ptr(), 0, w, h, div.data_ptr(), oy.data_ptr(), ocb.data_ptr(), ocr.data_ptr(), _stream())
    L.check(rc, f"ug_hip_uyvy_to_jpeg{subsampling}_coeffs")
    return oy, ocb, ocr


def uyvy_to_jpeg420_coeffs(src: torch.Tensor, w: int, h: int, div: torch.Tensor):
    return uyvy_to_jpeg_coeffs(src, w, h, div, 420)


def jpeg_colour_matrix(cs_in: int, cs_out: int):
    """the 3 x 4 affine map of the JPEG encoder's colour stage on 8-bit code values (ug_hip_jpeg_colour_matrix); no GPU involved"""
    import ctypes as C
    import numpy as np
    m = (C.c_float * 12)()
    L.check(L.load().ug_hip_jpeg_colour_matrix(cs_in, cs_out, m), "ug_hip_jpeg_colour_matrix")
    return np.frombuffer(m, np.float32).reshape(3, 4).copy()


def jpeg_colour_convert(fmt: int, cs_in: int, cs_out: int, src: torch.Tensor, w: int, h: int) -> torch.Tensor:
    """RGB (3 B/px) or UYVY frame from colour space cs_in to cs_out (L.JPEG_CS_RGB .. L.JPEG_CS_YCBCR_BT709), packed lines"""
    src = _u8(src)
    dst = torch.empty_like(src)
    L.check(L.load().ug_hip_jpeg_colour_convert(fmt, cs_in, cs_out, src.data_ptr(), 0, dst.data_ptr(), 0, w, h, _stream()), "ug_hip_jpeg_colour_convert")
    return dst


class JpegEncoder:
    """ug_hip_jpeg_encoder_* (gpujpeg_encoder_create / _encode / _destroy shape, gpujpeg.cpp:353,624,639)."""

    def __init__(self, w: int, h: int, quality: int = 75, restart_interval: int = 4, subsampling: int = 420, internal_cs: int = 0, flags: int = 0):
        """internal_cs: L.JPEG_CS_* (color_space_internal of gpujpeg.cpp:303-305), flags: L.JPEG_NONINTERLEAVED, L.JPEG_INPUT_UYVY (ug_hip_jpeg_encoder_create_ex)"""
        import ctypes as C
        self._h = C.c_void_p()
        if internal_cs or flags:
            L.check(L.load().ug_hip_jpeg_encoder_create_ex(w, h, quality, restart_interval, subsampling, internal_cs, flags, C.byref(self._h)),
                    "ug_hip_jpeg_encoder_create_ex")
        else:
            L.check(L.load().ug_hip_jpeg_encoder_create_sub(w, h, quality, restart_interval, subsampling, C.byref(self._h)),
                    "ug_hip_jpeg_encoder_create_sub")
        self.max_size = L.load().ug_hip_jpeg_encoder_max_size(self._h)
        self._out = None

    def encode(self, src: torch.Tensor, in_fmt: int = L.PF_UYVY) -> bytes:
        """src: UYVY (4:2:0 / 4:2:2 encoder), RGB (4:4:4 encoder) or I420 planes back to back (4:2:0 encoder)."""
        import ctypes as C
        src = _u8(src)
        if self._out is None:
            self._out = torch.empty(self.max_size, dtype=torch.uint8, device=src.device)
        n = C.c_size_t(0)
        rc = L.load().ug_hip_jpeg_encoder_encode(self._h, in_fmt, src.data_ptr(), 0, self._out.data_ptr(), self.max_size, C.byref(n), _stream())
        L.check(rc, "ug_hip_jpeg_encoder_encode")
        return bytes(self._out[: n.value].cpu().numpy())

    def encode_batch(self, srcs: torch.Tensor, in_fmt: int = L.PF_UYVY) -> list:
        """srcs: (n, frame bytes) -- n frames, one launch sequence, one synchronisation; returns the n streams."""
        import ctypes as C
        srcs = srcs if srcs.dtype == torch.uint8 else srcs.view(torch.uint8)
        assert srcs.dim() == 2 and srcs.is_contiguous()
        n = srcs.shape[0]
        stride = (self.max_size + 15) // 16 * 16
        out = torch.empty((n, stride), dtype=torch.uint8, device=srcs.device)
        lens = (C.c_size_t * n)()
        rc = L.load().ug_hip_jpeg_encoder_encode_batch(self._h, in_fmt, n, srcs.data_ptr(), 0, srcs.shape[1], out.data_ptr(), stride, self.max_size, lens, _stream())
        L.check(rc, "ug_hip_jpeg_encoder_encode_batch")
        # ABI 3: with frames > 1 a stream that did not fit its slice is reported per frame (out_len[f] > capacity), not through the return code
        for f in range(n):
            if lens[f] > self.max_size:
                raise L.UgHipError(L.EINVAL, f"ug_hip_jpeg_encoder_encode_batch: the stream of frame {f} needs {lens[f]} bytes, its slice holds {self.max_size}")
        return [bytes(out[f, : lens[f]].cpu().numpy()) for f in range(n)]

    def close(self):
        if self._h:
            L.load().ug_hip_jpeg_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jpeg_read_info(data: bytes) -> dict:
    import ctypes as C
    w, h, sub, rgb, ri = (C.c_int() for _ in range(5))
    L.check(L.load().ug_hip_jpeg_read_info(data, len(data), C.byref(w), C.byref(h), C.byref(sub), C.byref(rgb), C.byref(ri)), "ug_hip_jpeg_read_info")
    return dict(width=w.value, height=h.value, subsampling=sub.value, is_rgb=bool(rgb.value), restart=ri.value)


class JpegDecoder:
    """ug_hip_jpeg_decoder_* (gpujpeg_decoder_create / _decode / _destroy shape, video_decompress/gpujpeg.c:74-140,292-301)."""

    def __init__(self):
        import ctypes as C
        self._h = C.c_void_p()
        L.check(L.load().ug_hip_jpeg_decoder_create(C.byref(self._h)), "ug_hip_jpeg_decoder_create")

    def decode(self, data: bytes, out_fmt: int, shifts=(0, 8, 16), device="cuda") -> torch.Tensor:
        info = jpeg_read_info(data)
        w, h = info["width"], info["height"]
        n = w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2) if out_fmt == L.PF_I420 else linesize(out_fmt, w) * h
        dst = torch.empty(n, dtype=torch.uint8, device=device)
        # _sized: dst was made for the size the first look at the headers gave; the decoder refuses a stream whose own parse says otherwise
        L.check(L.load().ug_hip_jpeg_decoder_decode_sized(self._h, data, len(data), w, h, out_fmt, dst.data_ptr(), 0, *shifts, _stream()),
                "ug_hip_jpeg_decoder_decode_sized")
        return dst

    def planes(self, data: bytes):
        """Decode to the component planes only; returns them cropped to the component sizes (device tensors)."""
        import ctypes as C
        L.check(L.load().ug_hip_jpeg_decoder_decode(self._h, data, len(data), L.PF_NONE, None, 0, 0, 8, 16, _stream()), "ug_hip_jpeg_decoder_decode")
        torch.cuda.synchronize()
        out = []
        for c in range(3):
            p, pitch, w, h = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
            if L.load().ug_hip_jpeg_decoder_plane(self._h, c, C.byref(p), C.byref(pitch), C.byref(w), C.byref(h)) != 0:
                break
            t = torch.empty((h.value, pitch.value), dtype=torch.uint8, device="cuda")
            L.check(L.load().ug_hip_memcpy(t.data_ptr(), p.value, pitch.value * h.value, 2), "ug_hip_memcpy")
            out.append(t[:, : w.value].contiguous())
        return out

    def close(self):
        if self._h:
            L.load().ug_hip_jpeg_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
