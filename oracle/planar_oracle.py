"""TEST INFRASTRUCTURE ONLY -- numpy restatement of every whole-buffer converter of src/from_planar.h (decode_planar_func_t,
from_planar.h:86-113) and src/to_planar.h (decode_buffer_func_t, to_planar.h:65-74), plus by-name calls into the compiled
reference (oracle/_ref/libugref.so, built from /root/reference/src/{from,to}_planar.c where they lie).

PARITY PINNED: tests/test_planar_api.py checks every restatement against the compiled reference on the same inputs, including
samples that carry bits above the nominal depth (the reference never masks them) and ragged widths.

Only tests/ may import this module; the product (ultragrid_amd/) never does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import pyoracle as O

FROM_NAMES = [
    "gbrap_to_rgb", "gbrap_to_rgba", "gbrp10le_to_rgb", "gbrp10le_to_rgba", "gbrp10le_to_rg48", "gbrp10le_to_r10k",
    "gbrp12le_to_rgb", "gbrp12le_to_rgba", "gbrp12le_to_rg48", "gbrp12le_to_r10k", "gbrp12le_to_r12l",
    "gbrp16le_to_rgb", "gbrp16le_to_rgba", "gbrp16le_to_rg48", "gbrp16le_to_r10k", "gbrp16le_to_r12l",
    "rgbpXX_to_rgb", "rgbpXXle_to_rg48", "rgbpXXle_to_r10k", "rgbpXXle_to_r12l",
    "yuv444p_to_vuya", "yuv420p_to_uyvy", "yuv420_to_i420", "yuv422p_to_uyvy", "yuv422p_to_yuyv", "yuv422pXX_to_uyvy",
    "yuv422p10le_to_uyvy", "yuv422p10le_to_v210",
]
TO_NAMES = ["v210_to_p010le", "y216_to_p010le", "uyvy_to_nv12", "rgba_to_bgra", "vuya_to_i444", "uyvy_to_i420",
            "r12l_to_gbrp12le", "r12l_to_gbrp16le", "r12l_to_rgbp12le"]


def from_info(name: str, in_depth: int = 0):
    """(family, out, depth, plane order r/g/b/a) of a decode_planar_func_t, as from_planar.c wires it"""
    if name.startswith(("gbrap", "gbrp", "rgbp")):
        out = name.split("_to_")[1]
        if name.startswith("gbrap"):
            return "rgbp", out, 8, (2, 0, 1, 3 if out == "rgba" else -1)
        if name.startswith("gbrp"):
            return "rgbp", out, int(name[4:6]), (2, 0, 1, -1)
        return "rgbp", out, in_depth, (0, 1, 2, -1)
    return "yuv", name, {"yuv422pXX_to_uyvy": in_depth, "yuv422p10le_to_uyvy": 10, "yuv422p10le_to_v210": 10}.get(name, 8), None


def out_linesize(name: str, w: int) -> int:
    """bytes of one output line (tightly packed; R12L / v210 as vc_get_linesize)"""
    out = name.split("_to_")[1]
    return {"rgb": 3 * w, "rgba": 4 * w, "rg48": 6 * w, "r10k": 4 * w, "r12l": (w + 7) // 8 * 36, "vuya": 4 * w,
            "uyvy": 4 * ((w + 1) // 2), "yuyv": 4 * ((w + 1) // 2), "v210": O.linesize(w, "v210"), "i420": w}[out]


def _r12l_pack(vals: np.ndarray) -> np.ndarray:
    """vals: (h, groups, 24) uint32 (unmasked) -> (h, groups*36) bytes, the statement sequence of from_planar.c:82-126"""
    e, o = vals[..., 0::2], vals[..., 1::2]
    b = np.empty(vals.shape[:2] + (12, 3), np.uint32)
    b[..., 0] = e
    b[..., 1] = (o & 0xF) << 4 | e >> 8
    b[..., 2] = o >> 4
    return (b & 0xFF).astype(np.uint8).reshape(vals.shape[0], -1)


def from_planar(name: str, planes, w: int, h: int, in_depth: int = 0, rgb_shift=(0, 8, 16), out_pitch: int = 0) -> np.ndarray:
    """numpy restatement; returns (h, out_pitch) bytes (yuv420_to_i420: flat I420).  planes: 2-D arrays (8-bit: uint8, else uint16)."""
    fam, out, depth, idx = from_info(name, in_depth)
    pitch = out_pitch or out_linesize(name, w)
    dst = np.zeros((h, pitch), np.uint8)
    if fam == "rgbp":
        if depth == 8:  # gbrap_to_rgb_rgba, from_planar.c:335-354 (every plane strided by in_linesize[0]: the caller passes equal strides)
            comps = [np.asarray(planes[i])[:h, :w] for i in idx if i >= 0]
            px = np.stack(comps, -1).astype(np.uint8)
            dst[:, : px.shape[2] * w] = px.reshape(h, -1)
            return dst
        r, g, b = (np.asarray(planes[i])[:h, :w].astype(np.uint32) for i in idx[:3])
        d = depth
        if out == "rgb":  # from_planar.c:477-497
            px = np.stack([r >> (d - 8), g >> (d - 8), b >> (d - 8)], -1) & 0xFF
            dst[:, : 3 * w] = px.astype(np.uint8).reshape(h, -1)
        elif out == "rgba":  # :499-529 (planes fixed G,B,R)
            rs, gs, bs = rgb_shift
            am = 0xFFFFFFFF ^ (0xFF << rs) ^ (0xFF << gs) ^ (0xFF << bs)
            v = (am | ((r >> (d - 8)).astype(np.uint64) << rs) | ((g >> (d - 8)).astype(np.uint64) << gs) | ((b >> (d - 8)).astype(np.uint64) << bs)) & 0xFFFFFFFF
            dst[:, : 4 * w] = v.astype("<u4").view(np.uint8).reshape(h, -1)
        elif out == "rg48":  # :159-178
            px = np.stack([r << (16 - d), g << (16 - d), b << (16 - d)], -1) & 0xFFFF
            dst[:, : 6 * w] = px.astype("<u2").view(np.uint8).reshape(h, -1)
        elif out == "r10k":  # :204-230
            b0 = r >> (d - 8)
            b1 = ((r >> (d - 10)) & 3) << 6 | g >> (d - 6)
            b2 = ((g >> (d - 10)) & 0xF) << 4 | b >> (d - 4)
            b3 = ((b >> (d - 10)) & 0x3F) << 2 | 3
            dst[:, : 4 * w] = (np.stack([b0, b1, b2, b3], -1) & 0xFF).astype(np.uint8).reshape(h, -1)
        elif out == "r12l":  # :60-134; samples past the end of the line packed as 0
            gw = (w + 7) // 8
            v = np.zeros((h, gw * 8, 3), np.uint32)
            v[:, :w, 0], v[:, :w, 1], v[:, :w, 2] = r >> (d - 12), g >> (d - 12), b >> (d - 12)
            dst[:, : gw * 36] = _r12l_pack(v.reshape(h, gw, 24))
        else:
            raise ValueError(name)
        return dst
    y, cb, cr = (np.asarray(p) for p in planes[:3])
    if name == "yuv444p_to_vuya":  # :565-581
        px = np.stack([cr[:h, :w], cb[:h, :w], y[:h, :w], np.full((h, w), 0xFF, np.uint8)], -1)
        dst[:, : 4 * w] = px.astype(np.uint8).reshape(h, -1)
        return dst
    if name == "yuv420_to_i420":  # :371-389
        return np.concatenate([y[:h, :w].ravel(), cb[: h // 2, : w // 2].ravel(), cr[: h // 2, : w // 2].ravel()]).astype(np.uint8)
    if name == "yuv420p_to_uyvy":
        return O.planar_to_uyvy(y[:h, :w], cb, cr, w, h, 420).reshape(h, -1)
    if name == "yuv422p10le_to_v210":
        return O.yuv422p10le_to_v210(y, cb, cr, w, h).reshape(h, -1)
    # planar 4:2:2 -> UYVY / YUYV, :391-475: width / 2 pairs, >> (depth - 8)
    pairs = w // 2
    sh = depth - 8
    ys = (y[:h, : 2 * pairs].astype(np.uint32) >> sh) & 0xFF
    u = (cb[:h, :pairs].astype(np.uint32) >> sh) & 0xFF
    v = (cr[:h, :pairs].astype(np.uint32) >> sh) & 0xFF
    q = [ys[:, 0::2], u, ys[:, 1::2], v] if name == "yuv422p_to_yuyv" else [u, ys[:, 0::2], v, ys[:, 1::2]]
    dst[:, : 4 * pairs] = np.stack(q, -1).astype(np.uint8).reshape(h, -1)
    return dst


def ref_from_planar(name: str, planes, w: int, h: int, in_depth: int = 0, rgb_shift=(0, 8, 16), out_pitch: int = 0, scalar: bool = False) -> np.ndarray:
    """the compiled reference function `name` on the same arguments"""
    pitch = out_pitch or out_linesize(name, w)
    rows = h if name != "yuv420_to_i420" else (h * 3 + 1) // 2
    out = np.zeros(rows * pitch + O.MAX_PADDING, np.uint8)
    d = O._FromPlanar()
    d.width, d.height = w, h
    d.out_data, d.out_pitch = out.ctypes.data, pitch
    keep = [np.ascontiguousarray(p) for p in planes]
    for i, pl in enumerate(keep):
        d.in_data[i] = pl.ctypes.data
        d.in_linesize[i] = pl.strides[0]
    d.in_depth = in_depth
    d.rgb_shift[0], d.rgb_shift[1], d.rgb_shift[2] = rgb_shift
    fn = getattr(O.ref(scalar), name)
    fn.restype, fn.argtypes = None, [O._FromPlanar]
    fn(d)
    if name == "yuv420_to_i420":
        return out[: w * h + 2 * (w // 2) * (h // 2)].copy()
    return out[: h * pitch].reshape(h, pitch).copy()


# ---- packed -> planar ---------------------------------------------------------------------------------------------------------------
def in_linesize(name: str, w: int) -> int:
    src = name.split("_to_")[0]
    return {"v210": O.linesize(w, "v210"), "y216": (w + 1) // 2 * 8, "uyvy": 4 * ((w + 1) // 2), "rgba": 4 * w, "vuya": 4 * w, "r12l": (w + 7) // 8 * 36}[src]


def to_shapes(name: str, w: int, h: int):
    """[(rows, samples per row, dtype)] of the output planes, tightly packed"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return {
        "v210_to_p010le": [(h, w, np.uint16), (ch, w, np.uint16)],
        "y216_to_p010le": [(h, w, np.uint16), (ch, 2 * cw, np.uint16)],
        "uyvy_to_nv12": [(h, w, np.uint8), (ch, 2 * cw, np.uint8)],
        "rgba_to_bgra": [(h, 4 * w, np.uint8)],
        "vuya_to_i444": [(h, w, np.uint8)] * 3,
        "uyvy_to_i420": [(h, w, np.uint8), (ch, cw, np.uint8), (ch, cw, np.uint8)],
    }.get(name, [(h, w, np.uint16)] * 3)


def to_planar(name: str, src: np.ndarray, w: int, h: int):
    """numpy restatement; returns the list of output planes (2-D arrays)"""
    src = np.ascontiguousarray(src, np.uint8).ravel()
    ls = in_linesize(name, w)
    if name == "v210_to_p010le":
        return list(O.v210_to_p010le(src, w, h))
    if name == "uyvy_to_nv12":
        return list(O.uyvy_to_nv12(src, w, h, src_pitch=2 * w))
    if name == "uyvy_to_i420":
        return list(O.uyvy_to_i420(src, w, h))
    rows = src[: ls * h].reshape(h, ls)
    if name == "rgba_to_bgra":  # to_planar.c:304-319
        px = rows.reshape(h, w, 4)
        return [np.ascontiguousarray(px[..., [2, 1, 0, 3]]).reshape(h, 4 * w)]
    if name == "vuya_to_i444":  # :321-337
        px = rows.reshape(h, w, 4)
        return [np.ascontiguousarray(px[..., 2]), np.ascontiguousarray(px[..., 1]), np.ascontiguousarray(px[..., 0])]
    if name == "y216_to_p010le":  # :157-203
        cw, ch = (w + 1) // 2, (h + 1) // 2
        s = rows.view("<u2").reshape(h, cw, 4)  # Y0 Cb Y1 Cr
        y = np.zeros((h, 2 * cw), np.uint16)
        y[:, 0::2], y[:, 1::2] = s[..., 0], s[..., 2]
        c = np.zeros((ch, 2 * cw), np.uint16)
        c[:, 0::2], c[:, 1::2] = s[0::2, :, 1], s[0::2, :, 3]
        return [np.ascontiguousarray(y[:, :w]), c]
    # r12l_to_gbrpXXle, :380-461: 12-bit little-endian stream r,g,b
    depth = 16 if name.endswith("16le") else 12
    gw = (w + 7) // 8
    bits = np.unpackbits(rows[:, : gw * 36], axis=1, bitorder="little").reshape(h, gw * 24, 12)
    vals = (bits.astype(np.uint32) << np.arange(12, dtype=np.uint32)).sum(-1).reshape(h, gw * 8, 3)
    vals = ((vals << (depth - 12)) & 0xFFFF).astype(np.uint16)
    r, g, b = (np.ascontiguousarray(vals[:, :w, i]) for i in range(3))
    return [g, b, r] if "gbrp" in name else [r, g, b]


def ref_to_planar(name: str, src: np.ndarray, w: int, h: int, scalar: bool = False):
    src = np.concatenate([np.ascontiguousarray(src, np.uint8).ravel(), np.zeros(O.MAX_PADDING, np.uint8)])
    d = O._ToPlanar()
    d.width, d.height = w, h
    planes = []
    for i, (rows, n, dt) in enumerate(to_shapes(name, w, h)):
        # r12l_*: room for the reference's whole-group writes past `width`.  Everything else tightly packed: y216_to_p010le
        # finds the odd luma line by running on from the even one (to_planar.c:191-199), right only when out_linesize[0] == 2 * width
        pl = np.zeros((max(rows, 1), n + (16 if name.startswith("r12l") else 0)), dt)
        planes.append(pl)
        d.out_data[i] = pl.ctypes.data
        d.out_linesize[i] = pl.strides[0]
    d.in_data = src.ctypes.data
    fn = getattr(O.ref(scalar), name)
    fn.restype, fn.argtypes = None, [O._ToPlanar]
    fn(d)
    return [np.ascontiguousarray(pl[:rows, :n]) for pl, (rows, n, _) in zip(planes, to_shapes(name, w, h))]
