"""Seeded synthetic frames, shared by tests/ and bench.py.

All frames are tightly packed with the reference line sizes (vc_get_linesize,
video_codec.c:507-521).  numpy only -- no device code.
"""
from __future__ import annotations

import numpy as np

SEED_S1 = 0x55470001
SEED_S2 = 0x55470002


def linesize(fmt: str, w: int) -> int:
    if fmt in ("UYVY", "YUYV"):
        return (w + 1) // 2 * 4
    if fmt in ("RGB", "BGR", "YUV444"):
        return 3 * w
    if fmt == "RGBA":
        return 4 * w
    if fmt == "RG48":
        return 6 * w
    if fmt == "v210":
        return (w + 47) // 48 * 128
    raise ValueError(fmt)


def _rng(seed: int, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed + salt))


def _mask_v210(buf: np.ndarray) -> np.ndarray:
    w32 = buf.view(np.uint32) & np.uint32(0x3FFFFFFF)  # pad bits 0
    return w32.view(np.uint8)


def s1_random(fmt: str, w: int, h: int, salt: int = 0) -> np.ndarray:
    """S1: uniform random bytes (every block has max range; exercises clamps / all indices)."""
    n = linesize(fmt, w) * h
    buf = _rng(SEED_S1, salt).integers(0, 256, n, dtype=np.uint8)
    return _mask_v210(buf) if fmt == "v210" else buf


def _smooth_planes(w: int, h: int, salt: int):
    """Legal-range Y (full res), U, V (half horizontal res) video-like noise: 8x8 box-filtered + N(0,2)."""
    rng = _rng(SEED_S2, salt)

    def plane(pw, lo, hi):
        coarse = rng.uniform(lo, hi, ((h + 7) // 8 + 2, (pw + 7) // 8 + 2))
        # bilinear upsample of the coarse grid = low-pass content
        ys = (np.arange(h) + 0.5) / 8.0
        xs = (np.arange(pw) + 0.5) / 8.0
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        a = coarse[y0][:, x0]; b = coarse[y0][:, x0 + 1]; c = coarse[y0 + 1][:, x0]; d = coarse[y0 + 1][:, x0 + 1]
        p = a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx
        return np.clip(p + rng.normal(0, 2, p.shape), lo, hi)

    cw = (w + 1) // 2
    return plane(w, 16, 235), plane(cw, 16, 240), plane(cw, 16, 240)


def s2_video(fmt: str, w: int, h: int, salt: int = 0) -> np.ndarray:
    """S2: legal-range low-pass video noise (typical content; exercises scale 2/4 in ScaleYCoCg)."""
    y, u, v = _smooth_planes(w, h, salt)
    cw = (w + 1) // 2
    if fmt == "UYVY":
        out = np.zeros((h, cw, 4), np.uint8)
        yy = np.zeros((h, cw * 2))
        yy[:, :w] = y
        out[..., 0] = u.round(); out[..., 1] = yy[:, 0::2].round(); out[..., 2] = v.round(); out[..., 3] = yy[:, 1::2].round()
        return out.ravel()
    if fmt == "v210":
        assert w % 6 == 0
        y10 = (y * 4).round().astype(np.uint32); u10 = (u * 4).round().astype(np.uint32); v10 = (v * 4).round().astype(np.uint32)
        # sample stream in UYVY order, 3 samples per word
        s = np.zeros((h, w * 2), np.uint32)
        s[:, 0::4] = u10; s[:, 1::4] = y10[:, 0::2]; s[:, 2::4] = v10; s[:, 3::4] = y10[:, 1::2]
        words = s[:, 0::3] | (s[:, 1::3] << 10) | (s[:, 2::3] << 20)
        ls = linesize("v210", w)
        out = np.zeros((h, ls // 4), np.uint32)
        out[:, : words.shape[1]] = words
        return out.view(np.uint8).ravel()
    if fmt in ("RGB", "RGBA", "BGR"):
        # BT.709 limited -> RGB in float, good enough for content (not a parity path)
        yy = (y - 16) * 1.1643
        uu = np.repeat(u, 2, axis=1)[:, :w] - 128
        vv = np.repeat(v, 2, axis=1)[:, :w] - 128
        r = yy + 1.7926 * vv; g = yy - 0.2132 * uu - 0.5328 * vv; b = yy + 2.1124 * uu
        rgb = np.clip(np.stack([r, g, b], -1), 0, 255).round().astype(np.uint8)
        if fmt == "BGR":
            rgb = rgb[..., ::-1]
        if fmt == "RGBA":
            rgb = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], -1)
        return np.ascontiguousarray(rgb).ravel()
    raise ValueError(fmt)


def s3_bars(fmt: str, w: int, h: int) -> np.ndarray:
    """S3: 8 colour bars + horizontal luma ramp in the lower third (testcard-like, seed-free)."""
    bars = np.array([[235, 128, 128], [210, 16, 146], [170, 166, 16], [145, 54, 34],
                     [106, 202, 222], [81, 90, 240], [41, 240, 110], [16, 128, 128]], np.float64)
    x = np.arange(w)
    idx = np.minimum(x * 8 // max(w, 1), 7)
    y = np.tile(bars[idx, 0], (h, 1)); u = np.tile(bars[idx, 1], (h, 1)); v = np.tile(bars[idx, 2], (h, 1))
    ramp = 16 + (235 - 16) * x / max(w - 1, 1)
    y[2 * h // 3:] = ramp; u[2 * h // 3:] = 128; v[2 * h // 3:] = 128
    if fmt == "UYVY":
        cw = (w + 1) // 2
        out = np.zeros((h, cw, 4), np.uint8)
        yy = np.zeros((h, cw * 2)); yy[:, :w] = y
        uu = np.zeros((h, cw * 2)); uu[:, :w] = u
        vv = np.zeros((h, cw * 2)); vv[:, :w] = v
        out[..., 0] = uu[:, 0::2].round(); out[..., 1] = yy[:, 0::2].round(); out[..., 2] = vv[:, 0::2].round(); out[..., 3] = yy[:, 1::2].round()
        return out.ravel()
    if fmt == "RGB":
        yy = (y - 16) * 1.1643
        r = yy + 1.7926 * (v - 128); g = yy - 0.2132 * (u - 128) - 0.5328 * (v - 128); b = yy + 2.1124 * (u - 128)
        return np.ascontiguousarray(np.clip(np.stack([r, g, b], -1), 0, 255).round().astype(np.uint8)).ravel()
    raise ValueError(fmt)


def s4_flat(fmt: str, w: int, h: int, value: int = 127) -> np.ndarray:
    """S4: flat frame (test/gpujpeg_test.cpp:78 fixture; degenerate min == max blocks)."""
    buf = np.full(linesize(fmt, w) * h, value, np.uint8)
    return _mask_v210(buf) if fmt == "v210" else buf


def frame(kind: str, fmt: str, w: int, h: int, salt: int = 0) -> np.ndarray:
    if kind == "S1":
        return s1_random(fmt, w, h, salt)
    if kind == "S2":
        return s2_video(fmt, w, h, salt)
    if kind == "S3":
        return s3_bars(fmt, w, h)
    if kind == "S4":
        return s4_flat(fmt, w, h)
    raise ValueError(kind)
