#!/usr/bin/env python3
"""Per-kernel roofline table for every hot-path kernel (SURVEY.md 8(d) algorithmic bytes per pixel).
Inputs resident in HBM, >= 256 MB working set per launch sequence (rotating distinct frames), HIP events on the
launch stream.  Usage (GPU box): python tools/bench_kernels.py [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ultragrid_amd import codec, lib, synth

L = lib
PEAK = 8000.0


def timeit(fn, iters=30, warm=3, min_ms=120.0):
    """ms per call: HIP events on the launch stream around back-to-back calls, repeated until at least `min_ms` of GPU work has been
    timed (a handful of 10 us launches measures the clock ramp, not the kernel: the same DXT encode reads 0.219 ms over 30 launches and
    0.180 ms in steady state), after an untimed warm-up of a third of that."""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(max(warm, 1)):
        fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) / max(warm, 1), 1e-3)
    for _ in range(int(min_ms / 3 / one)):
        fn()
    n = max(iters, int(min_ms / one))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def frames(fmt, w, h, n):
    """n distinct frames of S2-like content derived cheaply from one base (row rotations)."""
    base_fmt = {"YUV444": "RGB", "UYVY_RAW": "UYVY"}.get(fmt, fmt)
    try:
        one = synth.s2_video(base_fmt, w, min(h, 64))
    except (ValueError, AssertionError):
        one = synth.s1_random(base_fmt, w, min(h, 64))
    ls = one.size // min(h, 64)
    img = np.tile(one.reshape(min(h, 64), ls), ((h + 63) // 64, 1))[:h]
    out = np.stack([np.roll(img, 4 * i, axis=0) for i in range(n)])
    return torch.from_numpy(out.reshape(n, -1)).cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    lib.load()
    rows = []

    def add(name, w, h, n, bytes_per_px, ms):
        px = w * h * n
        gbs = bytes_per_px * px / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "size": f"{w}x{h}x{n}", "ms_per_launch": round(ms, 4), "Mpix_s": round(px / ms / 1e3, 1),
                     "fps": round(n / (ms * 1e-3), 1), "alg_B_per_px": bytes_per_px, "GB_s": round(gbs, 1), "frac_8TBs": round(gbs / PEAK, 4)})
        print(f"{name:34s} {w}x{h} x{n:<3d} {ms:8.4f} ms  {px / ms / 1e3:10.1f} Mpx/s  {gbs:7.1f} GB/s  {gbs / PEAK:6.3f}", flush=True)

    # ---- DXT encoders (batched launch over n frames) ----
    for (fmt, pf, out, oid, w, h, n, bpp) in [
        ("UYVY", L.PF_UYVY, "DXT5", L.DXT5_YCOCG, 3840, 2160, 16, 3.0),
        ("UYVY", L.PF_UYVY, "DXT5", L.DXT5_YCOCG, 1920, 1080, 64, 3.0),
        ("UYVY", L.PF_UYVY, "DXT5", L.DXT5_YCOCG, 7680, 4320, 4, 3.0),
        ("v210", L.PF_V210, "DXT5", L.DXT5_YCOCG, 7680, 4320, 4, 16 / 6 + 1),
        ("RGB", L.PF_RGB, "DXT1", L.DXT1, 1920, 1080, 64, 3.5),
        ("RGB", L.PF_RGB, "DXT5", L.DXT5_YCOCG, 3840, 2160, 12, 4.0),
        ("UYVY", L.PF_UYVY, "DXT1", L.DXT1, 3840, 2160, 16, 2.5),
        ("RGBA", L.PF_RGBA, "DXT1", L.DXT1, 3840, 2160, 8, 4.5),
    ]:
        src = frames(fmt, w, h, n)
        dst = torch.empty(codec.dxt_size(oid, w, h) * n, dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: codec.dxt_encode_batch(pf, oid, src, w, h, n, src.shape[1], dst=dst))
        add(f"dxt_encode {fmt}->{out}", w, h, n, bpp, ms)
        del src, dst

    # ---- pixfmt (one frame per launch, rotating over n frames) ----
    for (i, o, w, h, n, bpp) in [
        ("v210", "UYVY", 7680, 4320, 6, 16 / 6 + 2), ("v210", "UYVY", 3840, 2160, 16, 16 / 6 + 2),
        ("UYVY", "RGB", 7680, 4320, 4, 5.0), ("RGB", "UYVY", 7680, 4320, 4, 5.0), ("v210", "RGB", 7680, 4320, 4, 16 / 6 + 3),
        ("UYVY", "RGB", 3840, 2160, 16, 5.0), ("UYVY", "RGB", 1920, 1080, 64, 5.0), ("RGB", "UYVY", 3840, 2160, 12, 5.0),
        ("v210", "RGB", 3840, 2160, 12, 16 / 6 + 3), ("RGBA", "RGB", 3840, 2160, 8, 7.0), ("UYVY", "YUYV", 3840, 2160, 16, 4.0),
        ("UYVY", "RGBA", 3840, 2160, 12, 6.0), ("RGB", "RGBA", 3840, 2160, 8, 7.0), ("UYVY", "v210", 3840, 2160, 12, 2 + 16 / 6),
    ]:
        src = frames(i, w, h, n)
        dsts = torch.empty((n, codec.linesize(L.PF_NAMES[o], w) * h), dtype=torch.uint8, device="cuda")
        k = [0]
        l = lib.load()
        st = torch.cuda.current_stream().cuda_stream

        def run():
            j = k[0] % n
            k[0] += 1
            rc = l.ug_hip_pixfmt_convert(L.PF_NAMES[i], L.PF_NAMES[o], src[j].data_ptr(), dsts[j].data_ptr(), w, h, 0, 0, 0, 8, 16, st)
            assert rc == 0
        ms = timeit(run, iters=3 * n)
        add(f"pixfmt {i}->{o}", w, h, 1, bpp, ms)
        del src, dsts

    # ---- pixfmt, 8 frames of 4K per launch (ug_hip_pixfmt_convert_batch: frames one picture apart = one launch) ----
    for (i, o, bpp) in [("v210", "UYVY", 16 / 6 + 2), ("UYVY", "RGB", 5.0), ("RGB", "UYVY", 5.0), ("v210", "RGB", 16 / 6 + 3), ("RGBA", "RGB", 7.0),
                        ("UYVY", "RGBA", 6.0), ("UYVY", "v210", 2 + 16 / 6), ("UYVY", "Y216", 6.0), ("Y216", "UYVY", 6.0), ("R10k", "RGB", 7.0), ("RG48", "RGB", 9.0)]:
        w, h, nb, sets = 3840, 2160, 8, 3
        l = lib.load()
        st = torch.cuda.current_stream().cuda_stream
        sls, dls = l.ug_hip_linesize(L.PF_NAMES[i], w), l.ug_hip_linesize(L.PF_NAMES[o], w)
        try:
            src = torch.stack([frames(i, w, h, nb) for _ in range(sets)])
        except (ValueError, AssertionError, KeyError):
            src = torch.randint(0, 256, (sets, nb, sls * h), dtype=torch.uint8, device="cuda")
        dsts = torch.empty((sets, nb * dls * h), dtype=torch.uint8, device="cuda")
        k = [0]

        def run_b():
            j = k[0] % sets
            k[0] += 1
            rc = l.ug_hip_pixfmt_convert_batch(L.PF_NAMES[i], L.PF_NAMES[o], src[j].data_ptr(), dsts[j].data_ptr(), w, h, 0, 0, 0, 8, 16, nb, sls * h, dls * h, st)
            assert rc == 0
        ms = timeit(run_b, iters=9)
        add(f"pixfmt {i}->{o} (batch of 8)", w, h, nb, bpp, ms)
        del src, dsts

    # ---- planar + JPEG ----
    w, h, n = 3840, 2160, 16
    src = frames("UYVY", w, h, n)
    y = torch.empty((h, w), dtype=torch.uint8, device="cuda"); u = torch.empty((h // 2, w // 2), dtype=torch.uint8, device="cuda"); v = torch.empty_like(u)
    l = lib.load(); st = torch.cuda.current_stream().cuda_stream
    k = [0]

    def run_i420():
        j = k[0] % n; k[0] += 1
        assert l.ug_hip_uyvy_to_i420(src[j].data_ptr(), 0, y.data_ptr(), w, u.data_ptr(), w // 2, v.data_ptr(), w // 2, w, h, st) == 0
    add("uyvy_to_i420", w, h, 1, 3.5, timeit(run_i420, iters=3 * n))
    div = codec.jpeg_divisors_device(75, "cuda")
    mw, mh = (w + 15) // 16, (h + 15) // 16
    oy = torch.empty((4 * mw * mh, 64), dtype=torch.int16, device="cuda"); ocb = torch.empty((mw * mh, 64), dtype=torch.int16, device="cuda"); ocr = torch.empty_like(ocb)

    def run_jpeg():
        j = k[0] % n; k[0] += 1
        assert l.ug_hip_uyvy_to_jpeg420_coeffs(src[j].data_ptr(), 0, w, h, div.data_ptr(), oy.data_ptr(), ocb.data_ptr(), ocr.data_ptr(), st) == 0
    add("uyvy->420->FDCT+quant (fused)", w, h, 1, 5.0, timeit(run_jpeg, iters=3 * n))
    # the same front end over 8 frames per launch (ug_hip_uyvy_to_jpeg42x_coeffs_batch, grid.z = frame)
    nb = 8
    boy = torch.empty((nb, 4 * mw * mh, 64), dtype=torch.int16, device="cuda"); bocb = torch.empty((nb, mw * mh, 64), dtype=torch.int16, device="cuda"); bocr = torch.empty_like(bocb)

    def run_jpeg_b():
        j = (k[0] % 2) * nb; k[0] += 1
        assert l.ug_hip_uyvy_to_jpeg42x_coeffs_batch(420, src[j].data_ptr(), 0, w, h, div.data_ptr(), boy.data_ptr(), bocb.data_ptr(), bocr.data_ptr(), nb, src.shape[1],
                                                     4 * mw * mh * 128, mw * mh * 128, st) == 0
    add("uyvy->420->FDCT+quant (fused, batch of 8)", w, h, nb, 5.0, timeit(run_jpeg_b, iters=10))
    del boy, bocb, bocr
    plane = torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda")
    outp = torch.empty((w // 8 * h // 8, 64), dtype=torch.int16, device="cuda")

    def run_plane():
        assert l.ug_hip_jpeg_fdct_quant_plane(plane.data_ptr(), w, w, h, w // 8, h // 8, div.data_ptr(), outp.data_ptr(), None, st) == 0
    add("fdct_quant_plane (8-bit plane)", w, h, 1, 3.0, timeit(run_plane))
    oy2 = torch.empty((2 * mw * ((h + 7) // 8), 64), dtype=torch.int16, device="cuda"); ocb2 = torch.empty((mw * ((h + 7) // 8), 64), dtype=torch.int16, device="cuda"); ocr2 = torch.empty_like(ocb2)

    def run_jpeg422():
        j = k[0] % n; k[0] += 1
        assert l.ug_hip_uyvy_to_jpeg422_coeffs(src[j].data_ptr(), 0, w, h, div.data_ptr(), oy2.data_ptr(), ocb2.data_ptr(), ocr2.data_ptr(), st) == 0
    add("uyvy->422->FDCT+quant (fused)", w, h, 1, 6.0, timeit(run_jpeg422, iters=3 * n))

    # complete encoders (FDCT + entropy + compaction; excludes the 4-byte length read-back sync cost? no: encode() is synchronous)
    import ctypes as C
    for sub, fmt_in, pf in ((420, "UYVY", L.PF_UYVY), (422, "UYVY", L.PF_UYVY), (444, "RGB", L.PF_RGB)):
        srcs = src if fmt_in == "UYVY" else frames("RGB", w, h, 8)
        nn = srcs.shape[0]
        enc = C.c_void_p()
        assert l.ug_hip_jpeg_encoder_create_sub(w, h, 75, 4, sub, C.byref(enc)) == 0
        cap = l.ug_hip_jpeg_encoder_max_size(enc)
        outb = torch.empty(cap, dtype=torch.uint8, device="cuda")
        ln = C.c_size_t(0)

        def run_enc():
            j = k[0] % nn; k[0] += 1
            assert l.ug_hip_jpeg_encoder_encode(enc, pf, srcs[j].data_ptr(), 0, outb.data_ptr(), cap, C.byref(ln), st) == 0
        ms = timeit(run_enc, iters=2 * nn)
        add(f"jpeg encoder {fmt_in} {sub} q75 ri4 ({ln.value} B)", w, h, 1, {420: 5.0, 422: 6.0, 444: 9.0}[sub], ms)
        if fmt_in == "UYVY":   # the receive side on the stream just made: whole ug_hip_jpeg_decoder_decode call, upload of the stream included
            torch.cuda.synchronize()
            stream_bytes = bytes(outb[: ln.value].cpu().numpy())
            dec = C.c_void_p()
            assert l.ug_hip_jpeg_decoder_create(C.byref(dec)) == 0
            dst_uyvy = torch.empty(2 * w * h, dtype=torch.uint8, device="cuda")

            def run_dec():
                assert l.ug_hip_jpeg_decoder_decode(dec, stream_bytes, len(stream_bytes), L.PF_UYVY, dst_uyvy.data_ptr(), 0, 0, 8, 16, st) == 0
            ms = timeit(run_dec, iters=16)
            # algorithmic bytes per pixel: the stream in + UYVY out (2 B/px)
            add(f"jpeg decoder {sub} q75 ri4 -> UYVY ({ln.value} B, upload included)", w, h, 1, 2.0 + ln.value / (w * h), ms)
            l.ug_hip_jpeg_decoder_destroy(dec)
        l.ug_hip_jpeg_encoder_destroy(enc)

    # decode-direction shuffles
    yy, uu, vv = codec.uyvy_to_i422(src[0], w, h)
    dstu = torch.empty(2 * w * h, dtype=torch.uint8, device="cuda")

    def run_p422():
        assert l.ug_hip_yuv422p_to_uyvy(yy.data_ptr(), w, uu.data_ptr(), w // 2, vv.data_ptr(), w // 2, dstu.data_ptr(), 0, w, h, st) == 0
    add("yuv422p_to_uyvy", w, h, 1, 4.0, timeit(run_p422))

    def run_p420():
        assert l.ug_hip_yuv420p_to_uyvy(y.data_ptr(), w, u.data_ptr(), w // 2, v.data_ptr(), w // 2, dstu.data_ptr(), 0, w, h, st) == 0
    add("yuv420p_to_uyvy", w, h, 1, 3.5, timeit(run_p420))

    def run_i422():
        j = k[0] % n; k[0] += 1
        assert l.ug_hip_uyvy_to_i422(src[j].data_ptr(), 0, yy.data_ptr(), w, uu.data_ptr(), w // 2, vv.data_ptr(), w // 2, w, h, st) == 0
    add("uyvy_to_i422", w, h, 1, 4.0, timeit(run_i422, iters=3 * n))
    cplane = torch.empty((h // 2, w), dtype=torch.uint8, device="cuda")

    def run_nv12():
        j = k[0] % n; k[0] += 1
        assert l.ug_hip_uyvy_to_nv12(src[j].data_ptr(), 0, yy.data_ptr(), w, cplane.data_ptr(), w, w, h, st) == 0
    add("uyvy_to_nv12", w, h, 1, 3.5, timeit(run_nv12, iters=3 * n))
    y10 = torch.randint(0, 1024, (h, w), dtype=torch.int16, device="cuda"); u10 = torch.randint(0, 1024, (h, w // 2), dtype=torch.int16, device="cuda"); v10 = torch.randint(0, 1024, (h, w // 2), dtype=torch.int16, device="cuda")
    dv = torch.empty(codec.linesize(L.PF_V210, w) * h, dtype=torch.uint8, device="cuda")

    def run_v210():
        assert l.ug_hip_yuv422p10le_to_v210(y10.data_ptr(), 2 * w, u10.data_ptr(), w, v10.data_ptr(), w, dv.data_ptr(), 0, w, h, st) == 0
    add("yuv422p10le_to_v210", w, h, 1, 4 + 16 / 6, timeit(run_v210))

    # DXT decoders
    for (oid, name, outf, bpp) in ((L.DXT5_YCOCG, "DXT5", "RGBA", 5.0), (L.DXT5_YCOCG, "DXT5", "UYVY", 3.0), (L.DXT1, "DXT1", "RGBA", 4.5), (L.DXT1_YUV, "DXT1_YUV", "UYVY", 2.5)):
        blocks = codec.dxt_encode_batch(L.PF_UYVY, oid, src, w, h, n, src.shape[1])
        per = codec.dxt_size(oid, w, h)
        dd = torch.empty(codec.linesize(L.PF_NAMES[outf], w) * h, dtype=torch.uint8, device="cuda")

        def run_dec():
            j = k[0] % n; k[0] += 1
            assert l.ug_hip_dxt_decode(oid, L.PF_NAMES[outf], blocks.data_ptr() + j * per, dd.data_ptr(), w, h, 0, 0, 8, 16, st) == 0
        add(f"dxt_decode {name}->{outf}", w, h, 1, bpp, timeit(run_dec, iters=3 * n))
        # 8 frames per launch: compressed frames one picture apart decode as one image 8 times as tall
        dd8 = torch.empty(8 * codec.linesize(L.PF_NAMES[outf], w) * h, dtype=torch.uint8, device="cuda")

        def run_dec8():
            j = (k[0] % 2) * 8; k[0] += 1
            assert l.ug_hip_dxt_decode(oid, L.PF_NAMES[outf], blocks.data_ptr() + j * per, dd8.data_ptr(), w, 8 * h, 0, 0, 8, 16, st) == 0
        add(f"dxt_decode {name}->{outf} (batch of 8)", w, h, 8, bpp, timeit(run_dec8, iters=10))
        del blocks, dd8
    # from_planar.h / to_planar.h by name (planar_api.hip): 4K, rotating 4 plane sets so L2 does not hold the input
    NP = 4
    g16 = [[torch.randint(0, 4096, (h, w), dtype=torch.int16, device="cuda") for _ in range(3)] for _ in range(NP)]
    g8 = [[torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda") for _ in range(4)] for _ in range(NP)]
    c16 = [[torch.randint(0, 1024, (h, w), dtype=torch.int16, device="cuda")] + [torch.randint(0, 1024, (h, w // 2), dtype=torch.int16, device="cuda") for _ in range(2)] for _ in range(NP)]
    for func, planes, depth, out_bpp, in_bpp in (
            ("gbrp12le_to_rgb", g16, 0, 3, 6), ("gbrp12le_to_rgba", g16, 0, 4, 6), ("gbrp12le_to_rg48", g16, 0, 6, 6),
            ("gbrp12le_to_r10k", g16, 0, 4, 6), ("gbrp12le_to_r12l", g16, 0, 4.5, 6), ("gbrap_to_rgba", g8, 0, 4, 4),
            ("gbrap_to_rgb", g8, 0, 3, 3), ("yuv444p_to_vuya", g8, 0, 4, 3), ("yuv422p10le_to_uyvy", c16, 0, 2, 4), ("yuv422p_to_yuyv", None, 0, 2, 2)):
        if planes is None:
            planes = [[yy, uu, vv]] * NP
        pitch = int(out_bpp * w)
        outb = torch.empty((h, pitch), dtype=torch.uint8, device="cuda")

        def run_fp():
            j = k[0] % NP; k[0] += 1
            codec.from_planar(func, planes[j], w, h, outb, pitch, depth)
        add(f"from_planar {func}", w, h, 1, in_bpp + out_bpp, timeit(run_fp, iters=40))
    r12 = [torch.randint(0, 256, (h, w // 8 * 36), dtype=torch.uint8, device="cuda") for _ in range(NP)]
    p16 = [torch.empty((h, w), dtype=torch.int16, device="cuda") for _ in range(3)]
    rg = [torch.randint(0, 256, (h, 4 * w), dtype=torch.uint8, device="cuda") for _ in range(NP)]
    p8 = [torch.empty((h, w), dtype=torch.uint8, device="cuda") for _ in range(3)]
    y2 = [torch.randint(0, 256, (h, 4 * w), dtype=torch.uint8, device="cuda") for _ in range(NP)]  # Y216: 4 B / px
    for func, srcs, outs, in_bpp, out_bpp in (
            ("r12l_to_gbrp12le", r12, p16, 4.5, 6), ("r12l_to_gbrp16le", r12, p16, 4.5, 6), ("rgba_to_bgra", rg, [torch.empty((h, 4 * w), dtype=torch.uint8, device="cuda")], 4, 4),
            ("vuya_to_i444", rg, p8, 4, 3), ("y216_to_p010le", y2, [p16[0], torch.empty((h // 2, w), dtype=torch.int16, device="cuda")], 4, 3)):
        def run_tp():
            j = k[0] % NP; k[0] += 1
            codec.to_planar(func, srcs[j], w, h, outs)
        add(f"to_planar {func}", w, h, 1, in_bpp + out_bpp, timeit(run_tp, iters=40))
    # lavc converters (lavc_conv.hip), 4K: algorithmic bytes = input samples + output bytes
    def planes_for(av):
        d16 = any(t in av for t in ("10le", "12le", "16le", "p010", "p210"))
        bps = 2 if d16 else 1
        cw, ch = {"420": (w // 2, h // 2), "422": (w // 2, h), "444": (w, h)}.get(av[3:6], (w, h))
        if av in ("nv12", "p010le"):
            shp = [(h, w * bps), (h // 2, w * bps)]
        elif av == "p210le":
            shp = [(h, w * bps), (h, w * bps)]
        elif av.startswith("yuv"):
            shp = [(h, w * bps), (ch, cw * bps), (ch, cw * bps)]
        else:
            shp = [(h, w * bps)] * 3
        hi = 4 if d16 and "p0" not in av and "p2" not in av else 256  # keep 10-bit samples in range (high byte < 4)
        out = []
        for (r_, c_) in shp:
            t = torch.randint(0, 256, (r_, c_), dtype=torch.uint8, device="cuda")
            if d16 and hi == 4:
                t[:, 1::2] &= 3
            out.append(t)
        return out, sum(r_ * c_ for r_, c_ in shp) / (w * h)
    v210src = frames("v210", w, h, NP)
    for uvc, av, srcs, in_bpp in (("v210", "yuv422p10le", v210src, 16 / 6), ("v210", "yuv420p10le", v210src, 16 / 6), ("v210", "p210le", v210src, 16 / 6),
                                  ("UYVY", "yuv444p", src, 2), ("RGB", "gbrp", frames("RGB", w, h, NP), 3)):
        outs, out_bpp = planes_for(av)

        def run_to():
            j = k[0] % NP; k[0] += 1
            codec.uv_to_av(uvc, av, srcs[j], w, h, outs)
        add(f"uv_to_av {uvc}->{av}", w, h, 1, in_bpp + out_bpp, timeit(run_to, iters=40))
    for av, uvc, out_bpp in (("yuv420p", "RGB", 3), ("yuv420p", "RGBA", 4), ("yuv420p", "v210", 16 / 6), ("yuv422p", "RGBA", 4), ("yuv444p", "UYVY", 2), ("nv12", "UYVY", 2),
                             ("nv12", "RGBA", 4), ("p010le", "v210", 16 / 6), ("yuv420p10le", "v210", 16 / 6), ("yuv420p10le", "UYVY", 2), ("yuv422p10le", "RGBA", 4),
                             ("yuv444p10le", "v210", 16 / 6)):
        sets = [planes_for(av) for _ in range(NP)]
        pitch = int(round(out_bpp * w)) if uvc != "v210" else codec.linesize(L.PF_V210, w)
        dstb = torch.empty((h, pitch), dtype=torch.uint8, device="cuda")

        def run_from():
            j = k[0] % NP; k[0] += 1
            codec.av_to_uv(av, uvc, sets[j][0], w, h, dstb, pitch)
        add(f"av_to_uv {av}->{uvc}", w, h, 1, sets[0][1] + out_bpp, timeit(run_from, iters=40))
    # the decoders[] pairs outside the core (pixfmt_ext.hip), 4K
    BPP = {"UYVY": 2, "v210": 16 / 6, "RGB": 3, "RGBA": 4, "RG48": 6, "R10k": 4, "R12L": 4.5, "Y216": 4, "Y416": 8, "VUYA": 4}
    for i_, o_ in (("v210", "Y216"), ("UYVY", "Y216"), ("UYVY", "RG48"), ("R10k", "RGBA"), ("R10k", "UYVY"), ("R12L", "RGB"), ("R12L", "UYVY"), ("RG48", "R12L"),
                   ("RG48", "v210"), ("Y416", "UYVY"), ("Y416", "RGBA"), ("RGBA", "R10k"), ("VUYA", "RGB")):
        ls_in = l.ug_hip_linesize(L.PF_NAMES[i_], w)
        ls_out = l.ug_hip_linesize(L.PF_NAMES[o_], w)
        srcs = [torch.randint(0, 256, (h * ls_in + 64,), dtype=torch.uint8, device="cuda") for _ in range(NP)]
        dd = torch.empty(h * ls_out + 64, dtype=torch.uint8, device="cuda")

        def run_ext():
            j = k[0] % NP; k[0] += 1
            assert l.ug_hip_pixfmt_convert(L.PF_NAMES[i_], L.PF_NAMES[o_], srcs[j].data_ptr(), dd.data_ptr(), w, h, 0, 0, 0, 8, 16, st) == 0
        add(f"pixfmt {i_}->{o_}", w, h, 1, BPP[i_] + BPP[o_], timeit(run_ext, iters=40))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
