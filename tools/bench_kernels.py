#!/usr/bin/env python3
"""Per-kernel roofline table for every hot-path kernel (SURVEY.md 8(d) algorithmic bytes per pixel).
Inputs resident in HBM; EVERY row rotates its sources AND its destinations over >= 2.4 GB of distinct buffers (nrot(); the rule of
tools/bench_pixfmt_all.py).  Round 3 used 600 MB -- "more than twice the 256 MB Infinity Cache" -- and that was not enough: the cache does
not replace least-recently-used lines, a 620 MB rotation still hits in it, and the rows above ~0.79 of 8 TB/s were upper bounds.
tools/rotation_sweep.py (profiles/r04_rotation_sweep.txt): the rate is flat from 1.5 GB of rotating buffers up to 9.6 GB (UYVY->v210 x8:
0.929 at 0.62 GB, 0.699 at 1.55 GB, 0.700 at 9.6 GB), so 2.4 GB is on the flat part.  HIP events on the launch stream.
Usage (GPU box): python tools/bench_kernels.py [--json out.json]"""
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
ROT_BYTES = 2.4e9


def nrot(bytes_per_call: float) -> int:
    """how many distinct (source, destination) buffer pairs a row cycles through: >= 2.4 GB in total, at least 2"""
    return max(2, int(ROT_BYTES // max(bytes_per_call, 1)) + 1)


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
    base = torch.from_numpy(np.ascontiguousarray(img)).cuda()                 # (h, ls); the n frames are row rotations made on the device
    out = torch.empty((n, h * ls), dtype=torch.uint8, device="cuda")
    for i in range(n):
        out[i] = torch.roll(base, 4 * i, dims=0).reshape(-1)
    return out


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
        src0 = frames(fmt, w, h, n)
        nb_ = nrot(src0.numel() + codec.dxt_size(oid, w, h) * n)
        srcs = [src0] + [torch.roll(src0, 4096 * (j + 1), dims=1) for j in range(nb_ - 1)]   # distinct bytes, same statistics
        dsts = [torch.empty(codec.dxt_size(oid, w, h) * n, dtype=torch.uint8, device="cuda") for _ in range(nb_)]
        kk = [0]

        def run_enc_b():
            j = kk[0] % nb_
            kk[0] += 1
            codec.dxt_encode_batch(pf, oid, srcs[j], w, h, n, src0.shape[1], dst=dsts[j])
        ms = timeit(run_enc_b)
        add(f"dxt_encode {fmt}->{out}", w, h, n, bpp, ms)
        del src0, srcs, dsts

    # ---- pixfmt (one frame per launch, rotating over n frames) ----
    for (i, o, w, h, n, bpp) in [
        ("v210", "UYVY", 7680, 4320, 6, 16 / 6 + 2), ("v210", "UYVY", 3840, 2160, 16, 16 / 6 + 2),
        ("UYVY", "RGB", 7680, 4320, 4, 5.0), ("RGB", "UYVY", 7680, 4320, 4, 5.0), ("v210", "RGB", 7680, 4320, 4, 16 / 6 + 3),
        ("UYVY", "RGB", 3840, 2160, 16, 5.0), ("UYVY", "RGB", 1920, 1080, 64, 5.0), ("RGB", "UYVY", 3840, 2160, 12, 5.0),
        ("v210", "RGB", 3840, 2160, 12, 16 / 6 + 3), ("RGBA", "RGB", 3840, 2160, 8, 7.0), ("UYVY", "YUYV", 3840, 2160, 16, 4.0),
        ("UYVY", "RGBA", 3840, 2160, 12, 6.0), ("RGB", "RGBA", 3840, 2160, 8, 7.0), ("UYVY", "v210", 3840, 2160, 12, 2 + 16 / 6),
    ]:
        n = max(n, nrot((codec.linesize(L.PF_NAMES[i], w) + codec.linesize(L.PF_NAMES[o], w)) * h))
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
        w, h, nb = 3840, 2160, 8
        l = lib.load()
        st = torch.cuda.current_stream().cuda_stream
        sls, dls = l.ug_hip_linesize(L.PF_NAMES[i], w), l.ug_hip_linesize(L.PF_NAMES[o], w)
        sets = nrot(nb * (sls + dls) * h)
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
    # every row below: `sets` distinct (inputs, outputs) groups, sets = nrot(bytes one call reads + writes); call j uses group j % sets
    w, h = 3840, 2160
    l = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    k = [0]
    u8 = lambda *shape: torch.empty(shape, dtype=torch.uint8, device="cuda")          # noqa: E731
    i16 = lambda *shape: torch.empty(shape, dtype=torch.int16, device="cuda")         # noqa: E731
    r8 = lambda *shape: torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")  # noqa: E731

    def rotating(name, n_frames, bpp, bytes_per_call, build, call, iters=40):
        """build(j) -> the buffers of group j; call(group) launches once"""
        sets = nrot(bytes_per_call)
        groups = [build(j) for j in range(sets)]

        def run_():
            j = k[0] % sets
            k[0] += 1
            rc = call(groups[j])
            assert not isinstance(rc, int) or rc == 0, (name, lib.last_error())
        add(name, w, h, n_frames, bpp, timeit(run_, iters=iters))
        del groups

    uyvy = frames("UYVY", w, h, 16)
    fb = 2 * w * h
    mw, mh = (w + 15) // 16, (h + 15) // 16
    div = codec.jpeg_divisors_device(75, "cuda")

    rotating("uyvy_to_i420", 1, 3.5, 3.5 * w * h, lambda j: (uyvy[j % 16].clone(), u8(h, w), u8(h // 2, w // 2), u8(h // 2, w // 2)),
             lambda g: l.ug_hip_uyvy_to_i420(g[0].data_ptr(), 0, g[1].data_ptr(), w, g[2].data_ptr(), w // 2, g[3].data_ptr(), w // 2, w, h, st))
    rotating("uyvy->420->FDCT+quant (fused)", 1, 5.0, 5.0 * w * h,
             lambda j: (uyvy[j % 16].clone(), i16(4 * mw * mh, 64), i16(mw * mh, 64), i16(mw * mh, 64)),
             lambda g: l.ug_hip_uyvy_to_jpeg420_coeffs(g[0].data_ptr(), 0, w, h, div.data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), st))
    nb = 8
    rotating("uyvy->420->FDCT+quant (fused, batch of 8)", nb, 5.0, 5.0 * w * h * nb,
             lambda j: (torch.roll(uyvy[:nb], 4096 * j, dims=1).contiguous(), i16(nb, 4 * mw * mh, 64), i16(nb, mw * mh, 64), i16(nb, mw * mh, 64)),
             lambda g: l.ug_hip_uyvy_to_jpeg42x_coeffs_batch(420, g[0].data_ptr(), 0, w, h, div.data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), nb, fb,
                                                             4 * mw * mh * 128, mw * mh * 128, st), iters=10)
    rotating("fdct_quant_plane (8-bit plane)", 1, 3.0, 3.0 * w * h, lambda j: (r8(h, w), i16(w // 8 * h // 8, 64)),
             lambda g: l.ug_hip_jpeg_fdct_quant_plane(g[0].data_ptr(), w, w, h, w // 8, h // 8, div.data_ptr(), g[1].data_ptr(), None, st))
    mh2 = (h + 7) // 8
    rotating("uyvy->422->FDCT+quant (fused)", 1, 6.0, 6.0 * w * h, lambda j: (uyvy[j % 16].clone(), i16(2 * mw * mh2, 64), i16(mw * mh2, 64), i16(mw * mh2, 64)),
             lambda g: l.ug_hip_uyvy_to_jpeg422_coeffs(g[0].data_ptr(), 0, w, h, div.data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), st))

    # complete encoders (FDCT + entropy + compaction; encode() is synchronous: the 4-byte length read-back is in the figure)
    import ctypes as C
    for sub, fmt_in, pf in ((420, "UYVY", L.PF_UYVY), (422, "UYVY", L.PF_UYVY), (444, "RGB", L.PF_RGB)):
        bpp_in = 2 if fmt_in == "UYVY" else 3
        nn = nrot((bpp_in + 1) * w * h)
        base = uyvy if fmt_in == "UYVY" else frames("RGB", w, h, 8)
        srcs = [base[j % base.shape[0]].clone() for j in range(nn)]
        enc = C.c_void_p()
        assert l.ug_hip_jpeg_encoder_create_sub(w, h, 75, 4, sub, C.byref(enc)) == 0
        cap = l.ug_hip_jpeg_encoder_max_size(enc)
        outbs = [u8(cap) for _ in range(4)]
        ln = C.c_size_t(0)

        def run_enc():
            j = k[0] % nn; k[0] += 1
            assert l.ug_hip_jpeg_encoder_encode(enc, pf, srcs[j].data_ptr(), 0, outbs[j % 4].data_ptr(), cap, C.byref(ln), st) == 0
        ms = timeit(run_enc, iters=2 * nn)
        add(f"jpeg encoder {fmt_in} {sub} q75 ri4 ({ln.value} B)", w, h, 1, {420: 5.0, 422: 6.0, 444: 9.0}[sub], ms)
        if fmt_in == "UYVY":   # the receive side on the stream just made: whole ug_hip_jpeg_decoder_decode call, upload of the stream included
            torch.cuda.synchronize()
            stream_bytes = bytes(outbs[(k[0] - 1) % nn % 4][: ln.value].cpu().numpy())
            dec = C.c_void_p()
            assert l.ug_hip_jpeg_decoder_create(C.byref(dec)) == 0
            nd = nrot(2 * w * h)
            dst_uyvy = [u8(2 * w * h) for _ in range(nd)]

            def run_dec():
                j = k[0] % nd; k[0] += 1
                assert l.ug_hip_jpeg_decoder_decode(dec, stream_bytes, len(stream_bytes), L.PF_UYVY, dst_uyvy[j].data_ptr(), 0, 0, 8, 16, st) == 0
            ms = timeit(run_dec, iters=16)
            # algorithmic bytes per pixel: the stream in + UYVY out (2 B/px)
            add(f"jpeg decoder {sub} q75 ri4 -> UYVY ({ln.value} B, upload included)", w, h, 1, 2.0 + ln.value / (w * h), ms)
            l.ug_hip_jpeg_decoder_destroy(dec)
            del dst_uyvy
        l.ug_hip_jpeg_encoder_destroy(enc)
        del srcs, outbs

    # decode-direction shuffles
    rotating("yuv422p_to_uyvy", 1, 4.0, 4.0 * w * h, lambda j: (r8(h, w), r8(h, w // 2), r8(h, w // 2), u8(fb)),
             lambda g: l.ug_hip_yuv422p_to_uyvy(g[0].data_ptr(), w, g[1].data_ptr(), w // 2, g[2].data_ptr(), w // 2, g[3].data_ptr(), 0, w, h, st))
    rotating("yuv420p_to_uyvy", 1, 3.5, 3.5 * w * h, lambda j: (r8(h, w), r8(h // 2, w // 2), r8(h // 2, w // 2), u8(fb)),
             lambda g: l.ug_hip_yuv420p_to_uyvy(g[0].data_ptr(), w, g[1].data_ptr(), w // 2, g[2].data_ptr(), w // 2, g[3].data_ptr(), 0, w, h, st))
    rotating("uyvy_to_i422", 1, 4.0, 4.0 * w * h, lambda j: (uyvy[j % 16].clone(), u8(h, w), u8(h, w // 2), u8(h, w // 2)),
             lambda g: l.ug_hip_uyvy_to_i422(g[0].data_ptr(), 0, g[1].data_ptr(), w, g[2].data_ptr(), w // 2, g[3].data_ptr(), w // 2, w, h, st))
    rotating("uyvy_to_nv12", 1, 3.5, 3.5 * w * h, lambda j: (uyvy[j % 16].clone(), u8(h, w), u8(h // 2, w)),
             lambda g: l.ug_hip_uyvy_to_nv12(g[0].data_ptr(), 0, g[1].data_ptr(), w, g[2].data_ptr(), w, w, h, st))
    ls210 = codec.linesize(L.PF_V210, w)
    rotating("yuv422p10le_to_v210", 1, 4 + 16 / 6, (4 + 16 / 6) * w * h,
             lambda j: (torch.randint(0, 1024, (h, w), dtype=torch.int16, device="cuda"), torch.randint(0, 1024, (h, w // 2), dtype=torch.int16, device="cuda"),
                        torch.randint(0, 1024, (h, w // 2), dtype=torch.int16, device="cuda"), u8(ls210 * h)),
             lambda g: l.ug_hip_yuv422p10le_to_v210(g[0].data_ptr(), 2 * w, g[1].data_ptr(), w, g[2].data_ptr(), w, g[3].data_ptr(), 0, w, h, st))
    v210f = frames("v210", w, h, 4)
    rotating("v210_to_p010le", 1, 16 / 6 + 3, (16 / 6 + 3) * w * h, lambda j: (torch.roll(v210f[j % 4], 0).clone(), i16(h, w), i16(h // 2, w)),
             lambda g: l.ug_hip_v210_to_p010le(g[0].data_ptr(), 0, g[1].data_ptr(), 2 * w, g[2].data_ptr(), 2 * w, w, h, st))
    rotating("v210_to_p010le 3838x2159 (ragged path)", 1, 16 / 6 + 3, (16 / 6 + 3) * w * h, lambda j: (v210f[j % 4].clone(), i16(h, w), i16(h // 2 + 1, w)),
             lambda g: l.ug_hip_v210_to_p010le(g[0].data_ptr(), ls210, g[1].data_ptr(), 2 * w, g[2].data_ptr(), 2 * w, w - 2, h - 1, st))

    # DXT decoders: one frame per launch and 8 frames per launch (compressed frames one picture apart decode as one image 8 times as tall)
    for (oid, name, outf, bpp) in ((L.DXT5_YCOCG, "DXT5", "RGBA", 5.0), (L.DXT5_YCOCG, "DXT5", "RGB", 4.0), (L.DXT5_YCOCG, "DXT5", "UYVY", 3.0), (L.DXT1, "DXT1", "RGBA", 4.5),
                                   (L.DXT1, "DXT1", "UYVY", 2.5), (L.DXT1_YUV, "DXT1_YUV", "UYVY", 2.5)):
        blocks = codec.dxt_encode_batch(L.PF_UYVY, oid, uyvy, w, h, 16, fb)
        per = codec.dxt_size(oid, w, h)
        ols = codec.linesize(L.PF_NAMES[outf], w)
        for nf in (1, 8):
            rotating(f"dxt_decode {name}->{outf}" + (" (batch of 8)" if nf == 8 else ""), nf, bpp, bpp * w * h * nf,
                     lambda j: (torch.roll(blocks[: nf * per].view(-1, 16), j, dims=0).contiguous(), u8(nf * ols * h)),   # whole blocks moved: still valid streams
                     lambda g: l.ug_hip_dxt_decode(oid, L.PF_NAMES[outf], g[0].data_ptr(), g[1].data_ptr(), w, nf * h, 0, 0, 8, 16, st), iters=10 if nf == 8 else 40)
        del blocks

    # from_planar.h / to_planar.h by name (planar_api.hip), 4K
    def p16(hi, *shape):
        return torch.randint(0, hi, shape, dtype=torch.int16, device="cuda")
    for func, kind, depth, out_bpp, in_bpp in (
            ("gbrp12le_to_rgb", "g16", 0, 3, 6), ("gbrp12le_to_rgba", "g16", 0, 4, 6), ("gbrp12le_to_rg48", "g16", 0, 6, 6),
            ("gbrp12le_to_r10k", "g16", 0, 4, 6), ("gbrp12le_to_r12l", "g16", 0, 4.5, 6), ("gbrap_to_rgba", "g8", 0, 4, 4),
            ("gbrap_to_rgb", "g8", 0, 3, 3), ("yuv444p_to_vuya", "g8", 0, 4, 3), ("yuv422p10le_to_uyvy", "c16", 0, 2, 4), ("yuv422p_to_yuyv", "c8", 0, 2, 2)):
        pitch = int(out_bpp * w)

        def build_fp(j, kind=kind, pitch=pitch):
            planes = {"g16": lambda: [p16(4096, h, w) for _ in range(3)], "g8": lambda: [r8(h, w) for _ in range(4)],
                      "c16": lambda: [p16(1024, h, w), p16(1024, h, w // 2), p16(1024, h, w // 2)], "c8": lambda: [r8(h, w), r8(h, w // 2), r8(h, w // 2)]}[kind]()
            return planes, u8(h, pitch)
        rotating(f"from_planar {func}", 1, in_bpp + out_bpp, (in_bpp + out_bpp) * w * h, build_fp,
                 lambda g, func=func, pitch=pitch, depth=depth: codec.from_planar(func, g[0], w, h, g[1], pitch, depth))
    for func, in_shape, outs_of, in_bpp, out_bpp in (
            ("r12l_to_gbrp12le", (h, w // 8 * 36), lambda: [i16(h, w) for _ in range(3)], 4.5, 6), ("r12l_to_gbrp16le", (h, w // 8 * 36), lambda: [i16(h, w) for _ in range(3)], 4.5, 6),
            ("rgba_to_bgra", (h, 4 * w), lambda: [u8(h, 4 * w)], 4, 4), ("vuya_to_i444", (h, 4 * w), lambda: [u8(h, w) for _ in range(3)], 4, 3),
            ("y216_to_p010le", (h, 4 * w), lambda: [i16(h, w), i16(h // 2, w)], 4, 3)):
        rotating(f"to_planar {func}", 1, in_bpp + out_bpp, (in_bpp + out_bpp) * w * h, lambda j, in_shape=in_shape, outs_of=outs_of: (r8(*in_shape), outs_of()),
                 lambda g, func=func: codec.to_planar(func, g[0], w, h, g[1]))

    # lavc converters (lavc_conv.hip), 4K: algorithmic bytes = input samples + output bytes
    def planes_for(av, fill):
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
            t = r8(r_, c_) if fill else u8(r_, c_)
            if fill and d16 and hi == 4:
                t[:, 1::2] &= 3
            out.append(t)
        return out, sum(r_ * c_ for r_, c_ in shp) / (w * h)
    rgbf = frames("RGB", w, h, 4)
    for uvc, av, base, in_bpp in (("v210", "yuv422p10le", v210f, 16 / 6), ("v210", "yuv420p10le", v210f, 16 / 6), ("v210", "p210le", v210f, 16 / 6), ("v210", "p010le", v210f, 16 / 6),
                                  ("UYVY", "yuv444p", uyvy, 2), ("RGB", "gbrp", rgbf, 3)):
        out_bpp = planes_for(av, False)[1]
        rotating(f"uv_to_av {uvc}->{av}", 1, in_bpp + out_bpp, (in_bpp + out_bpp) * w * h, lambda j, base=base, av=av: (base[j % base.shape[0]].clone(), planes_for(av, False)[0]),
                 lambda g, uvc=uvc, av=av: codec.uv_to_av(uvc, av, g[0], w, h, g[1]))
    for av, uvc, out_bpp in (("yuv420p", "RGB", 3), ("yuv420p", "RGBA", 4), ("yuv420p", "v210", 16 / 6), ("yuv422p", "RGBA", 4), ("yuv444p", "UYVY", 2), ("nv12", "UYVY", 2),
                             ("nv12", "RGBA", 4), ("p010le", "v210", 16 / 6), ("yuv420p10le", "v210", 16 / 6), ("yuv420p10le", "UYVY", 2), ("yuv422p10le", "RGBA", 4),
                             ("yuv444p10le", "v210", 16 / 6)):
        pitch = int(round(out_bpp * w)) if uvc != "v210" else ls210
        in_bpp = planes_for(av, False)[1]
        rotating(f"av_to_uv {av}->{uvc}", 1, in_bpp + out_bpp, (in_bpp + out_bpp) * w * h, lambda j, av=av, pitch=pitch: (planes_for(av, True)[0], u8(h, pitch)),
                 lambda g, av=av, uvc=uvc, pitch=pitch: codec.av_to_uv(av, uvc, g[0], w, h, g[1], pitch))

    # the decoders[] pairs outside the core (pixfmt_ext.hip), 4K
    BPP = {"UYVY": 2, "v210": 16 / 6, "RGB": 3, "RGBA": 4, "RG48": 6, "R10k": 4, "R12L": 4.5, "Y216": 4, "Y416": 8, "VUYA": 4}
    for i_, o_ in (("v210", "Y216"), ("UYVY", "Y216"), ("UYVY", "RG48"), ("R10k", "RGBA"), ("R10k", "UYVY"), ("R12L", "RGB"), ("R12L", "UYVY"), ("RG48", "R12L"),
                   ("RG48", "v210"), ("Y416", "UYVY"), ("Y416", "RGBA"), ("RGBA", "R10k"), ("VUYA", "RGB")):
        ls_in = l.ug_hip_linesize(L.PF_NAMES[i_], w)
        ls_out = l.ug_hip_linesize(L.PF_NAMES[o_], w)
        rotating(f"pixfmt {i_}->{o_}", 1, BPP[i_] + BPP[o_], (ls_in + ls_out) * h, lambda j, ls_in=ls_in, ls_out=ls_out: (r8(h * ls_in + 64), u8(h * ls_out + 64)),
                 lambda g, i_=i_, o_=o_: l.ug_hip_pixfmt_convert(L.PF_NAMES[i_], L.PF_NAMES[o_], g[0].data_ptr(), g[1].data_ptr(), w, h, 0, 0, 0, 8, 16, st))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
