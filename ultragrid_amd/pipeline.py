"""Host-side frame pipeline of one GPU, as the UltraGrid modules drive it (module/mi355x_frame_sharder.h: `workers` frames in
flight per device, each on its own stream): pinned host frame -> H2D -> fused encode kernel -> D2H of the compressed frame.
Used by bench.py's `e2e` leg and tools/e2e_bench.py to measure the PCIe-inclusive rate (never bench.py's `value`).
torch = device memory, pinned memory and streams only; the kernel goes through the C ABI (ultragrid_amd/codec.py)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import codec, lib, synth

WORKLOADS = {
    "8k-v210": ("v210", lib.PF_V210, lib.DXT5_YCOCG, 7680, 4320),     # BASELINE.json configs[4]
    "8k-uyvy": ("UYVY", lib.PF_UYVY, lib.DXT5_YCOCG, 7680, 4320),     # the north star's target: >= 60 fps 8K UYVY -> DXT5-YCoCg on one GPU
    "4k-uyvy": ("UYVY", lib.PF_UYVY, lib.DXT5_YCOCG, 3840, 2160),     # configs[2]
    "1080p-rgb-dxt1": ("RGB", lib.PF_RGB, lib.DXT1, 1920, 1080),      # configs[1]
}


def gpu_numa_node(device_index: int) -> int:
    """NUMA node of the GPU's PCIe function (sysfs), -1 if unknown."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        return -1
    p = f"/sys/bus/pci/devices/{bdf}/numa_node"
    if os.path.exists(p):
        try:
            return int(open(p).read().strip())
        except ValueError:
            return -1
    return -1


def bind_to_numa_node(node: int) -> int:
    """Restrict this process to the CPUs of `node` (so that first-touch puts the pinned frames into node-local DRAM and the
    submitting thread runs beside them).  Returns the number of CPUs bound to, 0 if nothing was changed."""
    if node < 0 or not hasattr(os, "sched_setaffinity"):
        return 0
    p = f"/sys/devices/system/node/node{node}/cpulist"
    if not os.path.exists(p):
        return 0
    cpus = set()
    for part in open(p).read().strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return 0
    os.sched_setaffinity(0, cpus)
    return len(cpus)


def host_frame(workload: str, salt: int = 0) -> np.ndarray:
    fmt, _, _, w, h = WORKLOADS[workload]
    gen = synth.s1_random if fmt == "RGB" else synth.s2_video
    one = gen(fmt, w, 48, salt=salt)
    ls = one.size // 48
    return np.tile(one.reshape(48, ls), (h // 48, 1)).ravel().copy()


def _rate_loop(submit, slots, seconds: float, min_frames: int):
    """submit(slot, n) enqueues frame n on a slot; slots are reused round-robin after a synchronise of their last stream"""
    for i, s in enumerate(slots):   # warm-up
        submit(s, i)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while n < min_frames or time.perf_counter() - t0 < seconds:
        s = slots[n % len(slots)]
        s["done"].synchronize()
        submit(s, n)
        n += 1
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def run(workload: str = "8k-v210", depth: int = 3, seconds: float = 2.0, min_frames: int = 30, distinct: int = 3, salt: int = 0,
        mode: str = "per-frame-stream", encode: bool = True, copy_only_runs: int = 0, copy_only_seconds: float = 0.7) -> dict:
    """`depth` frames in flight, `distinct` different pinned input frames cycled.  Runs for about `seconds`; returns fps, Mpixel/s and the
    PCIe traffic both ways.
      mode "per-frame-stream": every frame in flight has its own stream carrying H2D, encode, D2H in order (what the modules do);
      mode "split": ONE upload stream, ONE compute stream, ONE download stream, events between the stages of a frame (copy engines never
                    share a stream with the kernel);
      encode=False: the same traffic without the kernel -- what the link alone gives for this in/out byte mix (the ceiling of the leg).
      copy_only_runs = k > 0: the ceiling measured BESIDE the leg it bounds -- k copy-only runs of copy_only_seconds each (same slots, same pinned frames,
                    same streams), the leg cut into k - 1 parts between them: copy, leg, copy, leg, ... copy.  The result carries "copy_only_fps" (the best
                    run: a ceiling is what the link CAN do) and "copy_only_fps_runs"; fps is frames / time over the leg's parts.  A ceiling taken once,
                    seconds after the leg, was beaten by the leg it was meant to bound by 5 % (VERDICT r5 "What's weak" #8)."""
    fmt, pf, oid, w, h = WORKLOADS[workload]
    lib.load()
    srcs = [torch.from_numpy(host_frame(workload, salt + i)).pin_memory() for i in range(distinct)]
    in_len = srcs[0].numel()
    out_len = codec.dxt_size(oid, w, h)
    up, comp, down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    slots = []
    for _ in range(depth):
        st = torch.cuda.Stream()
        slots.append(dict(st=st, dev_in=torch.empty(in_len, dtype=torch.uint8, device="cuda"), dev_out=torch.empty(out_len, dtype=torch.uint8, device="cuda"),
                          host_out=torch.empty(out_len, dtype=torch.uint8).pin_memory(), done=torch.cuda.Event(), e_up=torch.cuda.Event(), e_comp=torch.cuda.Event()))
    for s in slots:
        s["done"].record()

    kernel = [encode]

    def submit(s, i):
        encode = kernel[0]
        if mode == "split":
            with torch.cuda.stream(up):
                s["dev_in"].copy_(srcs[i % distinct], non_blocking=True)
                s["e_up"].record()
            with torch.cuda.stream(comp):
                comp.wait_event(s["e_up"])
                if encode:
                    codec.dxt_encode(pf, oid, s["dev_in"], w, h, dst=s["dev_out"])
                s["e_comp"].record()
            with torch.cuda.stream(down):
                down.wait_event(s["e_comp"])
                s["host_out"].copy_(s["dev_out"], non_blocking=True)
                s["done"].record()
        else:
            with torch.cuda.stream(s["st"]):
                s["dev_in"].copy_(srcs[i % distinct], non_blocking=True)
                if encode:
                    codec.dxt_encode(pf, oid, s["dev_in"], w, h, dst=s["dev_out"])
                s["host_out"].copy_(s["dev_out"], non_blocking=True)
                s["done"].record()

    extra = {}
    if copy_only_runs > 0 and encode:
        parts = max(1, copy_only_runs - 1)
        n, dt, ceil = 0, 0.0, []
        for k in range(copy_only_runs):
            kernel[0] = False
            cn, cdt = _rate_loop(submit, slots, copy_only_seconds, min(min_frames, 10))
            ceil.append(round(cn / cdt, 1))
            kernel[0] = True
            if k < parts:
                pn, pdt = _rate_loop(submit, slots, seconds / parts, -(-min_frames // parts))
                n, dt = n + pn, dt + pdt
        extra = {"copy_only_fps": max(ceil), "copy_only_fps_runs": ceil}
    else:
        n, dt = _rate_loop(submit, slots, seconds, min_frames)
    fps = n / dt
    return {"workload": workload, "fps": round(fps, 1), "mpixels_per_s": round(w * h * fps / 1e6, 1), "frames": n, "seconds": round(dt, 3),
            "in_flight": depth, "mode": mode, "kernel": encode, "pcie_gbs": round((in_len + out_len) * fps / 1e9, 2), "h2d_gbs": round(in_len * fps / 1e9, 2),
            "d2h_gbs": round(out_len * fps / 1e9, 2), "bytes_in_per_frame": in_len, "bytes_out_per_frame": out_len, **extra}


def latency_depth1(workload: str = "8k-v210", frames: int = 24, salt: int = 0, bands=(4, 8)) -> dict:
    """ONE frame in flight, as a display-rate source drives the module: pinned host frame -> H2D -> fused kernel -> D2H -> synchronise, then the
    next frame.  Returns the median wall-clock per frame (submit to synchronised, what the caller waits) and the three stages as HIP events on the
    frame's stream see them, plus what serial and perfectly overlapped copies would take at the rates the stages themselves ran at."""
    fmt, pf, oid, w, h = WORKLOADS[workload]
    lib.load()
    srcs = [torch.from_numpy(host_frame(workload, salt + i)).pin_memory() for i in range(2)]
    in_len, out_len = srcs[0].numel(), codec.dxt_size(oid, w, h)
    dev_in, dev_out = torch.empty(in_len, dtype=torch.uint8, device="cuda"), torch.empty(out_len, dtype=torch.uint8, device="cuda")
    host_out = torch.empty(out_len, dtype=torch.uint8).pin_memory()
    st = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    wall, stages = [], []
    for i in range(frames + 3):
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            ev[0].record()
            dev_in.copy_(srcs[i % 2], non_blocking=True)
            ev[1].record()
            codec.dxt_encode(pf, oid, dev_in, w, h, dst=dev_out)
            ev[2].record()
            host_out.copy_(dev_out, non_blocking=True)
            ev[3].record()
        st.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 3:   # the first frames page things in
            wall.append(dt)
            stages.append([ev[k].elapsed_time(ev[k + 1]) for k in range(3)])
    med = lambda v: float(np.median(v))  # noqa: E731
    h2d, kern, d2h = (med([s[k] for s in stages]) for k in range(3))
    # the same frame as row bands (the module's bands=<k>): upload of band i + 1 | kernel of band i | download of band i - 1, through the C ABI's copy lanes
    banded = {}
    l, dev = lib.load(), torch.cuda.current_device()
    ls, orow = in_len // h, out_len // (h // 4)
    for K in bands:
        edges = [0] + [min(h, (h * (k + 1) // K + 15) // 16 * 16) for k in range(K - 1)] + [h]
        t = []
        for i in range(frames + 3):
            src = srcs[i % 2]
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                for k in range(K):
                    r0, r1 = edges[k], edges[k + 1]
                    if r1 <= r0:
                        continue
                    lib.check(l.ug_hip_upload_ordered_ex(dev, dev_in.data_ptr() + r0 * ls, src.data_ptr() + r0 * ls, (r1 - r0) * ls, 0, st.cuda_stream, lib.COPY_NO_WAIT if k else 0),
                              "ug_hip_upload_ordered_ex")
                    codec.dxt_encode(pf, oid, dev_in[r0 * ls: r1 * ls], w, r1 - r0, dst=dev_out[r0 // 4 * orow: r1 // 4 * orow])
                    lib.check(l.ug_hip_download_ordered_ex(dev, host_out.data_ptr() + r0 // 4 * orow, dev_out.data_ptr() + r0 // 4 * orow, (r1 - r0) // 4 * orow, st.cuda_stream,
                                                           0 if r1 == h else lib.COPY_NO_JOIN), "ug_hip_download_ordered_ex")
            st.synchronize()
            if i >= 3:
                t.append((time.perf_counter() - t0) * 1e3)
        banded[f"bands{K}_ms"] = round(med(t), 3)
    return {"ms": round(med(wall), 3), **banded, "h2d_ms": round(h2d, 3), "kernel_ms": round(kern, 3), "d2h_ms": round(d2h, 3), "frames": frames,
            "bytes_in": in_len, "bytes_out": out_len, "h2d_gbs": round(in_len / h2d / 1e6, 1), "d2h_gbs": round(out_len / d2h / 1e6, 1),
            "over_longer_copy": round(med(wall) / max(h2d, d2h), 3)}


def link_probe(nbytes: int = 64 << 20, seconds: float = 1.0, streams: int = 2) -> dict:
    """What the host link of THIS box gives, pure copies between pinned (first-touched after NUMA binding) and device memory:
    H2D only, D2H only, and both directions at once, `streams` copies in flight per direction."""
    hs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    hd = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    ds = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    dd = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    for h_ in hs:
        h_.fill_(7)
    sup = [torch.cuda.Stream() for _ in range(streams)]
    sdn = [torch.cuda.Stream() for _ in range(streams)]

    def loop(do_up: bool, do_down: bool):
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for k in range(streams):
                if do_up:
                    with torch.cuda.stream(sup[k]):
                        ds[k].copy_(hs[k], non_blocking=True)
                if do_down:
                    with torch.cuda.stream(sdn[k]):
                        hd[k].copy_(dd[k], non_blocking=True)
            for k in range(streams):   # keep one round of copies queued per stream
                sup[k].synchronize()
                sdn[k].synchronize()
            n += streams
        dt = time.perf_counter() - t0
        return n * nbytes / dt / 1e9
    loop(True, True)
    h2d, d2h, both = loop(True, False), loop(False, True), loop(True, True)
    return {"copy_bytes": nbytes, "streams_per_direction": streams, "h2d_gbs": round(h2d, 2), "d2h_gbs": round(d2h, 2),
            "bidir_each_gbs": round(both, 2), "bidir_total_gbs": round(2 * both, 2)}


def run_jpeg_decode(width: int = 3840, height: int = 2160, depth: int = 3, seconds: float = 2.0, out: str = "UYVY", quality: int = 75, restart: int = 4) -> dict:
    """Receiver side, PCIe included: a compressed frame in pinned host memory -> ug_hip_jpeg_decoder_decode (upload + kernels) -> D2H of the
    raw frame into pinned memory, `depth` frames in flight (one decoder object and one stream each, as the C ABI asks).  The stream is made
    once with the repository's encoder from S2 content."""
    import ctypes as C
    lib.load()
    src = torch.from_numpy(synth.s2_video("UYVY", width, height)).cuda()
    enc = codec.JpegEncoder(width, height, quality, restart, subsampling=422)
    data = enc.encode(src)
    enc.close()
    pinned = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
    out_len = codec.linesize(lib.PF_NAMES[out], width) * height
    slots = [dict(st=torch.cuda.Stream(), dec=codec.JpegDecoder(), dev=torch.empty(out_len, dtype=torch.uint8, device="cuda"),
                  host=torch.empty(out_len, dtype=torch.uint8).pin_memory(), busy=False) for _ in range(depth)]
    l = lib.load()

    def submit(s):
        rc = l.ug_hip_jpeg_decoder_decode(s["dec"]._h, C.c_void_p(pinned.data_ptr()), len(data), lib.PF_NAMES[out], s["dev"].data_ptr(), 0, 0, 8, 16, s["st"].cuda_stream)
        lib.check(rc, "ug_hip_jpeg_decoder_decode")
        with torch.cuda.stream(s["st"]):
            s["host"].copy_(s["dev"], non_blocking=True)
        s["busy"] = True

    for s in slots:
        submit(s)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while n < 30 or time.perf_counter() - t0 < seconds:
        s = slots[n % depth]
        if s["busy"]:
            s["st"].synchronize()
        submit(s)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in slots:
        s["dec"].close()
    fps = n / dt
    return {"workload": f"jpeg-decode {width}x{height} 4:2:2 q{quality} restart {restart} -> {out}", "fps": round(fps, 1), "frames": n, "seconds": round(dt, 3),
            "in_flight": depth, "bytes_in_per_frame": len(data), "bytes_out_per_frame": out_len, "pcie_gbs": round((len(data) + out_len) * fps / 1e9, 2)}
