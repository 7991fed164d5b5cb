"""The run-time conventions of the drop-in boundary (VERDICT r4 "next" #1), driven through the reference's OWN compress framework
(video_compress.cpp + messaging.cpp + module.c, compiled from /root/reference into oracle/_ref/ug_runtime_harness*) with its thread roles -- capture
thread pushing, sender thread popping, control thread sending CHANGE_COMPRESS:

  (a) a stream whose format changes while frames are in flight through ONE compress_state with workers=4:batch=4
      (cuda_dxt.cpp:196-204 lazy reconfigure; here once per worker state, with batches queued across the change),
  (b) send_compess_change() mid-stream (video_compress.cpp:154-200: new state first, discard_frames, async_poison(old), delete old),
  (c) teardown with frames still queued in the workers and results un-popped (video_compress.cpp:508-525, rxtx.cpp:133-146): the pill pushed
      behind full queues, the sender draining, then compress_done(); and compress_done() on a state that never saw a pill, with the last result
      (the pill compress_done sends itself) never popped.  compress_done() CONCURRENT with a popping sender is not a convention of the reference:
      `delete proxy` frees the queue under the sender's last compress_pop() (ASan shows it inside the reference's synchronized_queue) -- not tested.

Every delivered frame must be bit-equal to the oracle for ITS OWN desc and configuration, in push order; frames of a replaced configuration may be
discarded (the reference discards them) but only as a suffix; nothing hangs (the harness has a progress watchdog: exit code 4).

CPU half: the test-only "fake" module (ultragrid_amd/module/ug_fake_compress.cpp: the product modules' structure on the product's sharder, the GPU
replaced by a hash) -- plain, under ThreadSanitizer and under AddressSanitizer.  GPU half: the product's `dxt` and `jpeg` modules."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from ultragrid_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
SUPP = os.path.join(ROOT, "ultragrid_amd", "module", "tsan_reference.supp")
SUPP_GPU = os.path.join(ROOT, "ultragrid_amd", "module", "tsan_gpu.supp")
HDR = struct.Struct("<4s7I16s")


def _binary(name):
    return os.path.join(REF, name)


def _needs(name):
    return pytest.mark.skipif(not os.path.exists(_binary(name)), reason=f"oracle/_ref/{name} not built (needs /root/reference)")


def _records(path):
    data = open(path, "rb").read()
    out, pos = [], 0
    while pos < len(data):
        magic, index, seq, w, h, il, tiles, n, codec = HDR.unpack_from(data, pos)
        assert magic == b"UGRF", (pos, magic)
        pos += HDR.size
        out.append(dict(index=index, seq=seq, w=w, h=h, il=il, tiles=tiles, codec=codec.rstrip(b"\0").decode(), data=data[pos:pos + n]))
        pos += n
    return out


def _run(binary, tmp_path, script, env=None, timeout=300):
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"oracle/_ref/{binary} not built")
    sp, out = tmp_path / "script.txt", tmp_path / "out.rec"
    sp.write_text(script)
    e = dict(os.environ, UG_RT_WATCHDOG_S="45")
    e.update(env or {})
    cmd = [_binary(binary), str(sp), str(out)]
    if binary.endswith("_tsan") and shutil.which("setarch"):
        # libtsan of gcc 11 cannot place its shadow under the 32 bits of mmap randomisation newer kernels use ("unexpected memory mapping"):
        # run without address-space randomisation, the documented way out
        cmd = ["setarch", "x86_64", "-R"] + cmd
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
    text = r.stdout + r.stderr
    if "ThreadSanitizer: unexpected memory mapping" in text:
        pytest.skip("this kernel's address-space layout cannot host libtsan, even without randomisation")
    if "ReserveShadowMemoryRange failed" in text or "Shadow memory range interleaves" in text:
        pytest.skip("this box's address-space layout cannot host the ASan shadow beside the HIP runtime's reservations")
    assert "WATCHDOG" not in text and r.returncode != 4, "hang:\n" + text[-3000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-6000:]
    assert r.returncode == 0, text[-3000:]
    return _records(out), r.stdout


SAN_ENV = {
    "ug_runtime_harness_fake": {},
    "ug_runtime_harness_fake_tsan": {"TSAN_OPTIONS": f"halt_on_error=1 exitcode=66 second_deadlock_stack=1 suppressions={SUPP}"},
    "ug_runtime_harness_fake_asan": {"ASAN_OPTIONS": "detect_leaks=1 exitcode=66", "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1"},
}

# ------------------------------------------------------------------------------------------------------------------------------------------
# CPU half: the fake module
# ------------------------------------------------------------------------------------------------------------------------------------------
CODEC_IDS = {}


def _fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


class FakeSet:
    def __init__(self, tmp_path, name, codec, w, h, il, n, bpp_num, bpp_den, salt):
        self.name, self.codec, self.w, self.h, self.il, self.n = name, codec, w, h, il, n
        self.frame_len = w * bpp_num // bpp_den * h
        rng = np.random.default_rng(salt)
        self.frames = [rng.integers(0, 256, self.frame_len, dtype=np.uint8).tobytes() for _ in range(n)]
        self.hashes = [_fnv1a(f) for f in self.frames]
        self.path = tmp_path / f"{name}.raw"
        self.path.write_bytes(b"".join(self.frames))

    def line(self):
        return f"frames {self.name} {self.codec} {self.w} {self.h} {'i' if self.il else 'p'} {self.path} {self.n}\n"


def _fake_sets(tmp_path):
    return {"A": FakeSet(tmp_path, "A", "UYVY", 64, 16, False, 5, 2, 1, 1), "B": FakeSet(tmp_path, "B", "RGB", 32, 8, True, 3, 3, 1, 2),
            "C": FakeSet(tmp_path, "C", "RGBA", 128, 4, False, 4, 4, 1, 3)}


def _check_fake(records, pushes, sets):
    """pushes: per push index (set name, allowed tags, generation).  Checks content, configuration, order; returns delivered indices per generation."""
    last = -1
    per_gen = {}
    for r in records:
        assert r["index"] > last, f"out of order: {r['index']} after {last}"
        last = r["index"]
        name, tags, gen = pushes[r["index"]]
        s = sets[name]
        f = struct.unpack("<10IQ8I", r["data"])
        assert f[0] == 0x454B4146 and len(r["data"]) == 80
        tag, cfg, own, h, batch_n, in_len = f[1], f[2:6], f[6:10], f[10], f[11], f[13]
        assert cfg == own, f"frame {r['index']} ({own}) was encoded under the configuration of another format ({cfg})"
        assert (own[0], own[1]) == (s.w, s.h) == (r["w"], r["h"]) and own[3] == (3 if s.il else 0), (r["index"], own)
        assert in_len == s.frame_len and h == s.hashes[r["index"] % s.n], f"frame {r['index']}: payload of another frame"
        assert tag in tags, f"frame {r['index']} encoded by configuration {tag}, expected one of {tags}"
        assert 1 <= batch_n <= 16
        per_gen.setdefault((gen, tag), []).append(r["index"])
    return per_gen


def _script(sets, body, head=""):
    return head + "".join(s.line() for s in sets.values()) + body


FAKE_BINARIES = ["ug_runtime_harness_fake", "ug_runtime_harness_fake_tsan", "ug_runtime_harness_fake_asan"]


@pytest.mark.parametrize("binary", FAKE_BINARIES)
@pytest.mark.parametrize("cfg", ["fake:tag=1:workers=4:batch=4:delay_us=1500", "fake:tag=1:workers=1:delay_us=300", "fake:tag=1:dev=0,1,2:workers=2:batch=8:delay_us=800"])
def test_format_changes_with_frames_in_flight(tmp_path, binary, cfg):
    """(a): blocks of one format, then single frames alternating between three formats (every worker state reconfigures again and again, batches
    are cut at every change) -- every frame delivered, in order, encoded under the configuration of its own format."""
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"{binary} not built")
    sets = _fake_sets(tmp_path)
    plan = [("A", 40), ("B", 40), ("C", 40), ("A", 40)] + [("ABC"[i % 3], 1 + i % 2) for i in range(40)]
    body = f"init {cfg}\n" + "".join(f"push {n} {c}\n" for n, c in plan) + "pill\ndone\n"
    pushes = [(n, {1}, 0) for n, c in plan for _ in range(c)]
    records, out = _run(binary, tmp_path, _script(sets, body), SAN_ENV[binary])
    _check_fake(records, pushes, sets)
    assert [r["index"] for r in records] == list(range(len(pushes)))
    assert "BAD_TIMES" not in out and "FAKE live_states=0" in out


@pytest.mark.parametrize("binary", FAKE_BINARIES)
@pytest.mark.parametrize("holds", [0, 1])
def test_change_compress_mid_stream(tmp_path, binary, holds):
    """(b): send_compess_change() between pushes.  Frames before the change carry the old configuration and may be cut off (a suffix of them is
    discarded, video_compress.cpp:191-193), frames after it carry the new one and all arrive; states of the old module are all destroyed.
    holds=1: a sender that keeps the previous frame while it pops the next -- with the reference's by-value video_frame_pool the capture thread
    would wait in done() for that frame for ever (video_frame_pool.cpp:150-155); the modules' frames keep their pool alive instead."""
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"{binary} not built")
    sets = _fake_sets(tmp_path)
    gens = [("fake:tag=1:workers=4:batch=4:delay_us=1500", [("A", 30), ("B", 7)]), ("fake:tag=2:workers=2:delay_us=500", [("B", 30)]),
            ("fake:tag=3:workers=4:batch=4:delay_us=1000", [("C", 25), ("A", 25)]), ("fake:tag=4:workers=1", [("A", 20)])]
    body, pushes = f"sender_holds {holds}\npop_delay_us 300\n", []
    for g, (cfg, plan) in enumerate(gens):
        body += (f"init {cfg}\n" if g == 0 else f"msg {cfg}\n") + "".join(f"push {n} {c}\n" for n, c in plan)
        pushes += [(n, {g + 1}, g) for n, c in plan for _ in range(c)]
    body += "done\n"
    records, out = _run(binary, tmp_path, _script(sets, body), SAN_ENV[binary])
    per_gen = _check_fake(records, pushes, sets)
    start = 0
    for g, (cfg, plan) in enumerate(gens):
        n = sum(c for _, c in plan)
        got = per_gen.get((g, g + 1), [])
        assert got == list(range(start, start + len(got))), f"generation {g}: delivered frames are not a prefix of what was pushed: {got}"
        if g == len(gens) - 1:
            assert len(got) == n, "frames of the configuration that was active at compress_done() are all delivered"
        start += n
    assert "FAKE live_states=0" in out, out


@pytest.mark.parametrize("binary", FAKE_BINARIES)
def test_change_compress_from_a_control_thread(tmp_path, binary):
    """(b) with the message arriving whenever: three control threads fire while the capture thread pushes at a steady pace.  The configuration
    a frame was encoded with never goes backwards, and every delivered frame is intact."""
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"{binary} not built")
    sets = _fake_sets(tmp_path)
    body = ("init fake:tag=1:workers=4:batch=4:delay_us=1000\npace_us 700\n"
            "msg_ctl 20 fake:tag=2:workers=2:batch=2:delay_us=500\nmsg_ctl 60 fake:tag=3:workers=4:delay_us=1500\nmsg_ctl 110 fake:tag=4:workers=3:batch=4\n"
            "push A 60\npush B 60\npush C 60\nsleep_ms 150\npush A 30\ndone\n")
    pushes = [(n, {1, 2, 3, 4}, 0) for n, c in (("A", 60), ("B", 60), ("C", 60), ("A", 30)) for _ in range(c)]
    records, out = _run(binary, tmp_path, _script(sets, body), SAN_ENV[binary])
    _check_fake(records, pushes, sets)
    tags = [struct.unpack_from("<I", r["data"], 4)[0] for r in records]
    assert tags == sorted(tags), "a frame of an older configuration was delivered after one of a newer"
    assert tags[-1] == 4 and [r["index"] for r in records][-30:] == list(range(180, 210))  # the last 30 were pushed long after the last change
    assert "FAKE live_states=0" in out


@pytest.mark.parametrize("binary", FAKE_BINARIES)
@pytest.mark.parametrize("cfg", ["fake:tag=7:workers=4:batch=4:delay_us=3000", "fake:tag=7:workers=2:delay_us=1000:fail_every=5"])
def test_teardown_with_frames_in_flight(tmp_path, binary, cfg):
    """(c): teardown straight after the last push -- the workers' queues full, a slow sender, most results un-popped: the pill goes in behind all
    that, everything that was pushed drains through the sender in order, then compress_done() deletes the module.  fail_every: frames the encoder
    dropped are skipped, the rest still in order (gpujpeg.cpp:695-713)."""
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"{binary} not built")
    sets = _fake_sets(tmp_path)
    body = f"init {cfg}\npop_delay_us 1500\npush A 30\npush C 30\ndone\n"
    pushes = [(n, {7}, 0) for n in ("A", "C") for _ in range(30)]
    records, out = _run(binary, tmp_path, _script(sets, body), SAN_ENV[binary])
    _check_fake(records, pushes, sets)
    if "fail_every" in cfg:
        assert 40 <= len(records) < 60
    else:
        assert [r["index"] for r in records] == list(range(60))
    assert "DONE_CALLED pushed=60" in out and "pill=0" in out and "FAKE live_states=0" in out
    popped_at_done = int(out.split("DONE_CALLED pushed=60 popped=")[1].split()[0])
    assert popped_at_done < 55, "the scenario needs results still un-popped when the teardown starts"


@pytest.mark.parametrize("binary", FAKE_BINARIES)
@pytest.mark.parametrize("frames", [0, 9])
def test_compress_done_without_a_pill_and_with_the_last_result_unpopped(tmp_path, binary, frames):
    """(c), the other half: no sender thread (rxtx.cpp:136-141), the capture thread pops what it pushed, then compress_done() on a state that never saw
    a pill: compress_done sends it (video_compress.cpp:516-518), the consumer thread leaves it in the queue, nobody pops it, the module is deleted."""
    if not os.path.exists(_binary(binary)):
        pytest.skip(f"{binary} not built")
    sets = _fake_sets(tmp_path)
    body = "init_nosender fake:tag=7:workers=4:batch=4:delay_us=500\n" + "".join(f"push A 1\npop 1\n" for _ in range(frames)) + "done\n"
    records, out = _run(binary, tmp_path, _script(sets, body), SAN_ENV[binary])
    _check_fake(records, [("A", {7}, 0)] * frames, sets)
    assert len(records) == frames and "pill=0" in out and "FAKE live_states=0" in out


# ------------------------------------------------------------------------------------------------------------------------------------------
# GPU half: the product's modules
# ------------------------------------------------------------------------------------------------------------------------------------------
class RealSet:
    def __init__(self, tmp_path, name, codec, w, h, il, n, salt, kind="s2"):
        self.name, self.codec, self.w, self.h, self.il, self.n = name, codec, w, h, il, n
        gen = synth.s2_video if kind == "s2" else synth.s1_random
        base = {"YUYV": "UYVY"}.get(codec, codec)
        self.frames = [np.ascontiguousarray(gen(base, w, h, salt=salt + i)) for i in range(n)]
        self.path = tmp_path / f"{name}.raw"
        np.concatenate([f.ravel() for f in self.frames]).tofile(self.path)
        self._want = {}

    def line(self):
        return f"frames {self.name} {self.codec} {self.w} {self.h} {'i' if self.il else 'p'} {self.path} {self.n}\n"

    def want(self, po, cfg, k):
        """the oracle's bytes for frame k of this set under module configuration cfg ("dxt:DXT5" | "dxt:DXT1" | "jpeg:q=<q>:restart=<r>")"""
        key = (cfg.split(":workers")[0].split(":batch")[0], k)
        if key not in self._want:
            self._want[key] = _oracle_bytes(po, key[0], self.codec, self.il, self.frames[k], self.w, self.h)
        return self._want[key]


def _oracle_bytes(po, cfg, codec, il, src, w, h):
    target = "UYVY" if codec in ("UYVY", "YUYV", "v210") else "RGB"
    decoded = src.ravel() if codec == target else po.convert_frame(codec, target, src, w, h)   # the reference's line decoder (cuda_dxt.cpp:206-220)
    if cfg.startswith("dxt"):
        if il:
            decoded = po.deinterlace_blend(decoded, {"UYVY": 2 * w, "RGB": 3 * w}[target], h)  # RTDXT's vc_deinterlace (dxt_glsl.cpp:291-293)
        if "DXT1_YUV" in cfg:                                # DXT1 over the Y,Cb,Cr samples of the UYVY form (dxt_glsl.cpp:104-110)
            assert target == "UYVY"
            return po.dxt_encode(po.IN_UYVY_RAW, po.OUT_DXT1, decoded, w, h).tobytes()
        oid = po.OUT_DXT5YCOCG if "DXT5" in cfg else po.OUT_DXT1
        return po.dxt_encode(po.IN_UYVY if target == "UYVY" else po.IN_RGB, oid, decoded, w, h).tobytes()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from jpeg_bitstream import write_jpeg
    opts = dict(kv.split("=") for kv in cfg.split(":")[1:] if "=" in kv)
    q, restart = int(opts["q"]), int(opts["restart"])
    ql, qc = po.jpeg_qtable(q, 0), po.jpeg_qtable(q, 1)
    if target == "RGB":
        comp = decoded.reshape(h, w, 3)
        coefs = [po.jpeg_fdct_quant_plane(np.ascontiguousarray(comp[..., c]), po.jpeg_divisors(ql), (w + 7) // 8, (h + 7) // 8) for c in range(3)]
        if "interleaved" in cfg:
            return write_jpeg(w, h, ql, qc, *coefs, restart=restart, sub=444)
        from jpeg_bitstream import write_jpeg_noninterleaved
        return write_jpeg_noninterleaved(w, h, ql, coefs, restart=restart)       # RGB input: one scan per component (gpujpeg.cpp:303)
    y, u, v = po.uyvy_to_i422(decoded, w, h)
    mw, mh = (w + 15) // 16, (h + 7) // 8
    return write_jpeg(w, h, ql, qc, po.jpeg_fdct_quant_plane(y, po.jpeg_divisors(ql), 2 * mw, mh), po.jpeg_fdct_quant_plane(u, po.jpeg_divisors(qc), mw, mh),
                      po.jpeg_fdct_quant_plane(v, po.jpeg_divisors(qc), mw, mh), restart=restart, sub=422)


def _check_real(po, records, pushes, sets):
    """pushes: per push index (set name, candidate configurations in the order they were active).  Returns the configuration index of every record."""
    last, used = -1, []
    for r in records:
        assert r["index"] > last, f"out of order: {r['index']} after {last}"
        last = r["index"]
        name, cfgs = pushes[r["index"]]
        s = sets[name]
        assert (r["w"], r["h"], r["tiles"]) == (s.w, s.h, 1), (r["index"], r["w"], r["h"])
        hit = None
        for ci, cfg in cfgs:
            name_out = "JPEG" if cfg.startswith("jpeg") else ("DXT5" if "DXT5" in cfg else "DXT1_YUV" if "DXT1_YUV" in cfg else "DXT1")
            if r["codec"] != name_out:
                continue
            want = s.want(po, cfg, r["index"] % s.n)
            if r["data"] == want:
                hit = ci
                break
        assert hit is not None, f"frame {r['index']} ({s.codec} {s.w}x{s.h}, delivered as {r['codec']}, {len(r['data'])} B) equals the oracle under none of {[c for _, c in cfgs]}"
        if r["codec"].startswith("DXT"):
            assert r["il"] == 0                              # blended -> announced as progressive (dxt_glsl.cpp:196-198)
        used.append(hit)
    return used


REAL = "ug_runtime_harness"


def _real_binaries():
    """the plain harness, and the ASan+UBSan and TSan builds of the same harness + the product's two compress modules (libug_mi355x.so and the HIP runtime
    under them are not instrumented; profiles/r05_runtime_conventions.txt: both run clean beside the HIP runtime on the MI355X box).  UG_RT_SANITIZED_GPU=0
    leaves the sanitized builds out; a box whose address-space layout cannot host a sanitizer runtime skips them (see _run)."""
    return [REAL] + ([] if os.environ.get("UG_RT_SANITIZED_GPU") == "0" else ["ug_runtime_harness_asan", "ug_runtime_harness_tsan"])


REAL_SAN_ENV = {
    REAL: {},
    "ug_runtime_harness_asan": {"ASAN_OPTIONS": "detect_leaks=0 exitcode=66 protect_shadow_gap=0", "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1"},
    "ug_runtime_harness_tsan": {"TSAN_OPTIONS": f"halt_on_error=1 exitcode=66 suppressions={SUPP_GPU}"},
}


@_needs(REAL)
@pytest.mark.gpu
@pytest.mark.parametrize("binary", _real_binaries())
@pytest.mark.parametrize("cfg", ["dxt:DXT5:workers=4:batch=4", "dxt:DXT1:workers=2"])
def test_dxt_module_format_changes_in_flight(tmp_path, po, binary, cfg):
    """40 x 1080p UYVY, 40 x 4K v210, 40 x 720p RGB interlaced (de-interlaced on the device), back to 1080p UYVY, then
    single frames alternating between small formats -- one compress_state, frames in flight across every change, sender on its own thread."""
    sets = {"A": RealSet(tmp_path, "A", "UYVY", 1920, 1080, False, 3, 10), "B": RealSet(tmp_path, "B", "v210", 3840, 2160, False, 2, 20),
            "C": RealSet(tmp_path, "C", "RGB", 1280, 720, True, 3, 30), "S": RealSet(tmp_path, "S", "UYVY", 192, 64, False, 4, 40, "s1"),
            "T": RealSet(tmp_path, "T", "RGBA", 200, 36, True, 3, 50, "s1"), "U": RealSet(tmp_path, "U", "v210", 96, 32, False, 3, 60, "s1")}
    plan = [("A", 40), ("B", 40), ("C", 40), ("A", 40)] + [("STU"[i % 3], 1 + i % 2) for i in range(30)]
    body = f"init {cfg}\n" + "".join(f"push {n} {c}\n" for n, c in plan) + "pill\ndone\n"
    pushes = [(n, [(0, cfg)]) for n, c in plan for _ in range(c)]
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=600)
    _check_real(po, records, pushes, sets)
    assert [r["index"] for r in records] == list(range(len(pushes))) and "BAD_TIMES" not in out


@_needs(REAL)
@pytest.mark.gpu
@pytest.mark.parametrize("binary", _real_binaries())
def test_jpeg_module_format_changes_in_flight(tmp_path, po, binary):
    cfg = "jpeg:q=75:restart=4:batch=4"
    sets = {"A": RealSet(tmp_path, "A", "UYVY", 640, 360, False, 3, 11), "B": RealSet(tmp_path, "B", "v210", 960, 544, False, 2, 21),
            "C": RealSet(tmp_path, "C", "RGB", 320, 240, False, 3, 31), "S": RealSet(tmp_path, "S", "UYVY", 192, 64, False, 4, 41),
            "T": RealSet(tmp_path, "T", "RGB", 200, 36, False, 3, 51)}
    plan = [("A", 40), ("B", 40), ("C", 40), ("A", 40)] + [("ST"[i % 2], 1 + i % 3) for i in range(30)]
    body = f"init {cfg}\n" + "".join(f"push {n} {c}\n" for n, c in plan) + "pill\ndone\n"
    pushes = [(n, [(0, cfg)]) for n, c in plan for _ in range(c)]
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=600)
    _check_real(po, records, pushes, sets)
    assert [r["index"] for r in records] == list(range(len(pushes))) and "BAD_TIMES" not in out


@_needs(REAL)
@pytest.mark.gpu
@pytest.mark.parametrize("binary", _real_binaries())
@pytest.mark.parametrize("holds", [0, 1])
def test_change_compress_between_the_product_modules(tmp_path, po, binary, holds):
    """dxt:DXT5 -> dxt:DXT1 -> jpeg -> dxt:DXT5 through send_compess_change() while frames are in flight: the new module is
    created while the old one still encodes on the same GPU (the copy lanes of ug_runtime.hip are process-wide), the old one is poisoned, drained
    and deleted on the capture thread.  Every delivered frame equals the oracle under the configuration that was active when it was pushed."""
    sets = {"A": RealSet(tmp_path, "A", "UYVY", 1920, 1080, False, 3, 12), "S": RealSet(tmp_path, "S", "UYVY", 320, 192, False, 4, 42),
            "C": RealSet(tmp_path, "C", "RGB", 1280, 720, True, 2, 32)}
    gens = [("dxt:DXT5:workers=4:batch=4", [("A", 30), ("C", 10)]), ("dxt:DXT1:workers=2", [("C", 20), ("A", 20)]),
            ("jpeg:q=50:restart=4:batch=4", [("S", 40)]), ("dxt:DXT5:workers=3:batch=2", [("S", 10), ("A", 30)])]
    body, pushes = f"sender_holds {holds}\npop_delay_us 200\n", []
    for g, (cfg, plan) in enumerate(gens):
        body += (f"init {cfg}\n" if g == 0 else f"msg {cfg}\n") + "".join(f"push {n} {c}\n" for n, c in plan)
        pushes += [(n, [(g, cfg)]) for n, c in plan for _ in range(c)]
    body += "done\n"
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=600)
    used = _check_real(po, records, pushes, sets)
    idx = [r["index"] for r in records]
    start = 0
    for g, (cfg, plan) in enumerate(gens):
        n = sum(c for _, c in plan)
        got = [i for i, u in zip(idx, used) if u == g]
        assert got == list(range(start, start + len(got))), f"generation {g}: {got}"
        if g == len(gens) - 1:
            assert len(got) == n
        start += n


@_needs(REAL)
@pytest.mark.gpu
@pytest.mark.parametrize("binary", _real_binaries())
def test_change_compress_from_a_control_thread_on_the_gpu(tmp_path, po, binary):
    sets = {"S": RealSet(tmp_path, "S", "UYVY", 320, 192, False, 4, 43), "A": RealSet(tmp_path, "A", "UYVY", 1920, 1080, False, 2, 13)}
    cfgs = ["dxt:DXT5:workers=4:batch=4", "dxt:DXT1:workers=2:batch=2", "jpeg:q=50:restart=4:batch=4", "dxt:DXT1_YUV:workers=3"]   # (told apart by their bytes)
    body = (f"init {cfgs[0]}\npace_us 500\nmsg_ctl 30 {cfgs[1]}\nmsg_ctl 90 {cfgs[2]}\nmsg_ctl 160 {cfgs[3]}\n"
            "push S 100\npush A 100\npush S 150\nsleep_ms 200\npush A 20\ndone\n")
    pushes = [(n, list(enumerate(cfgs))) for n, c in (("S", 100), ("A", 100), ("S", 150), ("A", 20)) for _ in range(c)]
    # (1080p through the JPEG test writer in pure Python is slow: JPEG may only meet the small set if the timing holds; either way the oracle decides)
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=900)
    used = _check_real(po, records, pushes, sets)
    assert used == sorted(used) and used[-1] == 3
    assert [r["index"] for r in records][-20:] == list(range(350, 370))


@_needs(REAL)
@pytest.mark.gpu
@pytest.mark.parametrize("binary", _real_binaries())
@pytest.mark.parametrize("cfg", ["dxt:DXT5:workers=4:batch=4", "jpeg:q=75:restart=4:workers=2:batch=4"])
def test_teardown_with_frames_in_the_workers(tmp_path, po, binary, cfg):
    """teardown straight after the last push -- queues full, sender slow, results un-popped; then a state that is deleted
    without ever having seen a pill (no sender: the capture thread pops for itself)."""
    sets = {"A": RealSet(tmp_path, "A", "UYVY", 640, 360, False, 3, 14), "C": RealSet(tmp_path, "C", "RGB", 320, 240, False, 2, 34)}
    body = f"init {cfg}\npop_delay_us 1500\npush A 30\npush C 30\ndone\n"
    pushes = [(n, [(0, cfg)]) for n in ("A", "C") for _ in range(30)]
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=600)
    _check_real(po, records, pushes, sets)
    assert [r["index"] for r in records] == list(range(60))
    assert int(out.split("DONE_CALLED pushed=60 popped=")[1].split()[0]) < 55
    body = f"init_nosender {cfg}\n" + "push A 1\npop 1\npush C 1\npop 1\n" * 4 + "done\n"
    records, out = _run(binary, tmp_path, _script(sets, body), REAL_SAN_ENV[binary], timeout=600)
    _check_real(po, records, [(n, [(0, cfg)]) for _ in range(4) for n in ("A", "C")], sets)
    assert len(records) == 8 and "pill=0" in out


def test_random_scripts_on_the_fake_module():
    """tools/fuzz_runtime_conventions.py: random sequences of format changes, CHANGE_COMPRESS from the capture side and from control threads, paces, slow and
    holding senders, encoder failures, teardown with and without a pill -- 45 of them here (plain / TSan / ASan in turn); 292 more were run for
    profiles/r05_runtime_conventions.txt"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_runtime_conventions.py"), "45", "20260925"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "every check held" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
