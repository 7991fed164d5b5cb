"""src/from_planar.h and src/to_planar.h whole (SURVEY.md 8(f) N3 building blocks), called by the reference's function names.
CPU: the numpy restatement (oracle/planar_oracle.py) against the compiled reference (oracle/_ref).  GPU: ug_hip_from_planar /
ug_hip_to_planar against the restatement -- aligned fast paths, ragged widths, odd pitches and misaligned pointers."""
import numpy as np
import pytest

from oracle import planar_oracle as PO

SIZES = [(64, 8), (8, 2), (70, 5), (17, 3), (1, 1), (33, 4), (128, 6)]


def from_case(name, w, h, seed, dirty=False, in_depth=0):
    """(planes, in_depth, kwargs) for a decode_planar_func_t"""
    rng = np.random.default_rng(seed)
    fam, out, depth, idx = PO.from_info(name, in_depth)
    cw = (w + 1) // 2
    if fam == "rgbp":
        n = 4 if (idx[3] >= 0) else 3
        if depth == 8:
            return [rng.integers(0, 256, (h, w)).astype(np.uint8) for _ in range(n)], in_depth
        hi = 65536 if dirty else 1 << depth
        return [rng.integers(0, hi, (h, w)).astype(np.uint16) for _ in range(n)], in_depth
    if name == "yuv444p_to_vuya":
        return [rng.integers(0, 256, (h, w)).astype(np.uint8) for _ in range(3)], 0
    if name in ("yuv420p_to_uyvy", "yuv420_to_i420"):
        ch = (h + 1) // 2
        return [rng.integers(0, 256, (h, w)).astype(np.uint8)] + [rng.integers(0, 256, (ch, cw)).astype(np.uint8) for _ in range(2)], 0
    if depth == 8:
        return [rng.integers(0, 256, (h, w)).astype(np.uint8)] + [rng.integers(0, 256, (h, cw)).astype(np.uint8) for _ in range(2)], in_depth
    hi = 65536 if dirty and name != "yuv422p10le_to_v210" else 1 << depth
    return [rng.integers(0, hi, (h, w)).astype(np.uint16)] + [rng.integers(0, hi, (h, cw)).astype(np.uint16) for _ in range(2)], in_depth


def from_variants():
    """(name, in_depth) pairs: the XX conversions at every depth the reference routes to them"""
    out = []
    for name in PO.FROM_NAMES:
        if name == "rgbpXX_to_rgb":
            out += [(name, d) for d in (8, 10, 12, 16)]
        elif name in ("rgbpXXle_to_rg48", "rgbpXXle_to_r10k"):
            out += [(name, d) for d in (10, 12, 16)]
        elif name == "rgbpXXle_to_r12l":
            out += [(name, d) for d in (12, 16)]
        elif name == "yuv422pXX_to_uyvy":
            out += [(name, d) for d in (8, 10, 12, 16)]
        else:
            out.append((name, 0))
    return out


def size_ok(name, w, h):
    if name == "yuv420_to_i420":
        return w % 2 == 0 and h % 2 == 0
    return True


def r12l_valid_equal(a, b, w, h):
    """R12L lines compared on the pixels inside the picture (the reference packs stack garbage behind a ragged line end)"""
    da = PO.to_planar("r12l_to_rgbp12le", a.ravel(), w, h)
    db = PO.to_planar("r12l_to_rgbp12le", b.ravel(), w, h)
    return all(np.array_equal(x, y) for x, y in zip(da, db))


@pytest.mark.parametrize("name,in_depth", from_variants())
def test_from_planar_restatement_vs_reference(po, name, in_depth):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    for i, (w, h) in enumerate(SIZES):
        if not size_ok(name, w, h):
            continue
        for dirty in (False, True):
            planes, dep = from_case(name, w, h, 100 * i + dirty, dirty, in_depth)
            shifts = [(0, 8, 16), (16, 8, 0), (8, 16, 0)][i % 3]
            got = PO.from_planar(name, planes, w, h, dep, shifts)
            want = PO.ref_from_planar(name, planes, w, h, dep, shifts, scalar=(name == "yuv420p_to_uyvy" and w < 16))
            if name.endswith("_r12l") and w % 8:
                assert r12l_valid_equal(got, want, w, h), (name, w, h, dirty)
            else:
                assert np.array_equal(got, want), (name, w, h, dirty, in_depth)


def to_case(name, w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, PO.in_linesize(name, w) * h + 64).astype(np.uint8)


@pytest.mark.parametrize("name", PO.TO_NAMES)
def test_to_planar_restatement_vs_reference(po, name):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    for i, (w, h) in enumerate(SIZES):
        if name in ("uyvy_to_nv12", "uyvy_to_i420") and (w % 2 or h % 2):
            continue  # these two have their own ragged-size tests (tests/test_pixfmt*.py, tests/test_planar.py)
        if name == "v210_to_p010le" and w % 6 and h < 5:
            continue  # the reference reads in front of its planes (to_planar.c:142-148 with y < 4)
        src = to_case(name, w, h, i)
        got = PO.to_planar(name, src, w, h)
        want = PO.ref_to_planar(name, src, w, h)
        assert len(got) == len(want)
        for k, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape and np.array_equal(a, b), (name, w, h, k)


def test_r12l_round_trip(po):
    """gbrp12le_to_r12l then r12l_to_gbrp12le is the identity on 12-bit samples (size-independent property)"""
    rng = np.random.default_rng(5)
    w, h = 1928, 4
    planes = [rng.integers(0, 4096, (h, w)).astype(np.uint16) for _ in range(3)]
    packed = PO.from_planar("gbrp12le_to_r12l", planes, w, h)
    back = PO.to_planar("r12l_to_gbrp12le", packed.ravel(), w, h)
    assert all(np.array_equal(a, b) for a, b in zip(planes, back))


# ---------------------------------------------------------------------------------------------------------------------------------
GPU_SIZES = [(1920, 16), (64, 8), (70, 5), (17, 3), (1, 1), (136, 7)]


def _dev_plane(torch, arr, misalign):
    """device copy of a 2-D array; misalign: odd row stride and a pointer one sample past an aligned address"""
    t = torch.from_numpy(arr.view(np.int16) if arr.dtype == np.uint16 else arr)
    if not misalign:
        return t.cuda()
    h, w = arr.shape
    buf = torch.zeros((h, w + 3), dtype=t.dtype, device="cuda")
    buf[:, 1:1 + w] = t.cuda()
    return buf[:, 1:1 + w]


@pytest.mark.gpu
@pytest.mark.parametrize("name,in_depth", from_variants())
def test_gpu_from_planar(hip, name, in_depth):
    import torch
    codec = hip
    assert hip.L.load().ug_hip_from_planar_supported(name.encode()) == 1
    for i, (w, h) in enumerate(GPU_SIZES):
        if not size_ok(name, w, h):
            continue
        for misalign in (False, True):
            planes, dep = from_case(name, w, h, 7 * i + misalign, dirty=misalign, in_depth=in_depth)
            shifts = [(0, 8, 16), (16, 8, 0), (8, 16, 0)][i % 3]
            want = PO.from_planar(name, planes, w, h, dep, shifts)
            if name == "yuv420_to_i420":
                out = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
                pitch = w
            else:
                pitch = want.shape[1] + (4 if misalign and name != "yuv422p10le_to_v210" else 0)
                if misalign and name in ("yuv420p_to_uyvy", "yuv422p10le_to_v210"):
                    pitch = want.shape[1]
                out = torch.full((h, pitch), 0xA5, dtype=torch.uint8, device="cuda")
            dplanes = [_dev_plane(torch, p, misalign and name not in ("gbrap_to_rgb", "gbrap_to_rgba", "yuv420p_to_uyvy", "yuv422p10le_to_v210")) for p in planes]
            if name.startswith("gbrap") or (name == "rgbpXX_to_rgb" and dep == 8):
                # these stride every plane by in_linesize[0]: keep the strides equal
                dplanes = [_dev_plane(torch, p, misalign) for p in planes]
            codec.from_planar(name, dplanes, w, h, out, pitch, dep, shifts)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            if name == "yuv420_to_i420":
                assert np.array_equal(got, want), (name, w, h)
                continue
            wb = want.shape[1]
            if name.endswith("_r12l") and w % 8:
                assert np.array_equal(got[:, :wb], want), (name, w, h, misalign)  # zero-filled tail on both sides
            else:
                nbytes = PO.out_linesize(name, w)
                if name.endswith("_v210"):
                    nbytes = 16 * (w // 6)     # width / 6 groups are written (from_planar.c:309)
                elif name.startswith("yuv422p"):
                    nbytes = 4 * (w // 2)      # width / 2 pairs (from_planar.c:400)
                assert np.array_equal(got[:, :nbytes], want[:, :nbytes]), (name, w, h, misalign, in_depth)
                if name != "yuv420p_to_uyvy":
                    assert (got[:, nbytes:] == 0xA5).all(), (name, w, h, "wrote past the converted pixels")


@pytest.mark.gpu
@pytest.mark.parametrize("name", PO.TO_NAMES)
def test_gpu_to_planar(hip, name):
    import torch
    codec = hip
    assert hip.L.load().ug_hip_to_planar_supported(name.encode()) == 1
    for i, (w, h) in enumerate(GPU_SIZES):
        if name in ("uyvy_to_nv12", "uyvy_to_i420") and (w % 2 or h % 2):
            continue
        if name == "v210_to_p010le" and w % 6 and h < 5:
            continue  # refused: the reference reads in front of its planes there (to_planar.c:142-148 with y < 4)
        for misalign in (False, True):
            src = to_case(name, w, h, 3 * i + misalign)
            want = PO.to_planar(name, src, w, h)
            dsrc = torch.from_numpy(src).cuda()
            planes = []
            for (rows, n, dt) in PO.to_shapes(name, w, h):
                tdt = torch.int16 if dt == np.uint16 else torch.uint8
                pad = 3 if (misalign and name not in ("uyvy_to_nv12", "uyvy_to_i420", "v210_to_p010le")) else 0
                buf = torch.full((max(rows, 1), n + pad), 0x5A, dtype=tdt, device="cuda")
                planes.append(buf[:rows, (1 if pad else 0):(1 if pad else 0) + n])
            codec.to_planar(name, dsrc, w, h, planes)
            torch.cuda.synchronize()
            for k, (p, wnt) in enumerate(zip(planes, want)):
                got = p.cpu().numpy()
                got = got.view(np.uint16) if wnt.dtype == np.uint16 else got
                assert np.array_equal(got, wnt), (name, w, h, k, misalign)


@pytest.mark.gpu
def test_gpu_planar_api_rejects_unknown_names_and_bad_depths(hip):
    import ctypes as C
    import torch
    lib = hip.L.load()
    d = hip.FromPlanarData()
    assert lib.ug_hip_from_planar(b"no_such_conversion", C.byref(d), None) == hip.L.EINVAL
    assert lib.ug_hip_from_planar_supported(b"gbrp12le_to_r12l") == 1 and lib.ug_hip_from_planar_supported(b"x") == 0
    p = torch.zeros((4, 8), dtype=torch.int16, device="cuda")
    out = torch.zeros((4, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        hip.from_planar("rgbpXXle_to_r12l", [p, p, p], 8, 4, out, 64, in_depth=10)  # the reference would shift by a negative count
    with pytest.raises(Exception):
        hip.from_planar("gbrp12le_to_rgb", [p, p, p], 8, 4, out, 0)  # a pitch of 0 would fold every line onto the first


@pytest.mark.gpu
def test_gpu_r12l_round_trip_4k(hip):
    """gbrp12le -> R12L -> gbrp12le at 3840x2160 is the identity (full-size property; no oracle involved)"""
    import torch
    w, h = 3840, 2160
    g = torch.Generator(device="cuda").manual_seed(9)
    planes = [torch.randint(0, 4096, (h, w), generator=g, device="cuda", dtype=torch.int16) for _ in range(3)]
    pitch = w // 8 * 36
    packed = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
    hip.from_planar("gbrp12le_to_r12l", planes, w, h, packed, pitch)
    back = [torch.zeros((h, w), dtype=torch.int16, device="cuda") for _ in range(3)]
    hip.to_planar("r12l_to_gbrp12le", packed, w, h, back)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(planes, back))
