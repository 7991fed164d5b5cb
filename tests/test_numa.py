"""NUMA placement entry points of the C-ABI, on CPU with a fake sysfs tree: the node of a PCI function,
the CPU list of a node, the calling thread's affinity -- never widened, restored by the test."""
import ctypes as C
import os
import threading

import pytest

from ultragrid_amd import lib


def _tree(tmp_path, bdf="0000:c1:00.0", node="1", cpulists=None):
    d = tmp_path / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text(node + "\n")
    for n, cl in (cpulists or {}).items():
        nd = tmp_path / "devices" / "system" / "node" / f"node{n}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    return str(tmp_path).encode()


def test_node_of_pci_function(tmp_path):
    l = lib.load()
    root = _tree(tmp_path, node="1")
    node = C.c_int(-7)
    assert l.ug_hip_numa_node_of_pci(b"0000:C1:00.0", root, C.byref(node)) == lib.SUCCESS and node.value == 1   # HIP prints upper-case hex
    assert l.ug_hip_numa_node_of_pci(b"0000:99:00.0", root, C.byref(node)) == lib.SUCCESS and node.value == -1  # not in sysfs: unknown
    assert l.ug_hip_numa_node_of_pci(b"../../etc", root, C.byref(node)) == lib.EINVAL
    assert l.ug_hip_numa_node_of_pci(None, root, C.byref(node)) == lib.EINVAL


def test_single_node_box_says_minus_one(tmp_path):
    l = lib.load()
    root = _tree(tmp_path, node="-1")
    node = C.c_int(5)
    assert l.ug_hip_numa_node_of_pci(b"0000:c1:00.0", root, C.byref(node)) == lib.SUCCESS and node.value == -1


def test_bind_calling_thread_only_and_never_widen(tmp_path):
    l = lib.load()
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    half = allowed[: len(allowed) // 2]
    # node 0 = the first half of what we may use + CPUs we may NOT use (4090..4095 are outside any cpuset here); node 2 = nothing of ours
    cl0 = ",".join(map(str, half)) + ",4090-4095"
    root = _tree(tmp_path, cpulists={0: cl0, 2: "4000-4001"})
    res = {}

    def worker():
        tid = threading.get_native_id()
        n = C.c_int(-1)
        res["rc"] = l.ug_hip_bind_thread_to_numa_node(0, root, C.byref(n))
        res["n"] = n.value
        res["mask"] = sorted(os.sched_getaffinity(tid))
        n2 = C.c_int(-1)
        res["rc2"] = l.ug_hip_bind_thread_to_numa_node(2, root, C.byref(n2))   # empty intersection: left alone
        res["n2"] = n2.value
        res["mask2"] = sorted(os.sched_getaffinity(tid))
        n3 = C.c_int(-1)
        res["rc3"] = l.ug_hip_bind_thread_to_numa_node(-1, root, C.byref(n3))  # unknown node: left alone
        res["n3"] = n3.value
        res["rc4"] = l.ug_hip_bind_thread_to_numa_node(9, root, C.byref(n3))   # node without a cpulist
        res["n4"] = n3.value

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert res["rc"] == lib.SUCCESS and res["n"] == len(half) and res["mask"] == half
    assert res["rc2"] == lib.SUCCESS and res["n2"] == 0 and res["mask2"] == half
    assert res["rc3"] == lib.SUCCESS and res["n3"] == 0 and res["rc4"] == lib.SUCCESS and res["n4"] == 0
    assert sorted(os.sched_getaffinity(0)) == allowed   # the main thread was not touched


def test_cpulist_forms(tmp_path):
    l = lib.load()
    allowed = sorted(os.sched_getaffinity(0))
    root = _tree(tmp_path, cpulists={3: f"{allowed[0]}", 4: f"{allowed[0]}-{allowed[-1]}"})
    out = {}

    def worker():
        n = C.c_int(0)
        l.ug_hip_bind_thread_to_numa_node(4, root, C.byref(n))
        out["all"] = n.value
        l.ug_hip_bind_thread_to_numa_node(3, root, C.byref(n))
        out["one"] = (n.value, sorted(os.sched_getaffinity(threading.get_native_id())))

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert out["all"] == len(allowed) and out["one"] == (1, [allowed[0]])
