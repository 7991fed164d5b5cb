"""bench.py quotes the HBM traffic / VALU counters of its kernel from profiles/pmc_traffic.json (rocprofv3 --pmc passes cannot run inside the
timed process).  The entry names the kernel sources it was taken on; counters of another build are not quoted (CPU test of that rule)."""
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_counters_are_quoted_only_for_the_committed_kernels():
    """Whatever state the tree is in -- counters re-taken after the last kernel edit or not yet --, an entry is either quoted with the note that
    it belongs to this build, or withheld as STALE: never quoted for other sources.  (Not a freshness requirement: the counter passes need a GPU.)"""
    bench = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("uyvy_dxt5_4k_x16", "v210_dxt5_8k_x4", "rgb_dxt1_1080p_x64", "uyvy_jpeg420_4k_x8"):
        assert d[key].get("kernel_sources_sha16"), key
        pmc = bench.load_pmc(key)
        if pmc.get("traffic") is not None:
            assert pmc["traffic"] == d[key]["traffic"] and "= the build timed here" in pmc["source"], (key, pmc.get("source"))
        else:
            assert pmc["source"].startswith("STALE") and pmc.get("valu_instr_per_wave") is None, (key, pmc)
    # headline: the committed traffic is within 0.5 % of the algorithmic 3.0 B/px x 16 x 3840 x 2160
    assert abs(d["uyvy_dxt5_4k_x16"]["traffic"] / (3.0 * 16 * 3840 * 2160) - 1) < 0.005


def test_counters_of_another_build_are_not_quoted(tmp_path):
    bench = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    (tmp_path / "profiles").mkdir()
    import hashlib
    h = hashlib.sha256()
    for f in d["uyvy_dxt5_4k_x16"]["kernel_sources"]:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(ROOT, f), tmp_path / f)
        h.update(open(os.path.join(ROOT, f), "rb").read())
    # (the entry as a counter pass over THIS tree would write it -- whether the committed one is of this tree or of an earlier kernel is the test above's subject)
    d["uyvy_dxt5_4k_x16"] = dict(d["uyvy_dxt5_4k_x16"], kernel_sources_sha16=h.hexdigest()[:16])
    json.dump(d, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    assert bench.load_pmc("uyvy_dxt5_4k_x16", root=str(tmp_path)).get("traffic") == d["uyvy_dxt5_4k_x16"]["traffic"]
    with open(tmp_path / "ultragrid_amd/csrc/dxt_encode.hip", "a") as f:
        f.write("\n// a kernel change\n")
    pmc = bench.load_pmc("uyvy_dxt5_4k_x16", root=str(tmp_path))
    assert pmc.get("traffic") is None and pmc.get("valu_instr_per_wave") is None and pmc["source"].startswith("STALE")
    # an entry without a hash (counters of an earlier round) is quoted as it is, with its own source note
    legacy = dict(d, legacy_entry={"traffic": 123, "valu_instr_per_wave": 4.5, "source": "counters of an earlier round"})
    json.dump(legacy, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    assert bench.load_pmc("legacy_entry", root=str(tmp_path)) == legacy["legacy_entry"]
    assert bench.load_pmc("no_such_workload", root=str(tmp_path)) == {}
    # a hash without its file list (a hand-edited entry): stale, not a KeyError that takes the whole bench line down
    json.dump({"broken": {"traffic": 1, "kernel_sources_sha16": "0123456789abcdef"}}, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    broken = bench.load_pmc("broken", root=str(tmp_path))
    assert broken.get("traffic") is None and broken["source"].startswith("STALE")
