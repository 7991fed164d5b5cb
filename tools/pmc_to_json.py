#!/usr/bin/env python3
"""Update profiles/pmc_traffic.json from rocprofv3 counter summaries (tools/pmc_summary.py output).
usage: pmc_to_json.py <entry> <kernel-substring> <source note> <summary.txt> [...]
Takes SQ_INSTS_VALU / SQ_WAVES, TCC_EA0_RDREQ_sum x 128 B and WRITE_SIZE x 1024 B of the named kernel (MI355X_MICROARCH.md: FETCH_SIZE
reports half of the read bytes on gfx950, so reads come from the TCC request count) and rewrites that entry; the others stay."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name substring -> the sources that kernel is compiled from; their hash travels with the counters so that bench.py can tell
# counters of another build from counters of the build it times ("the line cannot notice a kernel change")
KERNEL_SOURCES = {
    "dxt_encode_kernel": ["ultragrid_amd/csrc/dxt_encode.hip", "ultragrid_amd/csrc/ug_common.h"],
    "uyvy_jpeg_fast": ["ultragrid_amd/csrc/jpeg_fdct.hip", "ultragrid_amd/csrc/jpeg_fdct_device.h", "ultragrid_amd/csrc/ug_common.h"],
    "jpeg_code_kernel": ["ultragrid_amd/csrc/jpeg_entropy.hip", "ultragrid_amd/csrc/jpeg_fdct_device.h", "ultragrid_amd/csrc/jpeg_huffman_tables.h", "ultragrid_amd/csrc/ug_common.h"],
}


def sources_of(kernel: str):
    for k, files in KERNEL_SOURCES.items():
        if k in kernel:
            return files
    return None


def sha16(files, root=ROOT):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    entry, kern, note = sys.argv[1:4]
    vals = {}
    for path in sys.argv[4:]:
        for line in open(path):
            if not line.startswith("pmc") or kern not in line:
                continue
            m = re.search(r"(\S+)\s+n=\s*\d+\s+avg=([0-9.e+]+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
    need = ("SQ_INSTS_VALU", "SQ_WAVES", "TCC_EA0_RDREQ_sum", "WRITE_SIZE")
    missing = [k for k in need if k not in vals]
    if missing:
        sys.exit(f"pmc_to_json: {entry}: counters missing: {missing}")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    rd, wr = vals["TCC_EA0_RDREQ_sum"] * 128, vals["WRITE_SIZE"] * 1024
    d[entry] = {"traffic": int(round(rd + wr)), "valu_instr_per_wave": round(vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], 1),
                "waves": int(vals["SQ_WAVES"]),
                "source": f"{note}: TCC_EA0_RDREQ_sum {vals['TCC_EA0_RDREQ_sum']:.6g} x 128 B = {rd / 1e6:.1f} MB read + WRITE_SIZE {vals['WRITE_SIZE']:.6g} KiB = "
                          f"{wr / 1e6:.1f} MB written; SQ_INSTS_VALU {vals['SQ_INSTS_VALU']:.6g} / SQ_WAVES {int(vals['SQ_WAVES'])}"}
    files = sources_of(kern)
    if files:
        d[entry]["kernel_sources"] = files
        d[entry]["kernel_sources_sha16"] = sha16(files)
    json.dump(d, open(path, "w"), indent=1)
    print(entry, d[entry])

if __name__ == "__main__":
    main()
