#!/usr/bin/env python3
"""Classify the VALU instructions of one kernel by issue class (profiles/valu_microbench_mi355x.txt) and print the
issue-bound model  clk/wave/CU = max(N_valu / 1.65, N_slow / 0.95).

usage: isa_classify.py <file.hip> <kernel-substring> [extra hipcc flags]
"""
import collections, re, subprocess, sys, os, tempfile

FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32(_e32|_e64)?$|^v_(and|or|xor|not)_b32|^v_mov_b32|^v_(add|sub|subrev)_u32|^v_(add|sub)_co_u32")

def main():
    src, kern = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    here = os.path.dirname(os.path.abspath(src))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-S", "--cuda-device-only",
               "-I", os.path.join(here, "../../include"), "-I", here, src, "-o", out] + extra
        subprocess.check_call(cmd)
        text = open(out).read()
    # split into functions
    m = None
    for mm in re.finditer(r"^(_Z\S*" + re.escape(kern) + r"\S*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
        m = mm
        body = mm.group(2)
        cnt = collections.Counter()
        fast = slow = salu = vmem = lds = 0
        for line in body.splitlines():
            line = line.strip()
            if not line or line.startswith((";", ".")) or line.endswith(":"):
                continue
            op = line.split()[0]
            cnt[op] += 1
            if op.startswith("v_"):
                sgpr_operand = bool(re.search(r"[ ,]s\d+|[ ,]s\[\d+:\d+\]|vcc|exec", line.split(";")[0].split(None, 1)[1] if " " in line else ""))
                if FAST.match(op) and not (sgpr_operand and not op.startswith("v_mov")) and "dpp" not in line and "sdwa" not in line:
                    fast += 1
                else:
                    slow += 1
            elif op.startswith("s_"):
                salu += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                vmem += 1
            elif op.startswith("ds_"):
                lds += 1
        nv = fast + slow
        print(f"{mm.group(1)[:90]}")
        print(f"  VALU {nv} (fast {fast}, slow {slow})  SALU {salu}  VMEM {vmem}  LDS {lds}")
        print(f"  model clk/wave/CU: issue {nv / 1.65:.0f}, slow-pipe {slow / 0.95:.0f} -> {max(nv / 1.65, slow / 0.95):.0f}")
        top = ", ".join(f"{k}:{v}" for k, v in cnt.most_common(28))
        print("  " + top)
    if m is None:
        print("kernel not found")

if __name__ == "__main__":
    main()
