#!/usr/bin/env python3
"""Random scripts for the runtime-conventions harness (tests/test_runtime_conventions.py holds the fixed ones): sequences of format changes, CHANGE_COMPRESS
messages from the capture side and from control threads, paces, sender delays, senders that hold frames, encoder failures, teardown with and without a pill --
through the reference's compress framework with the test-only fake module (the product's sharder and state structure), plain / TSan / ASan in turn.
Checks per run: no hang, no sanitizer report, exit 0, every delivered frame intact and encoded under the configuration of its own format, push order kept,
a configuration never coming back once another took over, everything pushed after the last change delivered (minus injected failures), every module state destroyed.
CPU only.   usage: tools/fuzz_runtime_conventions.py [runs=200] [seed=1]"""
import os
import random
import struct
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_runtime_conventions as T  # noqa: E402


def one(rng: random.Random, binary: str, tmp: Path) -> str:
    sets = T._fake_sets(tmp)
    names = list(sets)
    tag = 1
    fail_every = rng.choice([0, 0, 0, 7])

    def cfg(t):
        c = f"fake:tag={t}:workers={rng.choice([1, 2, 3, 4])}:delay_us={rng.choice([0, 200, 800, 2500])}"
        if rng.random() < 0.6:
            c += f":batch={rng.choice([2, 4, 8])}"
        if rng.random() < 0.3:
            c += ":dev=0,1"
        if fail_every:
            c += f":fail_every={fail_every}"
        return c

    nosender = rng.random() < 0.1
    body = f"sender_holds {rng.choice([0, 1])}\npop_delay_us {rng.choice([0, 0, 200, 1200])}\n" + ("init_nosender " if nosender else "init ") + cfg(tag) + "\n"
    pushes = []          # per push index: (set, tags allowed, generation)
    ctl_tags = []
    last_change_at = 0
    for _ in range(rng.randint(2, 9)):
        r = rng.random()
        if nosender:
            n = rng.choice(names)
            k = rng.randint(1, 3)
            for _ in range(k):
                body += f"push {n} 1\npop 1\n"
                pushes.append((n, None, 0))
        elif r < 0.55:
            n, k = rng.choice(names), rng.choice([1, 2, 5, 17, 40])
            body += f"push {n} {k}\n"
            pushes += [(n, None, 0)] * k
        elif r < 0.7:
            tag += 1
            body += f"msg {cfg(tag)}\n"
            last_change_at = len(pushes)
        elif r < 0.8:
            tag += 1
            ctl_tags.append(tag)
            body += f"msg_ctl {rng.choice([0, 3, 20])} {cfg(tag)}\n"
            last_change_at = None          # unknown: lands wherever
        elif r < 0.9:
            body += f"pace_us {rng.choice([0, 100, 900])}\n"
        else:
            body += f"sleep_ms {rng.choice([1, 10, 40])}\n"
    if ctl_tags:                            # let the control threads' messages land, then a tail that must come through whole
        body += "sleep_ms 60\npush A 1\nsleep_ms 20\n"
        pushes.append(("A", None, 0))
        last_change_at = len(pushes)
        body += "push B 6\n"
        pushes += [("B", None, 0)] * 6
    if not nosender and rng.random() < 0.5:
        body += "pill\n"
    body += "done\n"
    all_tags = set(range(1, tag + 1))
    pushes = [(n, all_tags, 0) for n, _, _ in pushes]
    if nosender and fail_every:            # a failed frame would leave `pop 1` waiting for ever: the framework's own behaviour, not a scenario
        return "skipped"
    records, out = T._run(binary, tmp, T._script(sets, body), T.SAN_ENV[binary])
    T._check_fake(records, pushes, sets)
    tags = [struct.unpack_from("<I", r["data"], 4)[0] for r in records]
    runs = [t for i, t in enumerate(tags) if i == 0 or tags[i - 1] != t]     # (control-thread messages land in any order: a configuration is one contiguous run)
    assert len(runs) == len(set(runs)), f"a configuration came back after another had taken over: {runs}\n" + body
    assert "FAKE live_states=0" in out, out
    if last_change_at is not None and not fail_every:
        tail = [r["index"] for r in records if r["index"] >= last_change_at]
        assert tail == list(range(last_change_at, len(pushes))), f"frames pushed after the last change are missing: {tail} of {last_change_at}..{len(pushes) - 1}\n" + body
    return f"{len(records)}/{len(pushes)}"


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    done = skipped = delivered = pushed = 0
    for i in range(runs):
        binary = T.FAKE_BINARIES[i % 3]
        if not os.path.exists(T._binary(binary)):
            continue
        with tempfile.TemporaryDirectory() as td:
            try:
                r = one(rng, binary, Path(td))
            except BaseException as e:                       # pytest.skip raises a BaseException too
                if type(e).__name__ == "Skipped":
                    skipped += 1
                    continue
                print(f"run {i} ({binary}, seed {seed}) FAILED: {e}")
                print(open(os.path.join(td, "script.txt")).read() if os.path.exists(os.path.join(td, "script.txt")) else "")
                raise
        if r == "skipped":
            skipped += 1
            continue
        done += 1
        a, b = (int(x) for x in r.split("/"))
        delivered += a
        pushed += b
    print(f"fuzz_runtime_conventions: {done} random scripts (seed {seed}; plain / TSan / ASan in turn), {skipped} skipped: {delivered} of {pushed} frames delivered, "
          f"every check held")


if __name__ == "__main__":
    main()
