#!/usr/bin/env python3
"""Summarise rocprofv3 results databases (rocpd sqlite): per kernel, average duration and average PMC
counter values per dispatch.  Usage: pmc_summary.py <results.db> [...] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print(f"== {path}")
        try:
            for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                print(f"kernel {name[:110]}\n   calls={calls} total_us={total:.1f} avg_us={avg:.3f} pct={pct:.2f}")
        except sqlite3.Error as e:
            print("  (no kernel trace:", e, ")")
        try:
            rows = c.execute("""select kernel_name, counter_name, count(*), avg(value), sum(value)
                                from counters_collection group by kernel_name, counter_name""").fetchall()
        except sqlite3.Error as e:
            print("  (no counters:", e, ")")
            rows = []
        for kname, counter, n, avg, tot in rows:
            print(f"pmc {kname[:70]:70s} {counter:28s} n={n:5d} avg={avg:.6g}")


if __name__ == "__main__":
    main()
