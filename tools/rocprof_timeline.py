#!/usr/bin/env python3
"""Print the kernel timeline of `n` dispatches from the MIDDLE of a rocprofv3 rocpd database (the steady state of a
timed loop; pass a third argument "last" for the last n): start offset,
duration and the idle gap before each kernel (microseconds).  Shows how much of a step is spent between
kernels (launch latency, dependencies) rather than in them.

    python tools/rocprof_timeline.py gpurun_out/prof_x/x_results.db [n]
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    if len(sys.argv) > 3 and sys.argv[3] == "last":
        rows = c.execute(f"select name, {s}, {e} from kernels order by {s} desc limit {n}").fetchall()[::-1]
    else:
        total = c.execute("select count(*) from kernels").fetchone()[0]
        rows = c.execute(f"select name, {s}, {e} from kernels order by {s} limit {n} offset {max(0, total // 2 - n // 2)}").fetchall()
    t0 = rows[0][1]
    prev_end = None
    busy = 0
    print(f"{'start_us':>10} {'dur_us':>8} {'gap_us':>8}  kernel")
    for name, a, b in rows:
        gap = 0.0 if prev_end is None else (a - prev_end) / 1e3
        busy += b - a
        name = re.sub(r"\(.*$", "", name)[:90]
        print(f"{(a - t0) / 1e3:10.2f} {(b - a) / 1e3:8.2f} {gap:8.2f}  {name}")
        prev_end = b
    span = rows[-1][2] - t0
    print(f"# span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {100 * (1 - busy / span):.1f} %")


if __name__ == "__main__":
    main()
