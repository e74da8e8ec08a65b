#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (``*_results.db``) into the per-kernel summary committed under
profiles/ (same columns as rocprofv3's kernel_stats: calls, total, average, min, max, %).

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db profiles/r01_x_kernel_stats.txt ["header note"]
"""
import re
import sqlite3
import sys


def short(name, width=110):
    name = re.sub(r"\(.*$", "", name) if name.startswith("void nnpops::") else name
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    pmc = []
    try:
        pmc = c.execute("select k.name, p.counter_name, count(*), avg(p.counter_value) from pmc_events p "
                        "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        pass
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace summary ({db})\n")
        if note:
            f.write(f"# {note}\n")
        f.write("# durations in microseconds\n")
        f.write(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel\n")
        for name, n, tot, avg, mn, mx in rows:
            f.write(f"{n:6d} {tot / 1e3:12.1f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * tot / total:6.2f}  {short(name)}\n")
        if pmc:
            f.write("\n# PMC counters (average per dispatch)\n")
            for name, counter, n, avg in pmc:
                f.write(f"{counter:>28} {avg:16.1f}  n={n:<5d} {short(name, 80)}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
