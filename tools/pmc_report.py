#!/usr/bin/env python3
"""Per-kernel PMC digest from one or more rocprofv3 rocpd databases (counters are per shader-engine
samples: the digest multiplies the per-dispatch average by the number of samples per dispatch).

    python tools/pmc_report.py gpurun_out/pmc1/pmc1_results.db [more.db ...] [--filter nnpops]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    dbs = [a for a in sys.argv[1:] if a.endswith(".db")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else "nnpops"
    table = defaultdict(dict)
    dur = {}
    for db in dbs:
        c = sqlite3.connect(db)
        rows = []
        for one in flt.split(","):                               # --filter a,b: kernels whose name contains a OR b
            rows += c.execute("select k.name, p.counter_name, sum(p.counter_value), count(distinct k.dispatch_id), avg(k.duration) "
                              "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                              "where k.name like ? group by k.name, p.counter_name", (f"%{one}%",)).fetchall()
        for name, counter, total, ndisp, d in rows:
            key = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("nnpops::", "")
            key = key.split("(")[0][:48]
            table[key][counter] = total / ndisp          # chip total per dispatch
            dur[key] = d / 1e3
    for key, counters in table.items():
        print(f"== {key}   ({dur[key]:.1f} us per dispatch)")
        waves = counters.get("SQ_WAVES", 0)
        for cn in sorted(counters):
            v = counters[cn]
            per_wave = f"  per-wave {v / waves:12.1f}" if waves else ""
            print(f"   {cn:26s} {v:16.0f}{per_wave}")


if __name__ == "__main__":
    main()
