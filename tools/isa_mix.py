#!/usr/bin/env python3
"""Instruction mix of one kernel of a hipcc -S --cuda-device-only listing, whole body and per basic block.
    python tools/isa_mix.py /tmp/x.s <substring of the mangled name>"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".size"))
body = [l.strip() for l in lines[start + 1:end]]
blocks, cur, name = [], [], "entry"
for l in body:
    if not l or l.startswith((";", "//")):
        continue
    if l.endswith(":") or (l.split(";")[0].strip().endswith(":")):
        blocks.append((name, cur))
        name, cur = l.split(":")[0], []
        continue
    if l.startswith("."):
        continue
    cur.append(l.split()[0])
blocks.append((name, cur))


def classes(ops):
    c = collections.Counter()
    for op in ops:
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos")): c["trans"] += 1
        elif op.startswith("v_cvt"): c["cvt"] += 1
        elif op.startswith("v_pk"): c["pk"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    return dict(c)


tot = [op for _, ops in blocks for op in ops]
print("whole kernel:", len(tot), classes(tot))
for name, ops in blocks:
    if len(ops) >= 40:
        print(f"  block {name:14s} {len(ops):5d}", classes(ops))
