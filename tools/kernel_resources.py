#!/usr/bin/env python3
"""Registers / scratch / occupancy of the kernels of a hipcc -Rpass-analysis=kernel-resource-usage log.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o /tmp/x.o 2> /tmp/res.txt;  python tools/kernel_resources.py /tmp/res.txt [filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip().split()[0].strip("[]") for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")


def g(b, k):
    m = re.search(re.escape(k) + r": (\d+)", b)
    return int(m.group(1)) if m else -1


for b, dn in zip(blocks, dem):
    short = re.sub(r"\(.*", "", dn.replace("(anonymous namespace)::", "")).replace("void nnpops::", "").replace("void ", "")
    if filters and not any(f in short for f in filters):
        continue
    print("%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %2d" % (short[:100], g(b, "VGPRs"), g(b, "AGPRs"), g(b, "SGPRs"),
          g(b, "ScratchSize [bytes/lane]"), g(b, "Occupancy [waves/SIMD]")))
