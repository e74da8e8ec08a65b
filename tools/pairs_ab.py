#!/usr/bin/env python3
"""Interleaved timing of getNeighborPairs variants (environment read at every call) on a periodic box.
    python tools/pairs_ab.py "NNPOPS_PAIRS_DIVIDE=1" "" [--atoms 100000]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import neighbor_pairs_forward
ap = argparse.ArgumentParser(); ap.add_argument("variants", nargs="+"); ap.add_argument("--atoms", type=int, default=100000)
a = ap.parse_args()
pos, _, box = workloads.random_box(a.atoms, density=0.1, seed=5, n_species=7)
dev = torch.device("cuda:0"); tp, tb = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
res = {v: [] for v in a.variants}
for r in range(7):
    for v in a.variants:
        saved = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=", 1); os.environ[k] = val
        neighbor_pairs_forward(tp, 5.2, 3200000, tb); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): neighbor_pairs_forward(tp, 5.2, 3200000, tb)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 20 * 1e6)
        os.environ.clear(); os.environ.update(saved)
for v in a.variants: print(f"{v or '(default)':30s} {sorted(res[v])[len(res[v]) // 2]:8.1f} us")
