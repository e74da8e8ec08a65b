#!/usr/bin/env python3
"""Experiment: do two half-size evaluations on two HIP streams finish sooner than one full-size evaluation on one stream?
(ramps and tails of the per-atom kernels overlapping with the other stream's steady state)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import AniSymmetryFunctions

dev = torch.device("cuda:0")
rf, af = workloads.ani2x_functions()

def make(n, seed):
    pos, species, box = workloads.random_box(n, density=0.1, seed=seed, n_species=7)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    radial = torch.empty((n, sym.radial_width), device=dev); angular = torch.empty((n, sym.angular_width), device=dev)
    g_r, g_a = torch.randn_like(radial), torch.randn_like(angular); grad = torch.empty((n, 3), device=dev)
    sym.compute(tpos, tbox, radial, angular, check=True)
    def step():
        sym.compute(tpos, tbox, radial, angular, check=False); sym.backprop(g_r, g_a, grad)
    return step

def timeit(fn, k=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / k

full = make(10000, 100)
print("one 10k frame, one stream      : %.1f us" % timeit(full))
for prio in (False, True):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s1 = torch.cuda.Stream(priority=-1 if prio else 0); s2 = torch.cuda.Stream(priority=0)
    with torch.cuda.stream(s1): a = make(5000, 101)
    with torch.cuda.stream(s2): b = make(5000, 102)
    torch.cuda.synchronize()
    def both():
        with torch.cuda.stream(s1): a()
        with torch.cuda.stream(s2): b()
    print("two 5k frames, two streams%s: %.1f us" % (" (one high priority)" if prio else "                    ", timeit(both)))
    def serial():
        with torch.cuda.stream(s1): a(); b()
    print("two 5k frames, one stream      : %.1f us" % timeit(serial))
