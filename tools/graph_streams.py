#!/usr/bin/env python3
"""Experiment: one evaluation captured as a HIP graph, with the atoms split over 1..4 streams inside the handle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import AniSymmetryFunctions
dev = torch.device("cuda:0")
rf, af = workloads.ani2x_functions()
pos, species, box = workloads.random_box(10000, density=0.1, seed=100, n_species=7)
tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
for k in (1, 2, 3, 4):
    os.environ["NNPOPS_ANI_STREAMS"] = str(k)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    n = 10000
    radial = torch.empty((n, sym.radial_width), device=dev); angular = torch.empty((n, sym.angular_width), device=dev)
    g_r, g_a = torch.randn_like(radial), torch.randn_like(angular); grad = torch.empty((n, 3), device=dev)
    sym.compute(tpos, tbox, radial, angular, check=True)
    def step():
        sym.compute(tpos, tbox, radial, angular, check=False); sym.backprop(g_r, g_a, grad)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    ref = grad.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(ref, grad)
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize()
    print(f"streams={k}: graph replay {1e6 * (time.perf_counter() - t0) / 300:.1f} us per step")
