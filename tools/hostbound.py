import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import AniSymmetryFunctions
pos, species, box = workloads.random_box(10000, density=0.1, seed=100, n_species=7)
rf, af = workloads.ani2x_functions()
dev = torch.device("cuda:0")
sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
radial = torch.empty((10000, sym.radial_width), device=dev); angular = torch.empty((10000, sym.angular_width), device=dev)
g_r, g_a = torch.randn_like(radial), torch.randn_like(angular); grad = torch.empty((10000, 3), device=dev)
sym.compute(tpos, tbox, radial, angular, check=True)
for _ in range(50):
    sym.compute(tpos, tbox, radial, angular, check=False); sym.backprop(g_r, g_a, grad)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(500):
        sym.compute(tpos, tbox, radial, angular, check=False); sym.backprop(g_r, g_a, grad)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e6*(t1-t0)/500:.1f} us/step, total {1e6*(t2-t0)/500:.1f} us/step")
