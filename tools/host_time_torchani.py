"""Host-side cost of one OptimizedTorchANI energy+forces step at config 2 (2001-atom water box, 8 members): where the eager
loop's time goes once the device is no longer the bottleneck.  Run on the GPU box:  python tools/host_time_torchani.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nnpops_amd import workloads
from NNPOps import OptimizedTorchANI

dev = torch.device("cuda:0")
model = workloads.torchani_like_model(n_models=8, seed=2)
pos, species, box = workloads.water_box(667, seed=1)
numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=dev)
opt = OptimizedTorchANI(model, numbers.cpu()).to(dev)
cell, pbc = torch.tensor(box, device=dev), torch.tensor([True, True, True])
tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
for interval in (1, 0):
    opt.set_check_interval(interval)
    t_fwd = t_sum = t_bwd = 0.0
    n = 300
    for it in range(n + 20):
        if it == 20:
            torch.cuda.synchronize(); t_fwd = t_sum = t_bwd = 0.0; t0 = time.perf_counter()
        tpos.grad = None
        a = time.perf_counter()
        e = opt((numbers, tpos), cell, pbc).energies
        b = time.perf_counter()
        s = e.sum()
        c = time.perf_counter()
        s.backward()
        d = time.perf_counter()
        t_fwd += b - a; t_sum += c - b; t_bwd += d - c
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"check interval {interval}: per step  module forward {1e6 * t_fwd / n:.1f} us | .sum() {1e6 * t_sum / n:.1f} us | .backward() {1e6 * t_bwd / n:.1f} us | "
          f"host loop {1e6 * host / n:.1f} us | wall {1e6 * wall / n:.1f} us")
