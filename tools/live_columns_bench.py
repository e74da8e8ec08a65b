"""OptimizedTorchANI energy+forces step replayed as a HIP graph, networks over the live AEV columns vs over all 1008, for
frames with 2, 3, 4 and 7 of ANI-2x's species (2001 atoms, 8 members).  Run on the GPU box:  python tools/live_columns_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from nnpops_amd import workloads
from NNPOps import OptimizedTorchANI

dev = torch.device("cuda:0")
model = workloads.torchani_like_model(n_models=8, seed=2)
pos, species_w, box = workloads.water_box(667, seed=1)
rng = np.random.default_rng(3)
cell, pbc = torch.tensor(box, device=dev), torch.tensor([True, True, True])
for kinds in ([0, 3], [0, 1, 3], [0, 1, 2, 3], [0, 1, 2, 3, 4, 5, 6]):
    species = np.array(kinds)[rng.integers(0, len(kinds), size=len(species_w))] if len(kinds) != 2 else np.asarray(species_w)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[int(s)] for s in species]], device=dev)
    line = []
    for live in (True, False):
        opt = OptimizedTorchANI(model, numbers.cpu(), live_columns=live).to(dev)
        tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
        for _ in range(4):
            tpos.grad = None
            opt((numbers, tpos), cell, pbc).energies.sum().backward()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                tpos.grad = None
                opt((numbers, tpos), cell, pbc).energies.sum().backward()
        torch.cuda.current_stream().wait_stream(side)
        tpos.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            e = opt((numbers, tpos), cell, pbc).energies
            f = torch.autograd.grad(e.sum(), tpos)[0]
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            graph.replay()
        torch.cuda.synchronize()
        nets = opt.neural_networks[0]
        line.append((1e6 * (time.perf_counter() - t0) / 200, 16 * int(nets.x_blocks.numel()) or 1008, float(e), float(f.abs().max())))
        del graph, opt, e, f
    (tl, cl, el, fl), (td, cd, ed, fd) = line
    print(f"{len(kinds)} species: live columns {cl:4d}: {tl:6.1f} us   all {cd}: {td:6.1f} us   energy {el:.6f} / {ed:.6f}   max |force| {fl:.5f} / {fd:.5f}")
