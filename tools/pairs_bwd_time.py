import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import neighbor_pairs_forward, neighbor_pairs_backward
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.float64):
    pos, _, box = workloads.random_box(100000, density=0.1, seed=3)
    tp = torch.tensor(pos, device=dev, dtype=dt); tb = torch.tensor(box, device=dev, dtype=dt)
    nb, dl, ds, cnt = neighbor_pairs_forward(tp, 5.0, 3000000, tb)
    gd = torch.randn_like(dl); gs = torch.randn_like(ds)
    for _ in range(5): neighbor_pairs_backward(100000, nb, dl, ds, gd, gs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): neighbor_pairs_backward(100000, nb, dl, ds, gd, gs)
    e1.record(); torch.cuda.synchronize()
    print(dt, "pairs", int(cnt), "backward us per call", 1e3 * e0.elapsed_time(e1) / 50)
