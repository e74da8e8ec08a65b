"""getNeighborPairs backward at 100 000 atoms: fixed-point atomics (rounds 4-5) against the indexed gather (round 6), and what the
transposed index costs the forward op.   python tools/pairs_bwd_time.py [slots]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads
from nnpops_amd.capi import (neighbor_pairs_backward, neighbor_pairs_backward_indexed, neighbor_pairs_build_index,
                             neighbor_pairs_forward)

dev = torch.device("cuda:0")
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for dt in (torch.float32, torch.float64):
    pos, _, box = workloads.random_box(100000, density=0.1, seed=3)
    tp = torch.tensor(pos, device=dev, dtype=dt)
    tb = torch.tensor(box, device=dev, dtype=dt)
    nb, dl, ds, cnt = neighbor_pairs_forward(tp, 5.0, slots, tb)
    gd = torch.randn_like(dl)
    gs = torch.randn_like(ds)
    index = neighbor_pairs_build_index(100000, nb)
    a = neighbor_pairs_backward(100000, nb, dl, ds, gd, gs)
    b = neighbor_pairs_backward_indexed(100000, nb, dl, ds, gd, gs, index)
    err = float((a - b).abs().max() / a.abs().max())
    print(dt, "pairs", int(cnt), "slots", slots, "max rel diff fixed vs gather %.2e" % err)
    print("   forward op              %8.1f us" % timed(lambda: neighbor_pairs_forward(tp, 5.0, slots, tb), 20))
    print("   transposed index        %8.1f us" % timed(lambda: neighbor_pairs_build_index(100000, nb), 20))
    print("   backward, fixed point   %8.1f us" % timed(lambda: neighbor_pairs_backward(100000, nb, dl, ds, gd, gs)))
    print("   backward, gather        %8.1f us" % timed(lambda: neighbor_pairs_backward_indexed(100000, nb, dl, ds, gd, gs, index)))
