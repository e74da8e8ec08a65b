#!/usr/bin/env python3
"""Generate getNeighborPairs golden vectors from the REFERENCE's own CPU op.

Authoring container only (needs /root/reference and ~80 s to compile the reference's CPU sources
against the installed libtorch, under /tmp -- nothing of the reference is written into this repo):

    python tests/golden/make_golden_torch_ref.py

Output: tests/golden/neighbors_ref.npz -- for each case the inputs (positions, cutoff, max_num_pairs,
box) and the four outputs of torch.ops.neighbors.getNeighborPairs on CPU (reference semantics:
src/pytorch/neighbors/getNeighborPairsCPU.cpp).  Cases follow the reference's own test matrix
(src/pytorch/neighbors/TestNeighbors.py) at sizes small enough to commit, plus the docstring examples.
"""
import glob
import os

import numpy as np
import torch
from torch.utils.cpp_extension import load

R = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = "/tmp/oracle/ext_neighbors"


def load_reference():
    os.makedirs(BUILD, exist_ok=True)
    so = glob.glob(f"{BUILD}/*.so")
    if so:
        torch.ops.load_library(so[0])
        return
    load(name="libNNPOpsNeighborsRef", sources=[f"{R}/pytorch/neighbors/getNeighborPairsCPU.cpp", f"{R}/pytorch/neighbors/neighbors.cpp"],
         is_python_module=False, with_cuda=False, extra_cflags=["-O2"], build_directory=BUILD)


def main():
    load_reference()
    op = torch.ops.neighbors.getNeighborPairs
    out = {}
    cases = []
    rng = np.random.default_rng(2024)
    for dtype in (np.float32, np.float64):
        for n in (1, 2, 3, 4, 5, 10, 60):
            for cutoff in (1.0, 10.0, 100.0):
                pos = (10 * rng.standard_normal((n, 3))).astype(dtype)
                for mode in ("all", "compact"):
                    cases.append((pos, cutoff, mode, None))
    tric = np.array([[10, 0, 0], [2, 12, 0], [0, 1, 11]], np.float64)      # TestNeighbors.py:219
    cubic = np.eye(3) * 10.0
    for dtype in (np.float32, np.float64):
        pos = (rng.random((40, 3)) * 30 - 15).astype(dtype)
        for box in (cubic, tric):
            for mode in ("all", "compact"):
                cases.append((pos, 5.0, mode, box.astype(dtype)))
    doc = np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]], np.float32)   # getNeighborPairs.py:104-138
    for cutoff, mnp in ((3.0, -1), (1.5, -1), (3.0, 6), (1.5, 6)):
        cases.append((doc, cutoff, mnp, None))
    for k, (pos, cutoff, mode, box) in enumerate(cases):
        tp = torch.tensor(pos)
        tb = torch.tensor(box) if box is not None else torch.empty((0, 0), dtype=tp.dtype)
        if mode == "all":
            mnp = -1
        elif mode == "compact":
            full = op(tp, cutoff, -1, tb, False)
            mnp = max(int((full[0][0] >= 0).sum()), 1) + 2          # two padding slots
        else:
            mnp = int(mode)
        nb, dl, ds, npairs = op(tp, cutoff, mnp, tb, False)
        out[f"c{k}_positions"] = pos
        out[f"c{k}_cutoff"] = np.float64(cutoff)
        out[f"c{k}_max_num_pairs"] = np.int64(mnp)
        out[f"c{k}_box"] = box if box is not None else np.zeros((0, 0), pos.dtype)
        out[f"c{k}_neighbors"] = nb.numpy()
        out[f"c{k}_deltas"] = dl.numpy()
        out[f"c{k}_distances"] = ds.numpy()
        out[f"c{k}_num_pairs"] = npairs.numpy()
    out["num_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "neighbors_ref.npz"), **out)
    print(f"neighbors_ref.npz: {len(cases)} cases")


if __name__ == "__main__":
    main()
