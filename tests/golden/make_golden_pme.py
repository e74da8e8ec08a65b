#!/usr/bin/env python3
"""Generate direct-space PME golden vectors from the REFERENCE's own CPU ops.

Authoring container only (needs /root/reference; compiles src/pytorch/pme/{pme,pmeCPU}.cpp and
src/pytorch/neighbors/{neighbors,getNeighborPairsCPU}.cpp against the installed libtorch under /tmp -- nothing of the
reference is written into this repository):

    python tests/golden/make_golden_pme.py

Output: tests/golden/pme_ref.npz.  For each case the inputs (positions, charges, box, cutoff, alpha, coulomb, exclusions as
given), the pair list the reference's getNeighborPairs produced for them, and the outputs of torch.ops.pme.pme_direct with
its autograd: energy, dE/dpositions, dE/dcharges.  Cases 0-2 are the three systems of the reference's own test
(src/pytorch/pme/TestPme.py:17-170, rectangular / triclinic / triclinic with exclusions), whose expected OpenMM energies are
recorded alongside; the rest are seeded random systems with random exclusions.
"""
import glob
import os

import numpy as np
import torch
from torch.utils.cpp_extension import load

R = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = "/tmp/oracle/ext_pme"

POS_RECT = [[0.7713206433, 0.02075194936, 0.6336482349], [0.7488038825, 0.4985070123, 0.2247966455],
            [0.1980628648, 0.7605307122, 0.1691108366], [0.08833981417, 0.6853598184, 0.9533933462],
            [0.003948266328, 0.5121922634, 0.8126209617], [0.6125260668, 0.7217553174, 0.2918760682],
            [0.9177741225, 0.7145757834, 0.542544368], [0.1421700476, 0.3733407601, 0.6741336151],
            [0.4418331744, 0.4340139933, 0.6177669785]]
POS_TRIC = [[1.31396193, -0.9377441519, 0.9009447048], [1.246411648, 0.4955210369, -0.3256100634],
            [-0.4058114057, 1.281592137, -0.4926674903], [-0.7349805575, 1.056079455, 1.860180039],
            [-0.988155201, 0.5365767902, 1.437862885], [0.8375782005, 1.165265952, -0.1243717955],
            [1.753322368, 1.14372735, 0.627633104], [-0.5734898572, 0.1200222802, 1.022400845],
            [0.3254995233, 0.30204198, 0.8533009354]]
EXCL = [[3, -1], [-1, -1], [-1, 3], [0, 2], [-1, -1], [-1, -1], [-1, -1], [-1, 8], [7, -1]]


def load_reference():
    os.makedirs(BUILD, exist_ok=True)
    so = glob.glob(f"{BUILD}/*.so")
    if so:
        torch.ops.load_library(so[0])
        return
    load(name="libNNPOpsPmeRef", sources=[f"{R}/pytorch/pme/pme.cpp", f"{R}/pytorch/pme/pmeCPU.cpp",
                                          f"{R}/pytorch/neighbors/neighbors.cpp", f"{R}/pytorch/neighbors/getNeighborPairsCPU.cpp"],
         is_python_module=False, with_cuda=False, extra_cflags=["-O2"], build_directory=BUILD)


def run(pos, charges, box, cutoff, alpha, coulomb, excl, openmm_energy=None):
    positions = torch.tensor(pos, dtype=torch.float32, requires_grad=True)
    q = torch.tensor(charges, dtype=torch.float32, requires_grad=True)
    tb = torch.tensor(box, dtype=torch.float32)
    exclusions = torch.tensor(excl, dtype=torch.int32).reshape(len(pos), -1)
    srt, _ = torch.sort(exclusions, descending=True)                       # what the reference's PME class does (pme.py:93)
    nb, dl, ds, _ = torch.ops.neighbors.getNeighborPairs(positions, cutoff, -1, tb, False)
    e = torch.ops.pme.pme_direct(positions, q, nb, dl, ds, srt, alpha, coulomb)
    e.backward()
    return {"positions": positions.detach().numpy(), "charges": q.detach().numpy(), "box": tb.numpy(), "cutoff": np.float64(cutoff),
            "alpha": np.float64(alpha), "coulomb": np.float64(coulomb), "exclusions": exclusions.numpy(),
            "neighbors": nb.numpy(), "deltas": dl.detach().numpy(), "distances": ds.detach().numpy(),
            "energy": np.float64(e.item()), "pos_grad": positions.grad.numpy(), "charge_grad": q.grad.numpy(),
            "openmm_energy": np.float64(np.nan if openmm_energy is None else openmm_energy)}


def random_case(rng, n, max_excl):
    L = 2.4
    box = np.array([[L, 0, 0], [0.2 * L, 1.05 * L, 0], [-0.1 * L, 0.15 * L, 0.95 * L]])
    pos = rng.random((n, 3)) * L * 1.5 - 0.3                               # some atoms outside the primary cell
    charges = rng.normal(0, 0.4, n)
    excl = -np.ones((n, max_excl), np.int64)
    fill = np.zeros(n, int)
    for _ in range(n):                                                     # symmetric random exclusions
        i, j = rng.integers(0, n, 2)
        if i != j and fill[i] < max_excl and fill[j] < max_excl and j not in excl[i]:
            excl[i, fill[i]] = j; fill[i] += 1
            excl[j, fill[j]] = i; fill[j] += 1
    return run(pos, charges, box, 1.0, 3.2, 138.935, excl)


def main():
    load_reference()
    q9 = [(i - 4) * 0.1 for i in range(9)]
    cases = [run(POS_RECT, q9, [[1, 0, 0], [0, 1.1, 0], [0, 0, 1.2]], 0.5, 4.985823141035867, 138.935, np.zeros((9, 0)), 0.5811535194516182),
             run(POS_TRIC, q9, [[1, 0, 0], [-0.1, 1.2, 0], [0.2, -0.15, 1.1]], 0.5, 5.0, 138.935, np.zeros((9, 0)), -178.86083489656448),
             run(POS_TRIC, q9, [[1, 0, 0], [-0.1, 1.2, 0], [0.2, -0.15, 1.1]], 0.5, 5.0, 138.935, EXCL, -204.22671127319336)]
    rng = np.random.default_rng(11)
    cases += [random_case(rng, 60, 3), random_case(rng, 150, 4), random_case(rng, 150, 0)]
    out = {"num_cases": np.int64(len(cases))}
    for k, c in enumerate(cases):
        for name, v in c.items():
            out[f"c{k}_{name}"] = v
    np.savez_compressed(os.path.join(HERE, "pme_ref.npz"), **out)
    print("pme_ref.npz:", [(round(float(c["energy"]), 4), float(c["openmm_energy"])) for c in cases])


if __name__ == "__main__":
    main()
