#!/usr/bin/env python3
"""Generate BatchedNN golden vectors from the REFERENCE's own CPU op.

Authoring container only (needs /root/reference; compiles src/pytorch/BatchedNN.cpp against the installed libtorch
under /tmp -- nothing of the reference is written into this repository):

    python tests/golden/make_golden_batched_nn.py

Output: tests/golden/batched_nn_ref.npz.  The networks are the seeded ANI-2x-shaped stand-ins of
nnpops_amd.workloads.torchani_like_model (torchani's real weights are not available offline); the script packs
them into the reference's per-atom replicated weight tensors the way src/pytorch/BatchedNN.py:73-95 does, then runs the
reference composition (src/pytorch/BatchedNN.py:97-119: BatchedLinear, CELU(0.1) x3, BatchedLinear, sum / models) with
torch.ops.NNPOpsBatchedNN.BatchedLinear -- forward AND its autograd backward (src/pytorch/BatchedNN.cpp:30-42).
Stored: species, the AEV input, the energy, dE/dAEV, the output of the first BatchedLinear for atom 0, and a checksum of
the regenerated weights (so that a test failing because torch's generator changed says so) and the sum of |atomic
energies| (the scale an energy error is measured against: the random networks' energies cancel).
"""
import glob
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.cpp_extension import load

R = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = "/tmp/oracle/ext_batchednn"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def load_reference():
    os.makedirs(BUILD, exist_ok=True)
    so = glob.glob(f"{BUILD}/*.so")
    if so:
        torch.ops.load_library(so[0])
        return
    load(name="libNNPOpsBatchedNNRef", sources=[f"{R}/pytorch/BatchedNN.cpp"], is_python_module=False, with_cuda=False,
         extra_cflags=["-O2"], build_directory=BUILD)


def pack(layers):
    """[models][atoms] Linear -> weights [1, atoms, models, max_out, max_in], biases [1, atoms, models, max_out, 1]."""
    num_models, num_atoms = len(layers), len(layers[0])
    flat = sum(layers, [])
    max_out, max_in = max(l.out_features for l in flat), max(l.in_features for l in flat)
    w = torch.zeros((1, num_atoms, num_models, max_out, max_in))
    b = torch.zeros((1, num_atoms, num_models, max_out, 1))
    for m, sub in enumerate(layers):
        for a, layer in enumerate(sub):
            o, i = layer.weight.shape
            w[0, a, m, :o, :i] = layer.weight.detach()
            b[0, a, m, :o, 0] = layer.bias.detach()
    return w, b


def case(n_models, seed, species, aev_seed):
    from nnpops_amd import workloads
    op = torch.ops.NNPOpsBatchedNN.BatchedLinear
    model = workloads.torchani_like_model(n_models=n_models, seed=seed)
    names = list(workloads.ANI2X_WIDTHS)                       # species index -> element symbol, ANI-2x order
    nets = [list(m.values()) for m in model.neural_networks]
    params = []
    for il in (0, 2, 4, 6):
        params.append(pack([[nets[m][s][il] for s in species] for m in range(n_models)]))
    gen = torch.Generator().manual_seed(aev_seed)
    aev = (0.3 * torch.randn((1, len(species), 1008), generator=gen)).abs().requires_grad_(True)
    v = aev.unsqueeze(-2).unsqueeze(-1)
    first = None
    for k, (w, b) in enumerate(params):
        v = op(v, w, b)
        if k == 0:
            first = v[0, 0, :, :, 0].detach().clone()
        if k < 3:
            v = F.celu(v, alpha=0.1)
    energy = torch.sum(v, (1, 2, 3, 4)) / v.shape[2]
    energy.sum().backward()
    checksum = sum(float(p.detach().double().abs().sum()) for net in model.neural_networks for p in net.parameters())
    scale = float(v.detach().abs().sum() / v.shape[2])         # sum of |atomic energies|: what an energy error is measured against
    return {"species": np.array(species, np.int32), "n_models": np.int64(n_models), "model_seed": np.int64(seed),
            "aev": aev.detach().numpy(), "energy": energy.detach().numpy(), "aev_grad": aev.grad.numpy(),
            "first_layer_atom0": first.numpy(), "weights_checksum": np.float64(checksum), "energy_scale": np.float64(scale)}


def main():
    load_reference()
    rng = np.random.default_rng(7)
    cases = [case(8, 2, list(range(7)) + rng.integers(0, 7, size=17).tolist(), 41),      # every ANI-2x species, 8 members
             case(2, 21, [0] * 11 + [3] * 4 + [6], 42),                                  # H, O, one Cl; kinds that never occur
             case(1, 5, [1], 43)]                                                         # one atom, one model
    out = {"num_cases": np.int64(len(cases))}
    for k, c in enumerate(cases):
        for name, v in c.items():
            out[f"c{k}_{name}"] = v
    np.savez_compressed(os.path.join(HERE, "batched_nn_ref.npz"), **out)
    print("batched_nn_ref.npz:", {f"c{k}": float(c["energy"][0]) for k, c in enumerate(cases)})


if __name__ == "__main__":
    main()
