"""TorchANISymmetryFunctions -- drop-in replacement for ``torchani.AEVComputer`` backed by the HIP kernels.

Mirrors the reference wrapper (src/pytorch/SymmetryFunctions.py:31-123): same constructor arguments,
same ``forward((species, positions), cell, pbc)`` contract, same errors.  TorchANI itself is not
imported: the AEV computer and the species converter are duck-typed, so the module also works with
any object exposing the attributes read below.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import torch_binding

torch_binding.load()


class TorchANISymmetryFunctions(torch.nn.Module):
    """Optimized TorchANI symmetry functions.

    Arguments:
        converter: a ``torchani.nn.SpeciesConverter``-like callable; ``converter((Z, x)).species`` maps
            atomic numbers to species indices
        symmFunc: a ``torchani.AEVComputer``-like object (num_species, Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ)
        atomicNumbers: tensor of atomic numbers, shape [1, num_atoms]
    """

    def __init__(self, converter, symmFunc, atomicNumbers: Tensor) -> None:
        super().__init__()
        self.num_species = int(symmFunc.num_species)
        # attribute slicing as in the reference (SymmetryFunctions.py:75-83)
        constants = dict(
            EtaR=symmFunc.EtaR[:, 0], ShfR=symmFunc.ShfR[0, :],
            EtaA=symmFunc.EtaA[:, 0, 0, 0], Zeta=symmFunc.Zeta[0, :, 0, 0],
            ShfA=symmFunc.ShfA[0, 0, :, 0], ShfZ=symmFunc.ShfZ[0, 0, 0, :])
        lists = {k: [float(x) for x in v.tolist()] for k, v in constants.items()}
        species = converter((atomicNumbers, torch.empty(0))).species[0].tolist()
        self.holder = torch.classes.NNPOpsANISymmetryFunctions.Holder(
            self.num_species, float(symmFunc.Rcr), float(symmFunc.Rca), lists["EtaR"], lists["ShfR"], lists["EtaA"],
            lists["Zeta"], lists["ShfA"], lists["ShfZ"], [int(s) for s in species])
        self.triu_index = torch.tensor([0])      # kept for TorchScript compatibility with torchani.AEVComputer users

    @torch.jit.export
    def set_check_interval(self, interval: int) -> None:
        """Extension: verify the neighbour-buffer capacities (one host round trip) only on every ``interval``-th call
        instead of every call; 0 = only on the first.  Between checks an overflow goes unnoticed, exactly as inside a
        captured graph -- for production loops at known density (cf. ``check_errors`` of ``getNeighborPairs``)."""
        self.holder.set_check_interval(interval)

    def forward(self, species_positions: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """(species, positions[1, N, 3]) -> (species, aev[1, N, S*nR + S(S+1)/2*nA])"""
        species, positions = species_positions
        if species.shape[0] != 1:
            raise ValueError('Batched computation of molecules is not supported')
        if cell is not None:
            if pbc is None:
                raise ValueError('"pbc" has to be defined')
            else:
                pbc_: List[bool] = pbc.tolist()
                if pbc_ != [True, True, True]:
                    raise ValueError('Only fully periodic systems are supported, i.e. pbc = [True, True, True]')
        # the reference concatenates the two outputs of `operation` (SymmetryFunctions.py:120-122); `aev` has the kernels
        # write both parts into one [N, 1008] array in place (same values, no 4 KB/atom copy forward and backward)
        features = torch.ops.NNPOpsANISymmetryFunctions.aev(self.holder, positions[0], cell).unsqueeze(0)
        return species, features
