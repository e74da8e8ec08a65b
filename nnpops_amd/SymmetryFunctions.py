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

    __jit_ignored_attributes__ = ["_ctor", "_species", "_batch_holders"]      # host-side state of forward_batch

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
        # what an additive batched evaluation needs to build its own holder (forward_batch)
        self._ctor = (self.num_species, float(symmFunc.Rcr), float(symmFunc.Rca), lists["EtaR"], lists["ShfR"], lists["EtaA"],
                      lists["Zeta"], lists["ShfA"], lists["ShfZ"])
        self._species = [int(s) for s in species]
        self._batch_holders = {}

    @torch.jit.unused
    def live_column_blocks(self) -> List[int]:
        """Extension: the 16-column blocks of this molecule's AEV rows that can be non-zero.  The radial block of a species and
        the angular block of a species pair (columns in the order of SymmetryFunctions.cpp:110-120) are identically zero when the
        molecule has no atom of that species -- nothing downstream needs to read, multiply or differentiate them
        (:class:`OptimizedTorchANI` hands the list to the fused networks)."""
        S = self.num_species
        nR = len(self._ctor[3]) * len(self._ctor[4])
        nA = len(self._ctor[5]) * len(self._ctor[6]) * len(self._ctor[7]) * len(self._ctor[8])
        present = sorted(set(self._species))
        live = [False] * (S * nR + S * (S + 1) // 2 * nA)
        for s in present:
            for c in range(s * nR, (s + 1) * nR):
                live[c] = True
        for i, a in enumerate(present):
            for b in present[i:]:
                bucket = a * S - a * (a - 1) // 2 + (b - a)          # upper-triangular row-major (CpuANISymmetryFunctions.cpp:39-43)
                for c in range(S * nR + bucket * nA, S * nR + (bucket + 1) * nA):
                    live[c] = True
        return [g for g in range((len(live) + 15) // 16) if any(live[16 * g:16 * g + 16])]

    @torch.jit.export
    def set_check_interval(self, interval: int) -> None:
        """Extension: verify the neighbour-buffer capacities (one host round trip) only on every ``interval``-th call
        instead of every call; 0 = only on the first.  Between checks an overflow goes unnoticed, exactly as inside a
        captured graph -- for production loops at known density (cf. ``check_errors`` of ``getNeighborPairs``)."""
        self.holder.set_check_interval(interval)

    @torch.jit.export
    def overflow_flag(self) -> Tensor:
        """Extension: int32[1] device tensor, non-zero when a forward since the last capacity check overflowed a neighbour buffer
        (a view of the word the kernels set: no copy, no synchronisation).  Inside a captured graph no check can run and an
        overflow leaves the AEV of the affected atoms incomplete; the flag is sticky, so ``bool(module.overflow_flag())`` after
        any number of replays -- or the next eager ``forward``, whose check then re-fits the buffers -- tells."""
        return self.holder.overflow_flag()

    @torch.jit.unused
    def forward_batch(self, species_positions: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
        """Extension (the reference rejects batches, SymmetryFunctions.py:110): B conformers of THE molecule this module was
        built for, non-periodic.  (species[B, N], positions[B, N, 3]) -> (species, aev[B, N, W]), differentiable w.r.t.
        positions.  All B x N atoms are evaluated by ONE batched holder (molecule offsets 0, N, 2N, ...; atoms of different
        conformers never interact) in one launch sequence -- not B launches of a one-molecule holder."""
        species, positions = species_positions
        if positions.dim() != 3 or positions.shape[1] != len(self._species) or positions.shape[2] != 3:
            raise ValueError(f'"positions" has to have the shape [batch, {len(self._species)}, 3]')
        batch = int(positions.shape[0])
        holder = self._batch_holders.get(batch)
        if holder is None:
            holder = torch.classes.NNPOpsANISymmetryFunctions.Holder(*self._ctor, self._species * batch)
            holder.set_molecules([k * len(self._species) for k in range(batch + 1)])
            self._batch_holders[batch] = holder
        flat = positions.reshape(batch * len(self._species), 3)
        features = torch.ops.NNPOpsANISymmetryFunctions.aev(holder, flat, None)
        return species, features.reshape(batch, len(self._species), -1)

    def check_arguments(self, species: Tensor, cell: Optional[Tensor], pbc: Optional[Tensor]) -> None:
        """The reference's argument errors (SymmetryFunctions.py:110-118)."""
        if species.shape[0] != 1:
            raise ValueError('Batched computation of molecules is not supported')
        if cell is not None:
            if pbc is None:
                raise ValueError('"pbc" has to be defined')
            else:
                pbc_: List[bool] = pbc.tolist()
                if pbc_ != [True, True, True]:
                    raise ValueError('Only fully periodic systems are supported, i.e. pbc = [True, True, True]')

    def forward(self, species_positions: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """(species, positions[1, N, 3]) -> (species, aev[1, N, S*nR + S(S+1)/2*nA])"""
        species, positions = species_positions
        self.check_arguments(species, cell, pbc)
        # the reference concatenates the two outputs of `operation` (SymmetryFunctions.py:120-122); `aev` has the kernels
        # write both parts into one [N, 1008] array in place (same values, no 4 KB/atom copy forward and backward)
        features = torch.ops.NNPOpsANISymmetryFunctions.aev(self.holder, positions[0], cell).unsqueeze(0)
        return species, features
