"""TorchANIEnergyShifter -- self-energy sum precomputed for a fixed molecule
(reference src/pytorch/EnergyShifter.py:28-52)."""
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor


class SpeciesEnergies(NamedTuple):
    species: Tensor
    energies: Tensor


class TorchANIEnergyShifter(torch.nn.Module):

    def __init__(self, converter, shifter, atomicNumbers: Tensor) -> None:
        super().__init__()
        species = converter((atomicNumbers, torch.empty(0))).species
        self.register_buffer('self_energies', shifter.sae(species))

    def forward(self, species_energies: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> SpeciesEnergies:
        species, energies = species_energies
        return SpeciesEnergies(species, energies + self.self_energies)
