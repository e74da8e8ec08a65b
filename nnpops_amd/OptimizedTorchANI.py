"""OptimizedTorchANI -- an ANI model with all four stages replaced
(reference src/pytorch/OptimizedTorchANI.py:33-54)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .BatchedNN import TorchANIBatchedNN
from .EnergyShifter import SpeciesEnergies, TorchANIEnergyShifter
from .SpeciesConverter import TorchANISpeciesConverter
from .SymmetryFunctions import TorchANISymmetryFunctions


class OptimizedTorchANI(torch.nn.Module):

    def __init__(self, model, atomicNumbers: Tensor) -> None:
        super().__init__()
        self.species_converter = TorchANISpeciesConverter(model.species_converter, atomicNumbers)
        self.aev_computer = TorchANISymmetryFunctions(model.species_converter, model.aev_computer, atomicNumbers)
        self.neural_networks = TorchANIBatchedNN(model.species_converter, model.neural_networks, atomicNumbers)
        self.energy_shifter = TorchANIEnergyShifter(model.species_converter, model.energy_shifter, atomicNumbers)

    def forward(self, species_coordinates: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> SpeciesEnergies:
        species_coordinates = self.species_converter(species_coordinates)
        species_aevs = self.aev_computer(species_coordinates, cell, pbc)
        species_energies = self.neural_networks(species_aevs)
        return self.energy_shifter(species_energies)
