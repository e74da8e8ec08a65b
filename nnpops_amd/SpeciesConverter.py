"""TorchANISpeciesConverter -- species lookup precomputed for a fixed molecule
(reference src/pytorch/SpeciesConverter.py:26-44)."""
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor


class SpeciesCoordinates(NamedTuple):
    species: Tensor
    coordinates: Tensor


class TorchANISpeciesConverter(torch.nn.Module):

    def __init__(self, converter, atomicNumbers: Tensor) -> None:
        super().__init__()
        self.register_buffer('species', converter((atomicNumbers, torch.empty(0))).species)
        conv_tensor = getattr(converter, 'conv_tensor', None)
        self.conv_tensor = conv_tensor if conv_tensor is not None else torch.zeros(1, dtype=torch.long)

    def forward(self, species_coordinates: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> SpeciesCoordinates:
        _, coordinates = species_coordinates
        return SpeciesCoordinates(self.species, coordinates)
