"""OptimizedTorchANI -- an ANI model with all four stages replaced
(reference src/pytorch/OptimizedTorchANI.py:33-54)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .BatchedNN import TorchANIBatchedNN, _FusedSpeciesNN
from .EnergyShifter import SpeciesEnergies, TorchANIEnergyShifter
from .SpeciesConverter import TorchANISpeciesConverter
from .SymmetryFunctions import TorchANISymmetryFunctions


class OptimizedTorchANI(torch.nn.Module):
    """The reference's four-module composition.  ``nn_layout`` (additive) picks the layout of the networks
    (:class:`TorchANIBatchedNN`); with the default ``'fused'`` and networks the fused kernels take, the instance becomes a
    :class:`FusedOptimizedTorchANI`: same modules, same state dict, same results, but AEV and networks run as ONE autograd
    node (``fused_step=False`` keeps the plain composition)."""

    def __init__(self, model, atomicNumbers: Tensor, nn_layout: str = 'fused', fused_step: bool = True, live_columns: bool = True) -> None:
        super().__init__()
        self.species_converter = TorchANISpeciesConverter(model.species_converter, atomicNumbers)
        self.aev_computer = TorchANISymmetryFunctions(model.species_converter, model.aev_computer, atomicNumbers)
        self.neural_networks = TorchANIBatchedNN(model.species_converter, model.neural_networks, atomicNumbers, layout=nn_layout)
        self.energy_shifter = TorchANIEnergyShifter(model.species_converter, model.energy_shifter, atomicNumbers)
        nets = self.neural_networks[0]
        if fused_step and isinstance(nets, _FusedSpeciesNN) and nets.fused_ok:
            nets.holder = self.aev_computer.holder        # the networks' fused_energy() drives the AEV kernels itself
            if live_columns:                               # ... and multiplies only the AEV blocks this molecule's species can fill
                nets.set_live_blocks(self.aev_computer.live_column_blocks())
            self.__class__ = FusedOptimizedTorchANI

    def forward(self, species_coordinates: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> SpeciesEnergies:
        species_coordinates = self.species_converter(species_coordinates)
        species_aevs = self.aev_computer(species_coordinates, cell, pbc)
        species_energies = self.neural_networks(species_aevs)
        return self.energy_shifter(species_energies)


class FusedOptimizedTorchANI(OptimizedTorchANI):
    """OptimizedTorchANI whose forward is ``torch.ops.NNPOpsANISymmetryFunctions.energy``: neighbour search + AEV + the four
    layers of every atomic network (+, when the positions require a gradient, the networks' input gradient and the AEV
    backward) issued back to back from one C++ call -- 7 kernel launches (the capacity check rides in the first network launch), one autograd node whose backward is a single
    multiplication -- instead of the ~45 launches the composition records.  Not constructed directly: ``OptimizedTorchANI(...)``
    turns into it.  Second derivatives are refused (use ``fused_step=False``)."""

    @torch.jit.export
    def set_check_interval(self, interval: int) -> None:
        """Extension (cf. TorchANISymmetryFunctions.set_check_interval): verify the neighbour-buffer capacities only on every
        ``interval``-th call (0: only on the first).  Sets it on the AEV module's holder and on the one the fused step drives
        (the same object until the module has been through torch.jit.save / load, two copies after)."""
        self.aev_computer.set_check_interval(interval)
        self.neural_networks.set_check_interval(interval)

    @torch.jit.export
    def energy_and_forces(self, species_coordinates: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                          pbc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """Extension: (energies [1], forces [1, N, 3] = -dE/dpositions) from ONE call, outside autograd -- for MD drivers that take
        the forces as a model output instead of differentiating the energy (no autograd node, no ``.sum()`` / ``backward()``
        around the step: the same kernels as ``forward`` + ``backward``, three small launches and the autograd engine's host time
        fewer).  Same arguments, checks and energies as ``forward``."""
        converted = self.species_converter(species_coordinates)
        species, positions = converted.species, converted.coordinates
        self.aev_computer.check_arguments(species, cell, pbc)
        shift = self.energy_shifter.self_energies
        if shift.dtype == torch.float64 and shift.numel() == 1 and shift.device == positions.device:
            return self.neural_networks.fused_energy_forces(positions, cell, shift)
        energy, forces = self.neural_networks.fused_energy_forces(positions, cell)
        return energy + self.energy_shifter.self_energies, forces

    def forward(self, species_coordinates: Tuple[Tensor, Tensor], cell: Optional[Tensor] = None,
                pbc: Optional[Tensor] = None) -> SpeciesEnergies:
        converted = self.species_converter(species_coordinates)
        species, positions = converted.species, converted.coordinates
        self.aev_computer.check_arguments(species, cell, pbc)
        # (positions go in as [1, N, 3] and the self-energy shift of EnergyShifter.py:52 is added by the kernel that takes the
        #  ensemble mean: no select / add / type-promotion kernels around the node, forward or backward)
        shift = self.energy_shifter.self_energies
        if shift.dtype == torch.float64 and shift.numel() == 1 and shift.device == positions.device:
            return SpeciesEnergies(species, self.neural_networks.fused_energy(positions, cell, shift))
        return self.energy_shifter((species, self.neural_networks.fused_energy(positions, cell)))
