"""TorchANIBatchedNN -- the atomic networks of an ANI ensemble evaluated as batched linear layers
(reference src/pytorch/BatchedNN.py:37-122): per-atom, per-model weight matrices zero-padded to a
common shape, four BatchedLinear calls with CELU(0.1) in between, and one fused sum / mean.

Buffer names (``layer{0,2,4,6}_{weights,biases}``) and the ModuleList-of-one structure are kept so
that state dicts and TorchScript files stay interchangeable.
"""
from typing import List, NamedTuple, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import torch_binding

torch_binding.load()


class SpeciesEnergies(NamedTuple):
    species: Tensor
    energies: Tensor


def _members(ensemble) -> List:
    """An Ensemble is a ModuleList of ANIModels; a single ANIModel stands for an ensemble of one."""
    return list(ensemble) if isinstance(ensemble, nn.ModuleList) else [ensemble]


def _networks_of(model) -> List[nn.Module]:
    """ANIModel is an ordered dict species-symbol -> Sequential; accept dicts and plain sequences too."""
    return list(model.values()) if hasattr(model, 'values') else list(model)


class _BatchedNN(nn.Module):

    def __init__(self, converter, ensemble, atomicNumbers: Tensor):
        super().__init__()
        species_list = converter((atomicNumbers, torch.empty(0))).species[0].tolist()
        models = [_networks_of(m) for m in _members(ensemble)]
        for ilayer in (0, 2, 4, 6):
            layers = [[model[s][ilayer] for s in species_list] for model in models]
            weights, biases = self.batchLinearLayers(layers)
            self.register_buffer(f'layer{ilayer}_weights', weights)
            self.register_buffer(f'layer{ilayer}_biases', biases)

    @staticmethod
    def batchLinearLayers(layers: List[List[nn.Linear]]) -> Tuple[Tensor, Tensor]:
        num_models, num_atoms = len(layers), len(layers[0])
        flat = [layer for sub in layers for layer in sub]
        max_out = max(layer.out_features for layer in flat)
        max_in = max(layer.in_features for layer in flat)
        weights = torch.zeros((1, num_atoms, num_models, max_out, max_in), dtype=torch.float32)
        biases = torch.zeros((1, num_atoms, num_models, max_out, 1), dtype=torch.float32)
        for imodel, sub in enumerate(layers):
            for iatom, layer in enumerate(sub):
                n_out, n_in = layer.weight.shape
                weights[0, iatom, imodel, :n_out, :n_in] = layer.weight.detach()
                biases[0, iatom, imodel, :n_out, 0] = layer.bias.detach()
        return weights, biases

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        species, aev = species_aev
        linear = torch.ops.NNPOpsBatchedNN.BatchedLinear
        # [mols, atoms, features] -> [mols, atoms, 1, features, 1]
        v = aev.unsqueeze(-2).unsqueeze(-1)
        v = F.celu(linear(v, self.layer0_weights, self.layer0_biases), alpha=0.1)
        v = F.celu(linear(v, self.layer2_weights, self.layer2_biases), alpha=0.1)
        v = F.celu(linear(v, self.layer4_weights, self.layer4_biases), alpha=0.1)
        v = linear(v, self.layer6_weights, self.layer6_biases)
        # sum over atoms (and the padded dims) and mean over models in ONE reduction, as the reference does
        energies = torch.sum(v, (1, 2, 3, 4)) / v.shape[2]
        return SpeciesEnergies(species, energies)


class TorchANIBatchedNN(nn.ModuleList):

    def __init__(self, converter, ensemble, atomicNumbers: Tensor):
        super().__init__([_BatchedNN(converter, ensemble, atomicNumbers)])

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        return self[0].forward(species_aev)
