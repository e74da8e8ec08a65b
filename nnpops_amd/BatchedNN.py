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
    """The reference's layout, kept bit-for-bit: one zero-padded weight matrix PER ATOM per model
    (``layer{k}_weights [1, atoms, models, out, in]``, reference BatchedNN.py:55-83) driven through the
    ``BatchedLinear`` op.  State dicts / TorchScript archives of the reference load into it unchanged.  It is
    a pure weight-streaming workload (21.6 GB of replicated weights for 2 000 atoms x 8 models): use it for
    interchange, and :class:`_SpeciesGroupedNN` (the default of ``TorchANIBatchedNN``) for speed."""

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


class _SpeciesGroupedNN(nn.Module):
    """Same function, MI355X layout: atoms are grouped by species once (the species of a Holder never change),
    so each layer is ONE batched GEMM per species, ``[models, out, in] x [in, atoms_of_species]`` -- the distinct
    weights (18.7 MB for ANI-2x x 8 models) instead of a per-atom replica of them (SURVEY.md s8(a) a15,
    s8(d) config 2).  The GEMMs go to the matrix cores through hipBLASLt/rocBLAS (plain library GEMMs).

    Buffers keep the reference's names, ``layer{k}_weights [kinds, models, out, in]`` and ``layer{k}_biases
    [kinds, models, out, 1]`` (zero-padded to the widest network of the layer, like the reference pads);
    ``load_state_dict`` also accepts the reference's per-atom tensors and compacts them."""

    group_sizes: List[int]

    def __init__(self, converter, ensemble, atomicNumbers: Tensor):
        super().__init__()
        species_list = converter((atomicNumbers, torch.empty(0))).species[0].tolist()
        kinds = sorted(set(species_list))
        kind_of_atom = torch.tensor([kinds.index(s) for s in species_list], dtype=torch.long)
        order = torch.sort(kind_of_atom, stable=True).indices
        self.register_buffer('atom_order', order)                          # atoms grouped by species, ascending inside
        self.register_buffer('first_atom_of_kind', torch.tensor([species_list.index(s) for s in kinds], dtype=torch.long))
        self.group_sizes = [int((kind_of_atom == k).sum()) for k in range(len(kinds))]
        self.num_atoms = len(species_list)
        models = [_networks_of(m) for m in _members(ensemble)]
        for ilayer in (0, 2, 4, 6):
            layers = [[model[s][ilayer] for s in kinds] for model in models]        # [model][kind]
            weights, biases = _BatchedNN.batchLinearLayers(layers)                   # [1, kinds, models, out, in]
            self.register_buffer(f'layer{ilayer}_weights', weights[0].contiguous())
            self.register_buffer(f'layer{ilayer}_biases', biases[0].contiguous())

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # a reference state dict holds [1, atoms, models, out, in]: keep one atom per species
        for ilayer in (0, 2, 4, 6):
            for what in ('weights', 'biases'):
                key = f'{prefix}layer{ilayer}_{what}'
                t = state_dict.get(key)
                if t is not None and t.dim() == 5 and t.shape[1] == self.num_atoms:
                    state_dict[key] = t[0].index_select(0, self.first_atom_of_kind.to(t.device)).contiguous()
        for extra in ('atom_order', 'first_atom_of_kind'):
            state_dict.setdefault(prefix + extra, getattr(self, extra))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        species, aev = species_aev
        mols = aev.shape[0]
        x = aev.index_select(1, self.atom_order)                            # [mols, atoms, features], grouped
        num_models = self.layer0_weights.shape[1]
        energies = torch.zeros(mols, dtype=aev.dtype, device=aev.device)
        first = 0
        for kind, n in enumerate(self.group_sizes):
            xs = x[:, first:first + n].reshape(mols * n, -1).t()            # [features, mols * n]
            first += n
            cols = mols * n
            # layer 0: every model reads the same AEVs -> ONE GEMM [models*out, features] x [features, cols]
            w0, b0 = self.layer0_weights[kind], self.layer0_biases[kind]
            v = torch.addmm(b0.reshape(-1, 1), w0.reshape(-1, w0.shape[2]), xs).reshape(num_models, w0.shape[1], cols)
            v = F.celu(v, alpha=0.1)                                         # [models, out, cols]
            v = F.celu(torch.baddbmm(self.layer2_biases[kind].expand(-1, -1, cols), self.layer2_weights[kind], v), alpha=0.1)
            v = F.celu(torch.baddbmm(self.layer4_biases[kind].expand(-1, -1, cols), self.layer4_weights[kind], v), alpha=0.1)
            v = torch.baddbmm(self.layer6_biases[kind].expand(-1, -1, cols), self.layer6_weights[kind], v)
            energies = energies + v.reshape(-1, mols, n).sum((0, 2))        # padded outputs are exactly zero
        return SpeciesEnergies(species, energies / num_models)


class TorchANIBatchedNN(nn.ModuleList):
    """``layout='grouped'`` (default): species-grouped GEMMs; ``layout='reference'``: the reference's per-atom
    replicated weights through ``BatchedLinear`` (interchange with reference state dicts / archives)."""

    def __init__(self, converter, ensemble, atomicNumbers: Tensor, layout: str = 'grouped'):
        if layout not in ('grouped', 'reference'):
            raise ValueError("layout must be 'grouped' or 'reference'")
        impl = _SpeciesGroupedNN if layout == 'grouped' else _BatchedNN
        super().__init__([impl(converter, ensemble, atomicNumbers)])

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        return self[0].forward(species_aev)
