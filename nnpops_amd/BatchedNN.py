"""TorchANIBatchedNN -- the atomic networks of an ANI ensemble evaluated as batched linear layers
(reference src/pytorch/BatchedNN.py:37-122): per-atom, per-model weight matrices zero-padded to a
common shape, four BatchedLinear calls with CELU(0.1) in between, and one fused sum / mean.

Buffer names (``layer{0,2,4,6}_{weights,biases}``) and the ModuleList-of-one structure are kept so
that state dicts and TorchScript files stay interchangeable.
"""
from typing import List, NamedTuple, Optional, Tuple

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
        return self._grouped_forward(species_aev)

    def _grouped_forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
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


def _planes(w: Tensor) -> Tuple[Tensor, Tensor]:
    """fp32 [rows, cols] -> the two fp16 planes [rows, cols rounded up to 32] the split GEMM takes: w = hi + 2^-11 lo."""
    rows, cols = w.shape
    padded = torch.zeros((rows, (cols + 31) // 32 * 32), dtype=torch.float32, device=w.device)
    padded[:, :cols] = w
    hi = padded.half()
    lo = ((padded - hi.float()) * 2048.0).half()
    return hi, lo


def _pack_fragments(w: Tensor, permute: bool) -> Tensor:
    """fp32 [rows, cols] -> the fragment planes of csrc/mlp_fused.hip (what ``nnpops_mlp_pack`` writes; the GPU test compares
    the two): [row block][K step][plane][lane = r16 + 16 kg][8] fp16, ``w = hi + 2^-11 lo``; K order inside a step natural, or
    -- ``permute`` -- (4 kg + i | 16 + 4 kg + i - 4), the order in which a matrix-core accumulator hands over its rows."""
    rows, cols = w.shape
    nb, steps = (rows + 15) // 16, (cols + 31) // 32
    padded = torch.zeros((nb * 16, steps * 32), dtype=torch.float32, device=w.device)
    padded[:rows, :cols] = w
    kg = torch.arange(4, device=w.device).view(4, 1)
    i = torch.arange(8, device=w.device).view(1, 8)
    within = torch.where(i < 4, 4 * kg + i, 16 + 4 * kg + i - 4) if permute else 8 * kg + i          # [kg, i] -> k inside the step
    g = padded.view(nb, 16, steps, 32)[:, :, :, within.reshape(-1)]                                  # [rb, r16, s, (kg, i)]
    g = g.view(nb, 16, steps, 4, 8).permute(0, 2, 3, 1, 4)                                           # [rb, s, kg, r16, i]
    hi = g.half()
    lo = ((g - hi.float()) * 2048.0).half()
    return torch.stack([hi, lo], dim=2).reshape(-1)                                                  # [rb, s, plane, kg, r16, i]


def _up32(v: int) -> int:
    return (v + 31) // 32 * 32


class _FusedSpeciesNN(_SpeciesGroupedNN):
    """The species-grouped networks as TWO launches (``torch.ops.NNPOpsBatchedNN.FusedMLP``, csrc/mlp_fused.hip): a workgroup
    carries 64 atoms of one species and one ensemble member through all four layers -- and back through the small layers of
    the gradient -- with the activations in LDS / registers; a second launch forms dE/dAEV.  fp32 in and out, products as
    split-fp16 matrix instructions with fp32 accumulation.  Same buffers (and state dicts) as :class:`_SpeciesGroupedNN`; the
    packed operand planes are derived from them (non-persistent buffers, rebuilt when a state dict is loaded), each species at
    its OWN widths (the zero padding to the widest species that the reference's layout carries is stripped).  Frames with
    several molecules, CPU tensors, double precision and networks outside the kernels' shape (widths above 256, an input
    width that is not a multiple of 8, more than 8 species present) take the parent's path.

    Inside :class:`OptimizedTorchANI` the same kernels run behind ``torch.ops.NNPOpsANISymmetryFunctions.energy`` -- AEV and
    networks, forward and backward, as one autograd node (``fused_energy``)."""

    widths: List[int]
    live_blocks: List[int]
    act_scale_log2: int

    def __init__(self, converter, ensemble, atomicNumbers: Tensor):
        super().__init__(converter, ensemble, atomicNumbers)
        self.num_models = int(self.layer0_weights.shape[1])
        self.widths = []
        self.live_blocks = []
        self.fused_ok = True
        self.act_scale_log2 = 4
        self.register_buffer('atom_order32', self.atom_order.to(torch.int32), persistent=False)
        for name in ('mlp_planes', 'mlp_floats', 'live_planes'):
            self.register_buffer(name, torch.empty(0), persistent=False)
        for name in ('x_blocks', 'dead_blocks'):
            self.register_buffer(name, torch.empty(0, dtype=torch.int32), persistent=False)
        self._refresh_planes()

    @torch.jit.unused
    def set_live_blocks(self, blocks: List[int]) -> None:
        """(inside OptimizedTorchANI) the 16-column blocks of the AEV that can be non-zero for this molecule
        (TorchANISymmetryFunctions.live_column_blocks): fused_energy() then runs the networks over those columns only --
        first layer packed over them, the others neither read nor multiplied, their gradient zero (nnpops_hip.h: x_groups)."""
        F = int(self.layer0_weights.shape[3])
        self.live_blocks = sorted(int(b) for b in blocks) if F % 16 == 0 and len(blocks) < F // 16 else []
        self._refresh_planes()

    @staticmethod
    def _true_width(w: Tensor, b: Tensor) -> int:
        """Rows of a (zero padded) layer that are not identically zero, rounded up to 32: w [models, out, in], b [models, out, 1]."""
        used = (w.abs().amax(dim=(0, 2)) > 0) | (b.abs().amax(dim=(0, 2)) > 0)
        last = int(torch.nonzero(used).max()) + 1 if bool(used.any()) else 1
        return _up32(last)

    @torch.jit.unused
    def _refresh_planes(self) -> None:
        w0, w2, w4, w6 = self.layer0_weights.float(), self.layer2_weights.float(), self.layer4_weights.float(), self.layer6_weights.float()
        b0, b2, b4, b6 = self.layer0_biases.float(), self.layer2_biases.float(), self.layer4_biases.float(), self.layer6_biases.float()
        kinds, M, F = w0.shape[0], w0.shape[1], w0.shape[3]
        planes, floats, widths = [], [], []
        for k in range(kinds):
            h1, h2, h3 = self._true_width(w0[k], b0[k]), self._true_width(w2[k], b2[k]), self._true_width(w4[k], b4[k])
            h1, h2, h3 = min(h1, _up32(w0.shape[2])), min(h2, _up32(w2.shape[2])), min(h3, _up32(w4.shape[2]))
            widths += [h1, h2, h3]

            def cut(t: Tensor, rows: int, cols: int) -> Tensor:          # [M, out, in] -> [M, rows, cols], zero padded / trimmed
                out = torch.zeros((M, rows, cols), dtype=torch.float32, device=t.device)
                r, c = min(rows, t.shape[1]), min(cols, t.shape[2])
                out[:, :r, :c] = t[:, :r, :c]
                return out
            k0, k2, k4 = cut(w0[k], h1, F), cut(w2[k], h2, h1), cut(w4[k], h3, h2)
            planes += [_pack_fragments(k0[m], False) for m in range(M)]
            planes += [_pack_fragments(k2[m], True) for m in range(M)]
            planes += [_pack_fragments(k4[m], True) for m in range(M)]
            planes += [_pack_fragments(k4[m].t(), True) for m in range(M)]
            planes += [_pack_fragments(k2[m].t(), True) for m in range(M)]
            planes.append(_pack_fragments(k0.reshape(M * h1, F).t(), True))
            floats += [cut(b0[k], h1, 1).reshape(-1), cut(b2[k], h2, 1).reshape(-1), cut(b4[k], h3, 1).reshape(-1),
                       cut(w6[k][:, 0:1, :].transpose(1, 2), h3, 1).reshape(-1), b6[k].reshape(M, -1)[:, 0].reshape(-1)]
        self.mlp_planes = torch.cat(planes).contiguous()
        self.mlp_floats = torch.cat(floats).contiguous()
        self.widths = widths
        # the same networks over the live column blocks of the AEV only (set_live_blocks): w0 | w2 | w4 | w4t | w2t | w0t [| w0tm]
        dev = w0.device
        if self.live_blocks:
            cols = torch.tensor([16 * g + c for g in self.live_blocks for c in range(16)], dtype=torch.long, device=dev)
            Fl = int(cols.numel())
            live = []
            for k in range(kinds):
                h1, h2, h3 = widths[3 * k], widths[3 * k + 1], widths[3 * k + 2]

                def cut(t: Tensor, rows: int, ncols: int) -> Tensor:
                    out = torch.zeros((M, rows, ncols), dtype=torch.float32, device=t.device)
                    r, c = min(rows, t.shape[1]), min(ncols, t.shape[2])
                    out[:, :r, :c] = t[:, :r, :c]
                    return out
                k0, k2, k4 = cut(w0[k][:, :, cols], h1, Fl), cut(w2[k], h2, h1), cut(w4[k], h3, h2)
                live += [_pack_fragments(k0[m], False) for m in range(M)]
                live += [_pack_fragments(k2[m], True) for m in range(M)]
                live += [_pack_fragments(k4[m], True) for m in range(M)]
                live += [_pack_fragments(k4[m].t(), True) for m in range(M)]
                live += [_pack_fragments(k2[m].t(), True) for m in range(M)]
                live.append(_pack_fragments(k0.reshape(M * h1, Fl).t(), True))
                if Fl <= 256:
                    live += [_pack_fragments(k0[m].t(), True) for m in range(M)]
            self.live_planes = torch.cat(live).contiguous()
            self.x_blocks = torch.tensor(self.live_blocks, dtype=torch.int32, device=dev)
            self.dead_blocks = torch.tensor([g for g in range(F // 16) if g not in set(self.live_blocks)], dtype=torch.int32, device=dev)
        else:
            self.live_planes = torch.empty(0, device=dev)
            self.x_blocks = torch.empty(0, dtype=torch.int32, device=dev)
            self.dead_blocks = torch.empty(0, dtype=torch.int32, device=dev)
        # the kernels scale every activation by a power of two before the fp16 split (1/16: activations below ~1e6).  A crude
        # bound from the weights (AEV entries are sums of at most a few dozen terms <= 1) decides; networks that could
        # exceed it keep the library-GEMM path.  The backward operands take the same planes: |d3| <= |w6|,
        # |d2| <= |d3| ||W4||_1, |d1| <= |d2| ||W2||_1 (CELU' <= 1).
        bound = torch.full((1,), 64.0, device=w0.device)
        for w, b in ((w0, b0), (w2, b2), (w4, b4)):
            bound = (w.abs().sum(-1).amax() * bound + b.abs().amax()).reshape(1)
        back = w6.abs().amax().reshape(1)
        for w in (w4, w2):
            back = (w.abs().sum(-2).amax() * back).reshape(1)
        # The scale is 2^-k, k = act_scale_log2 of the ops: the smallest k >= 4 whose bound 62 500 * 2^k (fp16's largest
        # number, with margin) holds the crude bounds; 4 for networks of ordinary size, up to 12 (bounds below 2.56e8)
        # before the weights are left to the library-GEMM path.
        worst = max(float(bound), float(back)) if bool(torch.isfinite(bound).all()) and bool(torch.isfinite(back).all()) else float("inf")
        k = 4
        while k < 12 and worst >= 62500.0 * 2.0 ** k:
            k += 1
        self.act_scale_log2 = k
        self.fused_ok = (worst < 62500.0 * 2.0 ** k
                         and F % 8 == 0 and F <= 1024 and max(widths) <= 256 and kinds <= 8 and int(w6.shape[2]) == 1)
        holder = getattr(self, 'holder', None)         # (the one-node step keeps dE/dAEV between the steps, cleared once for ONE list of
        if holder is not None:                         #  live blocks: torch_binding.cpp, Holder::gradCache)
            holder.reset_gradient_cache()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self._refresh_planes()

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        species, aev = species_aev
        if (aev.shape[0] != 1 or not aev.is_cuda or aev.dtype != torch.float32 or self.mlp_planes.dtype != torch.float16
                or self.atom_order32.dtype != torch.int32 or not self.fused_ok):
            return self._grouped_forward(species_aev)
        total = torch.ops.NNPOpsBatchedNN.FusedMLP(aev[0].contiguous(), self.atom_order32, self.group_sizes, self.widths, self.num_models,
                                                   self.mlp_planes, self.mlp_floats, self.act_scale_log2)
        return SpeciesEnergies(species, total / self.num_models)

    def set_check_interval(self, interval: int) -> None:
        """(inside OptimizedTorchANI) how often the AEV holder behind fused_energy() verifies its neighbour capacities."""
        self.holder.set_check_interval(interval)

    def fused_energy(self, positions: Tensor, cell: Optional[Tensor], shift: Optional[Tensor] = None) -> Tensor:
        """AEV + networks of the whole frame as one autograd node (only inside OptimizedTorchANI, which hands over the AEV
        holder): positions [N, 3] or [1, N, 3] -> ensemble-mean energy [1]; with ``shift`` (the molecule's self energy, one
        float64 on the device) the energy comes back in float64, shifted as the reference's EnergyShifter does it."""
        if not self.fused_ok:        # (a state dict loaded since construction: the weights no longer bound the fp16 operands)
            raise RuntimeError("the loaded weights exceed the activation bound of the fused network kernels (BatchedNN.py: fused_ok); "
                               "build the model with OptimizedTorchANI(..., fused_step=False) or nn_layout='grouped'")
        if self.x_blocks.numel() > 0:
            return torch.ops.NNPOpsANISymmetryFunctions.energy(self.holder, positions, cell, self.atom_order32, self.group_sizes, self.widths,
                                                               self.num_models, self.live_planes, self.mlp_floats, shift, self.x_blocks,
                                                               self.dead_blocks, self.act_scale_log2)
        return torch.ops.NNPOpsANISymmetryFunctions.energy(self.holder, positions, cell, self.atom_order32, self.group_sizes, self.widths,
                                                           self.num_models, self.mlp_planes, self.mlp_floats, shift, None, None, self.act_scale_log2)

    def fused_energy_forces(self, positions: Tensor, cell: Optional[Tensor], shift: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """The same step outside autograd: (energy [1], forces = -dE/dpositions in the shape of ``positions``) from one call."""
        if not self.fused_ok:        # (a state dict loaded since construction: the weights no longer bound the fp16 operands)
            raise RuntimeError("the loaded weights exceed the activation bound of the fused network kernels (BatchedNN.py: fused_ok); "
                               "build the model with OptimizedTorchANI(..., fused_step=False) or nn_layout='grouped'")
        if self.x_blocks.numel() > 0:
            return torch.ops.NNPOpsANISymmetryFunctions.energy_forces(self.holder, positions, cell, self.atom_order32, self.group_sizes,
                                                                      self.widths, self.num_models, self.live_planes, self.mlp_floats, shift,
                                                                      self.x_blocks, self.dead_blocks, self.act_scale_log2)
        return torch.ops.NNPOpsANISymmetryFunctions.energy_forces(self.holder, positions, cell, self.atom_order32, self.group_sizes, self.widths,
                                                                  self.num_models, self.mlp_planes, self.mlp_floats, shift, None, None, self.act_scale_log2)


class _SplitGemmSpeciesNN(_SpeciesGroupedNN):
    """(Round-1/2 default, kept as ``layout='gemm'`` for comparison and for widths the fused kernels do not take.)  The
    species-grouped networks on the library's own GEMM (``torch.ops.NNPOpsBatchedNN.GroupedMLP``, csrc/batched_nn.hip): fp32 in and out, products as split-fp16 matrix instructions with fp32 accumulation, bias + CELU
    fused into the epilogues and CELU' into the input-gradient pass -- six launches per species and step instead of
    GEMM + elementwise kernels for every layer.  Same buffers (and state dicts) as :class:`_SpeciesGroupedNN`; the packed
    operand planes are derived from them (non-persistent buffers, rebuilt when a state dict is loaded).  Frames with
    several molecules, CPU tensors and double precision take the parent's path."""

    def __init__(self, converter, ensemble, atomicNumbers: Tensor):
        super().__init__(converter, ensemble, atomicNumbers)
        self.num_models = int(self.layer0_weights.shape[1])
        self.h1 = int(self.layer0_weights.shape[2])
        self.h2 = int(self.layer2_weights.shape[2])
        self.h3 = int(self.layer4_weights.shape[2])
        self.last_b: List[float] = []
        self.fused_ok = True
        self.register_buffer('atom_order32', self.atom_order.to(torch.int32), persistent=False)
        for name in ('fwd_hi', 'fwd_lo', 'bwd_hi', 'bwd_lo', 'packed_biases', 'last_w'):
            self.register_buffer(name, torch.empty(0), persistent=False)
        self._refresh_planes()

    @torch.jit.unused
    def _refresh_planes(self) -> None:
        w0, w2, w4, w6 = self.layer0_weights, self.layer2_weights, self.layer4_weights, self.layer6_weights
        kinds, models = w0.shape[0], w0.shape[1]
        fwd_hi, fwd_lo, bwd_hi, bwd_lo, biases, last_w, last_b = [], [], [], [], [], [], []
        for k in range(kinds):
            fwd = [w0[k].reshape(-1, w0.shape[3]), w2[k].reshape(-1, w2.shape[3]), w4[k].reshape(-1, w4.shape[3])]
            bwd = [w0[k].reshape(-1, w0.shape[3]).t().contiguous(),
                   torch.cat([w2[k][m].t() for m in range(models)], 0).contiguous(),
                   torch.cat([w4[k][m].t() for m in range(models)], 0).contiguous()]
            for mats, his, los in ((fwd, fwd_hi, fwd_lo), (bwd, bwd_hi, bwd_lo)):
                for w in mats:
                    hi, lo = _planes(w.float())
                    his.append(hi.reshape(-1))
                    los.append(lo.reshape(-1))
            biases += [self.layer0_biases[k].reshape(-1), self.layer2_biases[k].reshape(-1), self.layer4_biases[k].reshape(-1)]
            last_w.append(w6[k][:, 0, :].reshape(-1))
            last_b.append(float(self.layer6_biases[k].sum()))
        self.fwd_hi, self.fwd_lo = torch.cat(fwd_hi), torch.cat(fwd_lo)
        self.bwd_hi, self.bwd_lo = torch.cat(bwd_hi), torch.cat(bwd_lo)
        # the split GEMM scales its A operand by 1/16 before the fp16 split: activations must stay below ~1e6.  A crude
        # bound from the weights (AEV entries are sums of at most a few dozen terms <= 1) decides; networks that could
        # exceed it keep the library-GEMM path.
        bound = torch.full((1,), 64.0, device=w0.device)
        for w, b in ((w0, self.layer0_biases), (w2, self.layer2_biases), (w4, self.layer4_biases)):
            bound = (w.abs().sum(-1).amax() * bound + b.abs().amax()).reshape(1)
        # the backward operands take the same planes: |d3| <= |w6|, |d2| <= |d3| ||W4||_1, |d1| <= |d2| ||W2||_1 (CELU' <= 1)
        back = w6.abs().amax().reshape(1)
        for w in (w4, w2):
            back = (w.abs().sum(-2).amax() * back).reshape(1)
        self.fused_ok = (bool(torch.isfinite(bound).all()) and float(bound) < 1.0e6
                         and bool(torch.isfinite(back).all()) and float(back) < 1.0e6)
        self.packed_biases = torch.cat(biases).float().contiguous()
        self.last_w = torch.cat(last_w).float().contiguous()
        self.last_b = last_b

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self._refresh_planes()

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        species, aev = species_aev
        if (aev.shape[0] != 1 or not aev.is_cuda or aev.dtype != torch.float32 or self.fwd_hi.dtype != torch.float16
                or self.atom_order32.dtype != torch.int32 or not self.fused_ok):
            return self._grouped_forward(species_aev)
        # (the op reads the atoms of a species through atom_order32 and writes their gradients back through it: no gather
        #  into species order, no scatter back)
        per_atom = torch.ops.NNPOpsBatchedNN.GroupedMLP(aev[0].contiguous(), self.atom_order32, self.group_sizes, self.num_models,
                                                        self.h1, self.h2, self.h3,
                                                        self.fwd_hi, self.fwd_lo, self.bwd_hi, self.bwd_lo,
                                                        self.packed_biases, self.last_w, self.last_b)
        return SpeciesEnergies(species, per_atom.sum().reshape(1) / self.num_models)


class TorchANIBatchedNN(nn.ModuleList):
    """``layout='fused'`` (default): species-grouped networks as two launches of the fused kernels (csrc/mlp_fused.hip);
    ``'gemm'``: one split-fp16 GEMM per layer and species with fused activations (csrc/batched_nn.hip); ``'grouped'``: the same
    grouping on torch's library GEMMs; ``'reference'``: the reference's per-atom replicated weights through ``BatchedLinear``
    (interchange with reference state dicts / archives)."""

    def __init__(self, converter, ensemble, atomicNumbers: Tensor, layout: str = 'fused'):
        if layout not in ('fused', 'gemm', 'grouped', 'reference'):
            raise ValueError("layout must be 'fused', 'gemm', 'grouped' or 'reference'")
        impl = {'fused': _FusedSpeciesNN, 'gemm': _SplitGemmSpeciesNN, 'grouped': _SpeciesGroupedNN, 'reference': _BatchedNN}[layout]
        super().__init__([impl(converter, ensemble, atomicNumbers)])

    def forward(self, species_aev: Tuple[Tensor, Tensor]) -> SpeciesEnergies:
        return self[0].forward(species_aev)

    def fused_energy(self, positions: Tensor, cell: Optional[Tensor], shift: Optional[Tensor] = None) -> Tensor:
        return self[0].fused_energy(positions, cell, shift)

    def fused_energy_forces(self, positions: Tensor, cell: Optional[Tensor], shift: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        return self[0].fused_energy_forces(positions, cell, shift)

    def set_check_interval(self, interval: int) -> None:
        self[0].set_check_interval(interval)
