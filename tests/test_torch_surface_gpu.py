"""The reference's PyTorch surface (torch.ops / torch.classes / NNPOps.* modules) on the HIP kernels.

torchani is not installed here, so the TorchANI objects the wrappers consume are replaced by small
stand-ins with the same attributes (the wrappers duck-type them).  Numerics are checked against the
oracle (AEV, CFConv) or against plain PyTorch (neighbours, BatchedNN); TorchScript script/save/load,
a non-default stream and graph capture mirror the reference's own Python tests
(src/pytorch/TestSymmetryFunctions.py:107-179, TestCFConv.py:100-140, neighbors/TestNeighbors.py:170-206,273-289).
"""
import io
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from nnpops_amd import workloads
from oracle import AniOracle, CFConvNeighborsOracle, CFConvOracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
Z_OF_SPECIES = [1, 6, 7, 8, 16, 9, 17]          # ANI-2x order H C N O S F Cl


class FakeConverter(nn.Module):
    """Stands in for torchani.nn.SpeciesConverter."""

    def __init__(self):
        super().__init__()
        conv = torch.full((120,), -1, dtype=torch.long)
        for s, z in enumerate(Z_OF_SPECIES):
            conv[z] = s
        self.register_buffer("conv_tensor", conv)

    def forward(self, inp):
        numbers, coords = inp
        return SimpleNamespace(species=self.conv_tensor.to(numbers.device)[numbers], coordinates=coords)


def fake_aev_computer():
    """Stands in for torchani.AEVComputer with the ANI-2x constants, tensors shaped as TorchANI shapes them."""
    c = workloads.ANI2X
    t = torch.tensor
    return SimpleNamespace(num_species=7, Rcr=c["Rcr"], Rca=c["Rca"],
                           EtaR=t(c["EtaR"]).view(-1, 1), ShfR=t(c["ShfR"]).view(1, -1),
                           EtaA=t(c["EtaA"]).view(-1, 1, 1, 1), Zeta=t(c["Zeta"]).view(1, -1, 1, 1),
                           ShfA=t(c["ShfA"]).view(1, 1, -1, 1), ShfZ=t(c["ShfZ"]).view(1, 1, 1, -1))


def _numbers(species):
    return torch.tensor([[Z_OF_SPECIES[s] for s in species]], device=DEV)


@pytest.mark.parametrize("periodic", [False, True])
def test_symmetry_functions_module(periodic):
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    if periodic:
        pos, species, box = workloads.water_box(100, seed=3)
    else:
        pos, species = workloads.conformer(73, seed=4)
        box = None
    rf, af = workloads.ani2x_functions()
    numbers = _numbers(species)
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), numbers.cpu()).to(DEV)
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    cell = torch.tensor(box, device=DEV) if periodic else None
    pbc = torch.tensor([True, True, True], device=DEV) if periodic else None
    sp, aev = module((torch.tensor(species, device=DEV).unsqueeze(0), tpos), cell, pbc)
    assert aev.shape == (1, len(species), 1008)
    w = torch.randn(aev.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    (aev * w).sum().backward()
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=periodic)
    r_ref, a_ref = oracle.forward(pos, box)
    ref = np.concatenate([r_ref, a_ref], axis=1)
    np.testing.assert_allclose(aev[0].detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    wn = w[0].cpu().numpy()
    g_ref = oracle.backward(np.ascontiguousarray(wn[:, :112]), np.ascontiguousarray(wn[:, 112:]))
    g = tpos.grad[0].cpu().numpy()
    assert np.abs(g - g_ref).max() <= 1e-4 * np.abs(g_ref).max()


def test_symmetry_functions_errors():
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    pos, species = workloads.conformer(10, seed=1)
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu())
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    with pytest.raises(ValueError, match="Batched"):
        module((sp.repeat(2, 1), tpos.repeat(2, 1, 1)))
    with pytest.raises(ValueError, match="pbc"):
        module((sp, tpos), torch.eye(3, device=DEV) * 30)
    with pytest.raises(ValueError, match="fully periodic"):
        module((sp, tpos), torch.eye(3, device=DEV) * 30, torch.tensor([True, True, False]))
    with pytest.raises(RuntimeError, match="float32"):
        module((sp, tpos.double()))
    with pytest.raises(RuntimeError, match="no CPU path"):
        module((sp.cpu(), tpos.cpu()))


def test_symmetry_functions_torchscript_roundtrip_and_stream():
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    pos, species = workloads.conformer(46, seed=8)
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu()).to(DEV)
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    ref = module((sp, tpos))[1]
    scripted = torch.jit.script(module)
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf)
    out = loaded((sp, tpos))[1]
    assert torch.equal(out, ref)
    # non-default stream (TestSymmetryFunctions.py:145-179)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        p2 = tpos.clone().requires_grad_(True)
        e = loaded((sp, p2))[1].sum()
        e.backward()
    stream.synchronize()
    p3 = tpos.clone().requires_grad_(True)
    module((sp, p3))[1].sum().backward()
    assert torch.allclose(p2.grad, p3.grad, rtol=1e-5, atol=1e-6)


def _cfconv_weights(G, W, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.3 * torch.randn(G, W, generator=g), 0.3 * torch.randn(W, generator=g), 0.2 * torch.randn(W, W, generator=g),
            0.3 * torch.randn(W, generator=g))


@pytest.mark.parametrize("activation", ["ssp", "tanh"])
def test_cfconv_module(activation):
    from NNPOps.CFConv import CFConv
    from NNPOps.CFConvNeighbors import CFConvNeighbors
    n, W, G, cutoff, sigma = 90, 32, 12, 4.0, 0.4
    pos, _ = workloads.conformer(n, seed=12)
    w1, b1, w2, b2 = _cfconv_weights(G, W, 5)
    x = torch.randn(n, W, generator=torch.Generator().manual_seed(6))
    gy = torch.randn(n, W, generator=torch.Generator().manual_seed(7))
    neighbors = CFConvNeighbors(cutoff)
    conv = CFConv(sigma, activation, w1, b1, w2, b2)
    tpos = torch.tensor(pos, device=DEV, requires_grad=True)
    tx = x.to(DEV).requires_grad_(True)
    neighbors.build(tpos)
    out = conv(neighbors, tpos, tx)
    assert out.shape == (n, W) and out.dtype == torch.float32 and out.device == tpos.device
    (out * gy.to(DEV)).sum().backward()
    # the binding hands the [G, W] buffer to a core that reads it as [W][G] (SURVEY.md s8b "weight layout trap")
    core_w1 = w1.contiguous().view(-1).view(W, G).numpy()
    onb = CFConvNeighborsOracle(n, cutoff)
    onb.build(pos)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, activation, core_w1, b1.numpy(), w2.numpy(), b2.numpy())
    y_ref = ocf.forward(onb, pos, x.numpy())
    xg_ref, pg_ref = ocf.backward(onb, pos, x.numpy(), gy.numpy())
    np.testing.assert_allclose(out.detach().cpu().numpy(), y_ref, rtol=2e-5, atol=2e-6 * np.abs(y_ref).max())
    np.testing.assert_allclose(tx.grad.cpu().numpy(), xg_ref, rtol=2e-5, atol=2e-6 * np.abs(xg_ref).max())
    assert np.abs(tpos.grad.cpu().numpy() - pg_ref).max() <= 1e-4 * np.abs(pg_ref).max()


def test_cfconv_torchscript_roundtrip():
    from NNPOps.CFConv import CFConv
    from NNPOps.CFConvNeighbors import CFConvNeighbors

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.neighbors = CFConvNeighbors(3.0)
            self.conv = CFConv(0.5, "ssp", *_cfconv_weights(6, 16, 1))

        def forward(self, positions, x):
            self.neighbors.build(positions)
            return self.conv(self.neighbors, positions, x)

    pos, _ = workloads.conformer(30, seed=2)
    tpos, x = torch.tensor(pos, device=DEV), torch.randn(30, 16, device=DEV)
    layer = Layer()
    ref = layer(tpos, x)
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(layer), buf)
    buf.seek(0)
    out = torch.jit.load(buf)(tpos, x)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_get_neighbor_pairs_op_values_and_grads(dtype):
    from NNPOps.neighbors import getNeighborPairs
    n = 64
    pos = (3 * torch.randn(n, 3, generator=torch.Generator().manual_seed(0))).to(dtype).to(DEV)
    ref_nb = torch.tril_indices(n, n, -1, device=DEV)
    p_ref = pos.clone().requires_grad_(True)
    d_ref = p_ref[ref_nb[0]] - p_ref[ref_nb[1]]
    r_ref = torch.linalg.norm(d_ref, dim=1)
    p = pos.clone().requires_grad_(True)
    nb, deltas, dist, npairs = getNeighborPairs(p, cutoff=1000.0)
    assert torch.equal(nb.long(), ref_nb) and int(npairs) == n * (n - 1) // 2
    assert torch.allclose(deltas, d_ref) and torch.allclose(dist, r_ref)
    (deltas.sum() + (dist ** 2).sum()).backward()
    (d_ref.sum() + (r_ref ** 2).sum()).backward()
    assert torch.allclose(p.grad, p_ref.grad, rtol=1e-3 if dtype == torch.float32 else 1e-9, atol=1e-3 if dtype == torch.float32 else 1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [300, 9000])
def test_get_neighbor_pairs_compacted_list_backward_is_a_gather(dtype, n, monkeypatch):
    """Round 6: with max_num_pairs > 0 and positions that require a gradient the op saves the list's transposed index and its
    backward is the owner-computes gather (no atomics); $NNPOPS_PAIRS_BACKWARD=fixed keeps the fixed-point sums.  Both against plain
    autograd through the same pairs, each bitwise reproducible; also inside a captured graph (forward + backward replayed)."""
    from NNPOps.neighbors import getNeighborPairs
    gen = torch.Generator().manual_seed(n)
    edge = (n / 0.1) ** (1.0 / 3.0)
    pos = (edge * torch.rand(n, 3, generator=gen)).to(dtype).to(DEV)
    box = (edge * torch.eye(3)).to(dtype).to(DEV)
    slots = 40 * n
    grads = {}
    for mode in ("gather", "fixed"):
        if mode == "fixed":
            monkeypatch.setenv("NNPOPS_PAIRS_BACKWARD", "fixed")
        p = pos.clone().requires_grad_(True)
        nb, deltas, dist, npairs = getNeighborPairs(p, cutoff=4.0, max_num_pairs=slots, box_vectors=box)
        assert 0 < int(npairs) < slots
        used = nb[0] >= 0
        loss = (deltas[used] * torch.arange(3, device=DEV, dtype=dtype)).sum() + (dist[used] ** 2).sum()
        loss.backward()
        grads[mode] = p.grad.clone()
        p2 = pos.clone().requires_grad_(True)
        getNeighborPairs(p2, cutoff=4.0, max_num_pairs=slots, box_vectors=box)
        nb2, deltas2, dist2, _ = getNeighborPairs(p2, cutoff=4.0, max_num_pairs=slots, box_vectors=box)
        ((deltas2[used] * torch.arange(3, device=DEV, dtype=dtype)).sum() + (dist2[used] ** 2).sum()).backward()
        assert torch.equal(p2.grad, grads[mode])                     # the same bits on every call
    # plain autograd through the same pairs (minimum image taken from the op's own deltas: d = p_i - p_j + shift, shift constant)
    p_ref = pos.clone().requires_grad_(True)
    i, j = nb[0][used].long(), nb[1][used].long()
    shift = (deltas[used] - (pos[i] - pos[j])).detach()
    d_ref = p_ref[i] - p_ref[j] + shift
    ((d_ref * torch.arange(3, device=DEV, dtype=dtype)).sum() + (torch.linalg.norm(d_ref, dim=1) ** 2).sum()).backward()
    tol = 1e-4 if dtype == torch.float32 else 1e-10
    scale = float(p_ref.grad.abs().max())
    for mode in grads:
        assert float((grads[mode] - p_ref.grad).abs().max()) <= tol * scale, mode


def test_get_neighbor_pairs_check_errors_and_graph_capture():
    from NNPOps.neighbors import getNeighborPairs
    pos = torch.zeros(4, 3, device=DEV)
    pos[:, 0] = torch.arange(4) * 0.1
    with pytest.raises(RuntimeError, match="Too many neighbor pairs"):
        getNeighborPairs(pos, cutoff=1.0, max_num_pairs=4, check_errors=True)
    nb, _, _, npairs = getNeighborPairs(pos, cutoff=1.0, max_num_pairs=4, check_errors=False)
    assert int(npairs) == 6 and nb.shape == (2, 4)
    # graph capture (TestNeighbors.py:170-206)
    big = torch.randn(200, 3, device=DEV) * 5
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            getNeighborPairs(big, cutoff=4.0, max_num_pairs=4000)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_nb, g_dl, g_ds, g_n = getNeighborPairs(big, cutoff=4.0, max_num_pairs=4000)
    big.copy_(torch.randn(200, 3, device=DEV) * 5)
    graph.replay()
    torch.cuda.synchronize()
    e_nb, e_dl, e_ds, e_n = getNeighborPairs(big, cutoff=4.0, max_num_pairs=4000)
    assert torch.equal(g_nb, e_nb) and int(g_n) == int(e_n)
    assert torch.allclose(g_ds, e_ds, equal_nan=True)


class FakeANIModel(nn.ModuleDict):
    """species symbol -> Sequential(Linear, CELU, Linear, CELU, Linear, CELU, Linear), like torchani.ANIModel"""


def _fake_ensemble(n_models, seed, widths):
    torch.manual_seed(seed)
    models = []
    for _ in range(n_models):
        nets = {}
        for sym, (h1, h2, h3) in widths.items():
            nets[sym] = nn.Sequential(nn.Linear(1008, h1), nn.CELU(0.1), nn.Linear(h1, h2), nn.CELU(0.1), nn.Linear(h2, h3),
                                      nn.CELU(0.1), nn.Linear(h3, 1))
        models.append(FakeANIModel(nets))
    return nn.ModuleList(models)


def test_batched_nn_and_optimized_torchani():
    """OptimizedTorchANI = species converter + HIP AEV + BatchedNN + shifter; compared with the plain
    per-species evaluation of the same random networks on the oracle's AEV (BASELINE config 2 shapes)."""
    from NNPOps import OptimizedTorchANI
    widths = {"H": (256, 192, 160), "C": (224, 192, 160), "N": (192, 160, 128), "O": (192, 160, 128),
              "S": (160, 128, 96), "F": (160, 128, 96), "Cl": (160, 128, 96)}
    ensemble = _fake_ensemble(3, 0, widths)
    sae = torch.tensor([-0.5, -38.0, -54.7, -75.2, -398.1, -99.8, -460.1], dtype=torch.float64)
    model = SimpleNamespace(species_converter=FakeConverter(), aev_computer=fake_aev_computer(), neural_networks=ensemble,
                            energy_shifter=SimpleNamespace(sae=lambda species: sae[species].sum(dim=1)))
    pos, species, box = workloads.water_box(40, seed=9)
    numbers = _numbers(species)
    opt = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)
    energy = opt((numbers, tpos), cell, pbc).energies
    energy.sum().backward()
    # plain evaluation on the oracle's AEV, float64
    rf, af = workloads.ani2x_functions()
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True)
    r_ref, a_ref = oracle.forward(pos, box)
    aev = torch.tensor(np.concatenate([r_ref, a_ref], axis=1), dtype=torch.float64, requires_grad=True)
    syms = list(widths)
    total = 0
    for net_dict in ensemble:
        for i, s in enumerate(species):
            total = total + net_dict[syms[s]].double()(aev[i]).sum()
    e_ref = total / len(ensemble) + sae[torch.tensor(species, dtype=torch.long)].sum()
    for net_dict in ensemble:
        net_dict.float()
    assert abs(float(energy) - float(e_ref)) <= 5e-6 * abs(float(e_ref)) + 1e-4
    e_ref.backward()
    g_aev = aev.grad.float().numpy()
    f_ref = oracle.backward(np.ascontiguousarray(g_aev[:, :112]), np.ascontiguousarray(g_aev[:, 112:]))
    f = tpos.grad[0].cpu().numpy()
    assert np.abs(f - f_ref).max() <= 1e-4 * np.abs(f_ref).max()          # north_star: 1e-4 on forces
    # buffers keep the reference's names
    names = {k for k, _ in opt.neural_networks.named_buffers()}
    assert {"0.layer0_weights", "0.layer6_biases"} <= names


def test_species_grouped_nn_matches_reference_layout_and_loads_its_state_dict():
    """The default BatchedNN layout (one batched GEMM per species) against the reference's per-atom replicated
    layout driven through torch.ops.NNPOpsBatchedNN.BatchedLinear: same energies, same AEV gradients; a state
    dict in the reference's shapes loads into the grouped module; both script."""
    from NNPOps.BatchedNN import TorchANIBatchedNN
    model = workloads.torchani_like_model(n_models=3, seed=5)
    pos, species, _ = workloads.water_box(30, seed=2)
    species = np.concatenate([species, [1, 2, 4, 6, 5, 1]]).astype(np.int32)      # every ANI-2x species present
    numbers = _numbers(species)
    grouped = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu()).to(DEV)
    reference = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="reference").to(DEV)
    assert grouped[0].layer0_weights.shape == (7, 3, 256, 1008)
    assert reference[0].layer0_weights.shape == (1, len(species), 3, 256, 1008)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.randn(1, len(species), 1008, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).abs()
    a1 = aev.clone().requires_grad_(True)
    a2 = aev.clone().requires_grad_(True)
    e1 = grouped((sp, a1)).energies
    e2 = reference((sp, a2)).energies
    e1.sum().backward()
    e2.sum().backward()
    assert e1.shape == e2.shape == (1,)
    torch.testing.assert_close(e1, e2, rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(a1.grad, a2.grad, rtol=1e-4, atol=1e-5 * float(a2.grad.abs().max()))
    # interchange: reference-shaped state dict -> grouped module
    blank = workloads.torchani_like_model(n_models=3, seed=99)
    other = TorchANIBatchedNN(blank.species_converter, blank.neural_networks, numbers.cpu()).to(DEV)
    assert not torch.allclose(other((sp, aev)).energies, e1.detach())
    other.load_state_dict(reference.state_dict())
    torch.testing.assert_close(other((sp, aev)).energies, e1.detach(), rtol=1e-6, atol=1e-6)
    # TorchScript
    scripted = torch.jit.script(grouped)
    torch.testing.assert_close(scripted((sp, aev)).energies, e1.detach(), rtol=1e-6, atol=1e-6)


def test_optimized_torchani_scripted_model_2000_atoms():
    """BASELINE config 2 at full size (2 001-atom periodic water box, 8 models): the scripted OptimizedTorchANI
    gives a finite energy, forces that sum to zero (translation invariance of the whole pipeline), and the
    same numbers as the eager module."""
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=8, seed=2)
    pos, species, box = workloads.water_box(667, seed=1)
    numbers = _numbers(species)
    opt = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    energy = opt((numbers, tpos), cell, pbc).energies
    energy.sum().backward()
    forces = -tpos.grad[0]
    assert torch.isfinite(energy).all() and torch.isfinite(forces).all()
    assert float(forces.double().sum(0).abs().max()) <= 1e-3 * float(forces.abs().max())
    scripted = torch.jit.script(opt)
    tpos2 = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    energy2 = scripted((numbers, tpos2), cell, pbc).energies
    energy2.sum().backward()
    torch.testing.assert_close(energy2, energy.detach(), rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(tpos2.grad, tpos.grad, rtol=1e-4, atol=1e-5 * float(tpos.grad.abs().max()))


@pytest.mark.parametrize("kind", ["cubic", "triclinic"])
def test_cfconv_module_periodic_extension(kind):
    """CFConvNeighbors.build(positions, box): periodic boundary conditions at the Python surface (the reference's
    surface is non-periodic, its core is not: src/schnet/CFConv.h:57) against the periodic oracle; scripted too."""
    from NNPOps.CFConv import CFConv
    from NNPOps.CFConvNeighbors import CFConvNeighbors
    W, G, cutoff, sigma = 32, 12, 4.0, 0.4
    if kind == "cubic":
        pos, _, box = workloads.random_box(400, seed=51)
    else:
        pos, _, box = workloads.triclinic_box(350, seed=52)
    n = len(pos)
    w1, b1, w2, b2 = _cfconv_weights(G, W, 8)
    x = torch.randn(n, W, generator=torch.Generator().manual_seed(9))
    gy = torch.randn(n, W, generator=torch.Generator().manual_seed(10))

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.neighbors = CFConvNeighbors(cutoff)
            self.conv = CFConv(sigma, "ssp", w1, b1, w2, b2)

        def forward(self, positions, box, x):
            self.neighbors.build(positions, box)
            return self.conv(self.neighbors, positions, x)

    layer = torch.jit.script(Layer())
    tpos = torch.tensor(pos, device=DEV, requires_grad=True)
    tx = x.to(DEV).requires_grad_(True)
    out = layer(tpos, torch.tensor(box, device=DEV), tx)
    (out * gy.to(DEV)).sum().backward()
    core_w1 = w1.contiguous().view(-1).view(W, G).numpy()
    onb = CFConvNeighborsOracle(n, cutoff, True)
    onb.build(pos, box)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, "ssp", core_w1, b1.numpy(), w2.numpy(), b2.numpy(), periodic=True)
    y_ref = ocf.forward(onb, pos, x.numpy(), box)
    xg_ref, pg_ref = ocf.backward(onb, pos, x.numpy(), gy.numpy(), box)
    np.testing.assert_allclose(out.detach().cpu().numpy(), y_ref, rtol=2e-5, atol=2e-6 * np.abs(y_ref).max())
    np.testing.assert_allclose(tx.grad.cpu().numpy(), xg_ref, rtol=2e-5, atol=2e-6 * np.abs(xg_ref).max())
    assert np.abs(tpos.grad.cpu().numpy() - pg_ref).max() <= 1e-4 * np.abs(pg_ref).max()
    # a holder is periodic or not for life
    with pytest.raises(RuntimeError, match="periodicity"):
        layer.neighbors.build(tpos.detach())


def test_optimized_torchani_step_replays_as_one_graph():
    """Energy + forces of OptimizedTorchANI captured once and replayed on NEW positions (what an MD loop does): the
    AEV holder does no host round trip while capturing, the cell histogram is re-zeroed by the kernels themselves,
    so the replay must equal an eager evaluation of the new frame."""
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=2, seed=3)
    pos, species, box = workloads.water_box(400, seed=7)              # 1 200 atoms: cell-grid path
    numbers = _numbers(species)
    opt = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True])     # host pbc: .tolist() cannot be captured
    static_pos = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)

    def eager(p):
        q = p.detach().clone().requires_grad_(True)
        e = opt((numbers, q), cell, pbc).energies
        return e.detach().clone(), torch.autograd.grad(e.sum(), q)[0]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                                             # calibrates neighbour capacities, warms allocators
            e = opt((numbers, static_pos), cell, pbc).energies
            torch.autograd.grad(e.sum(), static_pos)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_e = opt((numbers, static_pos), cell, pbc).energies
        g_f = torch.autograd.grad(g_e.sum(), static_pos)[0]
    rng = np.random.default_rng(1)
    for _ in range(3):
        new = (pos + rng.normal(0, 0.05, pos.shape)).astype(np.float32)
        with torch.no_grad():
            static_pos.copy_(torch.tensor(new, device=DEV).unsqueeze(0))
        graph.replay()
        torch.cuda.synchronize()
        e_ref, f_ref = eager(static_pos)
        torch.testing.assert_close(g_e, e_ref, rtol=1e-6, atol=1e-4)
        torch.testing.assert_close(g_f, f_ref, rtol=1e-4, atol=1e-5 * float(f_ref.abs().max()))


def test_overflow_inside_a_replayed_graph_is_observable():
    """No capacity check can run inside a captured graph.  The builders' overflow word is sticky and exposed as a device tensor
    (``overflow_flag()``, nnpops_hip.h: nnpops_ani_overflow_word): a replay on a frame that outgrows the fitted neighbour rows
    sets it -- readable at any time, no call into the library -- and the next eager forward reports and repairs the capacity."""
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    pos, species, box = workloads.water_box(400, seed=9)              # 1 200 atoms, rows fitted to liquid density
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu()).to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True])
    static_pos = torch.tensor(pos, device=DEV).unsqueeze(0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            module((sp, static_pos), cell, pbc)
    torch.cuda.current_stream().wait_stream(side)
    flag = module.overflow_flag()
    assert flag.dtype == torch.int32 and flag.is_cuda and int(flag) == 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_aev = module((sp, static_pos), cell, pbc)[1]
    graph.replay()
    torch.cuda.synchronize()
    assert int(flag) == 0                                              # the frame the capacities were fitted to
    # the same atoms pulled into an eighth of the volume around the centre of the box: rows several times longer than fitted
    centre = 0.5 * np.diag(box).astype(np.float32)
    dense = (centre + 0.5 * (pos - centre)).astype(np.float32)
    with torch.no_grad():
        static_pos.copy_(torch.tensor(dense, device=DEV).unsqueeze(0))
    graph.replay()
    torch.cuda.synchronize()
    assert int(flag) != 0                                              # set by the replay, nobody asked the library
    graph.replay()
    torch.cuda.synchronize()
    assert int(flag) != 0                                              # sticky
    eager = module((sp, static_pos), cell, pbc)[1]                     # the eager call checks, grows the buffers and evaluates again
    assert int(flag) == 0 and bool(torch.isfinite(eager).all())
    oracle = AniOracle(7, 5.1, 3.5, species, *workloads.ani2x_functions(), periodic=True, torchani=True)
    r_ref, a_ref = oracle.forward(dense, box)
    np.testing.assert_allclose(eager[0].cpu().numpy(), np.concatenate([r_ref, a_ref], axis=1), rtol=2e-5, atol=2e-6)


def test_capacity_check_interval_knob():
    """set_check_interval(k): results are unchanged (the check only verifies), scripted modules expose it, a negative
    interval is refused."""
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    pos, species, box = workloads.water_box(120, seed=5)
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu()).to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    tpos = torch.tensor(pos, device=DEV).unsqueeze(0)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True])
    ref = module((sp, tpos), cell, pbc)[1].clone()
    scripted = torch.jit.script(module)
    for interval in (0, 3, 1):
        scripted.set_check_interval(interval)
        for _ in range(4):
            assert torch.equal(scripted((sp, tpos), cell, pbc)[1], ref)
    with pytest.raises(RuntimeError, match="interval"):
        module.set_check_interval(-1)


def test_forward_batch_evaluates_conformers_in_one_holder():
    """Additive API on the torch surface: B conformers of one molecule through ONE batched holder
    (Holder.set_molecules) must equal B single-molecule evaluations of the reference-shaped forward(), values and position
    gradients; the reference-shaped forward() still rejects batches with the reference's ValueError."""
    from nnpops_amd.SymmetryFunctions import TorchANISymmetryFunctions
    base, species = workloads.conformer(37, seed=71)
    rng = np.random.default_rng(72)
    confs = np.stack([base + rng.normal(0, 0.08, base.shape).astype(np.float32) for _ in range(5)])
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu()).to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    tpos = torch.tensor(confs, device=DEV, requires_grad=True)
    _, aev = module.forward_batch((sp.expand(5, -1), tpos))
    assert aev.shape == (5, 37, 1008)
    w = torch.randn(aev.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    (aev * w).sum().backward()
    for b in range(5):
        one = torch.tensor(confs[b:b + 1], device=DEV, requires_grad=True)
        _, ref = module((sp, one))
        torch.testing.assert_close(aev[b:b + 1], ref, rtol=1e-6, atol=1e-7)
        (ref * w[b:b + 1]).sum().backward()
        torch.testing.assert_close(tpos.grad[b:b + 1], one.grad, rtol=1e-5, atol=1e-6 * float(one.grad.abs().max()))
    with pytest.raises(ValueError, match="Batched computation of molecules is not supported"):
        module((sp.expand(5, -1), tpos))


def test_fused_optimized_torchani_is_one_autograd_node_and_equals_the_composition():
    """OptimizedTorchANI with the default network layout becomes a FusedOptimizedTorchANI: AEV + networks behind
    torch.ops.NNPOpsANISymmetryFunctions.energy.  Same energy and forces as the four-module composition
    (fused_step=False) on the same kernels; energy-only calls run no backward kernels; a recorded backward pass is refused;
    state dicts are interchangeable; the scripted module saves, loads and agrees."""
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=3, seed=12, self_energies=[-0.5, -38.0, -54.7, -75.2, -398.1, -99.8, -460.1])
    pos, species, box = workloads.water_box(150, seed=21)
    species = species.copy()
    species[[5, 17, 40]] = [1, 2, 4]                                      # a C, an N and an S among the waters: four kinds, one tiny
    numbers = _numbers(species)
    fused = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    plain = OptimizedTorchANI(model, numbers.cpu(), fused_step=False).to(DEV)
    assert type(fused).__name__ == "FusedOptimizedTorchANI" and type(plain).__name__ == "OptimizedTorchANI"
    assert isinstance(fused, OptimizedTorchANI)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)

    def run(module, scale=1.0):
        p = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
        e = module((numbers, p), cell, pbc).energies
        (scale * e.sum()).backward()
        return e.detach(), p.grad.detach()

    e1, f1 = run(fused, 2.5)                                               # (a non-trivial upstream gradient)
    e2, f2 = run(plain, 2.5)
    assert e1.dtype == e2.dtype == torch.float64 and e1.shape == (1,)
    torch.testing.assert_close(e1, e2, rtol=1e-7, atol=2e-4)
    torch.testing.assert_close(f1, f2, rtol=1e-4, atol=2e-5 * float(f2.abs().max()))
    # the graph of the fused module is ONE custom node between the positions and the shifter's addition
    p = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    e = fused((numbers, p), cell, pbc).energies
    names, fn = [], e.grad_fn
    while fn is not None:
        names.append(fn.name())
        fn = fn.next_functions[0][0] if fn.next_functions else None
    # energy node <- leaf: nothing else was recorded (the positions go in as [1, N, 3], the self-energy shift is added by the
    # kernel that takes the ensemble mean)
    assert len(names) == 2 and "EnergyFunction" in names[0] and "AccumulateGrad" in names[1], names
    # ... and that shift is the reference's `energies + self_energies` to the bit: the float32 mean, promoted, plus the float64 buffer
    with torch.no_grad():
        unshifted = fused.neural_networks.fused_energy(p[0], cell)
        assert unshifted.dtype == torch.float32
        assert torch.equal(e.detach(), unshifted + fused.energy_shifter.self_energies)
    # a float32 upstream gradient (the energy cast down before the loss) takes the same node
    p32 = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    (2.5 * fused((numbers, p32), cell, pbc).energies.float().sum()).backward()
    torch.testing.assert_close(p32.grad, f1, rtol=1e-6, atol=0.0)
    assert p32.grad.shape == (1, len(species), 3)
    with pytest.raises(RuntimeError, match="second derivatives are not implemented"):
        torch.autograd.grad(e.sum(), p, create_graph=True)
    with torch.no_grad():                                                 # energy only
        torch.testing.assert_close(fused((numbers, p), cell, pbc).energies, e1, rtol=1e-7, atol=1e-6)
    # the reference's argument errors survive the fusion
    with pytest.raises(ValueError, match='"pbc" has to be defined'):
        fused((numbers, p), cell, None)
    # state dicts interchange, scripting round-trips
    other = OptimizedTorchANI(workloads.torchani_like_model(n_models=3, seed=99), numbers.cpu()).to(DEV)
    other.load_state_dict(plain.state_dict())
    torch.testing.assert_close(run(other, 2.5)[1], f1, rtol=1e-6, atol=1e-7 * float(f1.abs().max()))
    scripted = torch.jit.script(fused)
    buffer = io.BytesIO()
    torch.jit.save(scripted, buffer)
    buffer.seek(0)
    loaded = torch.jit.load(buffer, map_location=DEV)
    loaded.set_check_interval(0)                                          # (exported: reaches both copies of the holder)
    loaded.set_check_interval(1)
    e3, f3 = run(loaded, 2.5)
    torch.testing.assert_close(e3, e1, rtol=1e-7, atol=1e-6)
    torch.testing.assert_close(f3, f1, rtol=1e-6, atol=1e-7 * float(f1.abs().max()))


def test_fused_step_regrows_its_neighbour_buffers_behind_the_deferred_check():
    """The one-node step reads the capacity check of its AEV holder AFTER it has launched the networks and the backward pass
    (torch_binding.cpp: forwardImpl(defer_check) / finishDeferredCheck).  A frame dense enough to overflow the fitted rows
    must therefore be detected there, the buffers grown and the whole step issued again: same energy and forces as a fresh
    four-module composition on that frame."""
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=2, seed=4)
    pos, species, box = workloads.water_box(120, seed=8)
    numbers = _numbers(species)
    fused = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    assert type(fused).__name__ == "FusedOptimizedTorchANI"
    pbc = torch.tensor([True, True, True], device=DEV)

    def run(module, scale):
        p = torch.tensor(pos * np.float32(scale), device=DEV).unsqueeze(0).requires_grad_(True)
        e = module((numbers, p), torch.tensor(box * np.float32(scale), device=DEV), pbc).energies
        e.sum().backward()
        return e.detach(), p.grad.detach()

    for _ in range(3):                                     # capacities fitted to this density, checks deferred from now on
        run(fused, 1.0)
    e1, f1 = run(fused, 0.72)                              # 2.7 x denser: overflows the fitted rows
    plain = OptimizedTorchANI(model, numbers.cpu(), fused_step=False).to(DEV)
    e2, f2 = run(plain, 0.72)
    torch.testing.assert_close(e1, e2, rtol=1e-7, atol=2e-4)
    torch.testing.assert_close(f1, f2, rtol=1e-4, atol=2e-5 * float(f2.abs().max()))
    e3, f3 = run(fused, 0.72)                              # and once more, now without growth: bitwise the same
    assert torch.equal(e1, e3) and torch.equal(f1, f3)


def test_energy_and_forces_in_one_call_equals_forward_plus_backward():
    """FusedOptimizedTorchANI.energy_and_forces (additive): the energy of forward() to the bit and the forces -dE/dpositions to the
    bit (the same launches with the sign folded into the networks' input gradient), no autograd graph recorded; also through
    the scripted module, and for a frame whose live AEV columns exceed 256 (five species: the separate gradient launch)."""
    from NNPOps import OptimizedTorchANI
    for seed, swaps in ((3, {}), (5, {4: 1, 9: 2, 33: 4})):
        model = workloads.torchani_like_model(n_models=3, seed=seed, self_energies=[-0.5, -38.0, -54.7, -75.2, -398.1, -99.8, -460.1])
        pos, species, box = workloads.water_box(110, seed=seed)
        species = species.copy()
        for at, sp in swaps.items():
            species[at] = sp
        numbers = _numbers(species)
        module = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
        cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)
        p = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
        e = module((numbers, p), cell, pbc).energies
        e.sum().backward()
        e2, f2 = module.energy_and_forces((numbers, p.detach()), cell, pbc)
        assert e2.grad_fn is None and f2.grad_fn is None and f2.shape == p.shape and e2.dtype == torch.float64
        assert torch.equal(e2, e.detach()) and torch.equal(f2, -p.grad)
        e3, f3 = torch.jit.script(module).energy_and_forces((numbers, p.detach()), cell, pbc)
        assert torch.equal(e3, e2) and torch.equal(f3, f2)
        with pytest.raises(ValueError, match='"pbc" has to be defined'):
            module.energy_and_forces((numbers, p.detach()), cell, None)


def test_gradient_buffer_kept_between_steps_follows_the_live_blocks():
    """The one-node step keeps dE/dAEV between the steps and clears it ONCE (the networks write the live column blocks only, the others
    stay zero: torch_binding.cpp, Holder::gradCache).  Forces are the same bits step after step; when the list of live blocks changes
    (set_live_blocks with more blocks than the species need) the buffer is made again and the forces do not change beyond rounding
    (the extra blocks multiply AEV columns that are identically zero); a frame with other positions in between leaves nothing behind."""
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=2, seed=12)
    pos, species, box = workloads.water_box(90, seed=4)
    numbers = _numbers(species)
    module = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)

    def forces(xyz):
        p = torch.tensor(xyz, device=DEV).unsqueeze(0).requires_grad_(True)
        module((numbers, p), cell, pbc).energies.backward()
        return p.grad.clone()

    f0 = forces(pos)
    moved = pos + 0.05 * np.random.default_rng(1).standard_normal(pos.shape).astype(np.float32)
    f_moved = forces(moved)
    assert torch.equal(forces(pos), f0) and not torch.equal(f_moved, f0)
    nets = module.neural_networks[0]
    live = list(nets.live_blocks)
    assert 0 < len(live) < 63
    wider = sorted(set(live) | {b for b in range(63) if b % 5 == 0})
    nets.set_live_blocks(wider)
    f_wide = forces(pos)
    assert float((f_wide - f0).abs().max()) <= 1e-6 * float(f0.abs().max())
    nets.set_live_blocks(live)
    assert torch.equal(forces(pos), f0)
    plain = OptimizedTorchANI(model, numbers.cpu(), fused_step=False).to(DEV)          # the four-module composition
    p = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
    plain((numbers, p), cell, pbc).energies.backward()
    assert float((p.grad - f0).abs().max()) <= 1e-5 * float(f0.abs().max())


def test_overflow_flag_outlives_its_module():
    """overflow_flag() is a view of a word the AEV handle owns; the tensor keeps the Holder (and the handle) alive, so reading it after
    the module is gone is a read of live memory (it used to dangle)."""
    import gc
    from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
    pos, species, box = workloads.water_box(40, seed=2)
    module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), _numbers(species).cpu()).to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True])
    module((sp, torch.tensor(pos, device=DEV).unsqueeze(0)), cell, pbc)
    flag = module.overflow_flag()
    del module
    gc.collect()
    torch.cuda.synchronize()
    junk = [torch.full((1 << 16,), 7, dtype=torch.int32, device=DEV) for _ in range(8)]      # (would land on freed device memory)
    torch.cuda.synchronize()
    assert int(flag) == 0 and len(junk) == 8


def test_fused_step_with_a_larger_activation_scale_survives_jit_save_and_load():
    """Weights beyond the 1/16 activation bound (first layers x 100): OptimizedTorchANI stays the one-node step with the scale the
    bound asks for (act_scale_log2 > 4), the scripted module saved and loaded gives the same bits, and energy / forces agree with the
    composition on the library GEMMs (1e-5 / 1e-4)."""
    import io
    from NNPOps import OptimizedTorchANI
    model = workloads.torchani_like_model(n_models=2, seed=5)
    for ens in model.neural_networks:
        for net in ens.values():
            net[0].weight.data *= 100.0
    pos, species, box = workloads.water_box(60, seed=2)
    numbers = _numbers(species)
    module = OptimizedTorchANI(model, numbers.cpu()).to(DEV)
    assert type(module).__name__ == "FusedOptimizedTorchANI" and 4 < module.neural_networks[0].act_scale_log2 <= 12
    cell, pbc = torch.tensor(box, device=DEV), torch.tensor([True, True, True], device=DEV)

    def run(m):
        p = torch.tensor(pos, device=DEV).unsqueeze(0).requires_grad_(True)
        e = m((numbers, p), cell, pbc).energies
        e.backward()
        return e.detach().clone(), p.grad.clone()

    e0, f0 = run(module)
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(module), buf)
    buf.seek(0)
    e1, f1 = run(torch.jit.load(buf, map_location=DEV))
    assert torch.equal(e0, e1) and torch.equal(f0, f1)
    e2, f2 = run(OptimizedTorchANI(model, numbers.cpu(), nn_layout="grouped").to(DEV))
    assert abs(float(e0) - float(e2)) <= 1e-5 * abs(float(e2))
    assert float((f0 - f2).abs().max()) <= 1e-4 * float(f2.abs().max())
