"""The split-fp16 GEMM behind the ANI atomic networks (batched_nn.hip) against float64 torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, scale, seed):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(shape, generator=g)).to(DEV)


@pytest.mark.parametrize("M,K,N", [(1, 32, 16), (64, 1008, 2048), (1333, 1008, 256), (667, 256, 192), (100, 160, 96), (70, 96, 1),
                                   (129, 33, 130), (2000, 2048, 1008)])
def test_gemm_matches_float64(M, K, N):
    from nnpops_amd import capi
    a = _rand((M, K), 1.5, 1)
    w = _rand((N, K), 1.0 / np.sqrt(K), 2)                   # a torch Linear weight [out][in]
    ref = (a.double() @ w.double().t())
    out = capi.gemm_split(a, capi.split_planes(w))
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    ref32 = (a @ w.t()).double()                               # what the library fp32 GEMM gives
    err32 = (ref32 - ref).abs().max().item()
    assert err <= 4e-6 * ref.abs().max().item() + 1e-7, (err, err32)
    assert err <= 3 * err32 + 1e-6


def test_epilogues_and_transpose():
    from nnpops_amd import capi
    M, K, N = 300, 224, 192
    a, w, b = _rand((M, K), 1.0, 3), _rand((N, K), 1.0 / np.sqrt(K), 4), _rand((N,), 0.3, 5)
    y = capi.gemm_split(a, capi.split_planes(w), bias=b)
    ref = torch.nn.functional.celu((a.double() @ w.double().t()) + b.double(), alpha=0.1)
    assert (y.double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    # backward step: d_in = (d_out @ W) * celu'(saved activation of the layer below)
    saved = torch.nn.functional.celu(_rand((M, K), 1.0, 6), alpha=0.1)
    d_out = _rand((M, N), 1.0, 7)
    d_in = capi.gemm_split(d_out, capi.split_planes(w, transpose=True), celu_of=saved)
    grad = torch.where(saved > 0, torch.ones_like(saved), saved / 0.1 + 1.0).double()
    ref = (d_out.double() @ w.double()) * grad
    assert (d_in.double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()


def test_large_inputs_are_scaled_into_range():
    from nnpops_amd import capi
    a = _rand((50, 128), 3.0e4, 8)                            # beyond fp16 without the scale
    w = _rand((64, 128), 0.1, 9)
    out = capi.gemm_split(a, capi.split_planes(w), a_scale=2.0 ** -6)
    ref = a.double() @ w.double().t()
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()


@pytest.mark.parametrize("layout", ["fused", "gemm"])
def test_fused_networks_match_the_library_gemm_path(layout):
    """TorchANIBatchedNN's default layout (the fused kernels of mlp_fused.hip) and the per-layer split-fp16 GEMMs ('gemm')
    against the same grouping on torch's fp32 library GEMMs: energies and AEV gradients, every ANI-2x species present,
    8 members."""
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    model = workloads.torchani_like_model(n_models=8, seed=11)
    pos, species, _ = workloads.water_box(200, seed=3)
    species = np.concatenate([species, [1, 2, 4, 6, 5, 1, 2, 2]]).astype(np.int32)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=DEV)
    fused = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout=layout).to(DEV)
    grouped = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="grouped").to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.randn(1, len(species), 1008, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4)).abs()
    a1, a2 = aev.clone().requires_grad_(True), aev.clone().requires_grad_(True)
    e1, e2 = fused((sp, a1)).energies, grouped((sp, a2)).energies
    (3.0 * e1.sum()).backward()
    (3.0 * e2.sum()).backward()
    # float64 reference of the same networks
    g64 = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="grouped").double().to(DEV)
    a3 = aev.double().clone().requires_grad_(True)
    e3 = g64((sp, a3)).energies
    (3.0 * e3.sum()).backward()
    assert abs(float(e1) - float(e3)) <= 1e-5 * abs(float(e3)) + 1e-4
    err_fused = float((a1.grad.double() - a3.grad).abs().max()); err_lib = float((a2.grad.double() - a3.grad).abs().max())
    scale = float(a3.grad.abs().max())
    assert err_fused <= 1e-5 * scale, (err_fused, err_lib, scale)
    assert err_fused <= 3 * err_lib + 1e-7 * scale
    scripted = torch.jit.script(fused)
    torch.testing.assert_close(scripted((sp, aev)).energies, e1.detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("layout", ["fused", "gemm"])
def test_fused_networks_edge_cases(layout):
    """A species with a single atom, species that do not occur at all, inference without gradients, and the paths the
    fused op hands back to the library GEMMs (two molecules in a frame, float64)."""
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    model = workloads.torchani_like_model(n_models=2, seed=21)
    species = np.array([0] * 37 + [3] * 5 + [6], dtype=np.int32)            # H, O and one Cl: four kinds never occur
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=DEV)
    fused = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout=layout).to(DEV)
    grouped = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="grouped").to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.randn(1, len(species), 1008, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)).abs()
    with torch.no_grad():
        e1, e2 = fused((sp, aev)).energies, grouped((sp, aev)).energies
    torch.testing.assert_close(e1, e2, rtol=1e-5, atol=1e-4)
    a1, a2 = aev.clone().requires_grad_(True), aev.clone().requires_grad_(True)
    fused((sp, a1)).energies.sum().backward()
    grouped((sp, a2)).energies.sum().backward()
    torch.testing.assert_close(a1.grad, a2.grad, rtol=1e-4, atol=1e-5 * float(a2.grad.abs().max()))
    # two molecules per frame and float64 inputs go through the library GEMMs
    two = torch.cat([aev, 0.5 * aev], 0)
    e_two = fused((sp.expand(2, -1), two)).energies
    assert e_two.shape == (2,)
    torch.testing.assert_close(e_two[0:1], e2, rtol=1e-5, atol=1e-4)
    e64 = fused.double()((sp, aev.double())).energies
    assert e64.dtype == torch.float64 and abs(float(e64) - float(e2)) <= 1e-5 * abs(float(e2)) + 1e-4


@pytest.mark.parametrize("layout,factor", [("fused", 3.0e5), ("gemm", 300.0)])
def test_networks_with_huge_weights_keep_the_library_gemms(layout, factor):
    """The fused paths carry activations through fp16 planes after a power-of-two scale: networks whose weights allow
    activations beyond what the largest scale holds (2^-4 for the split GEMM, 2^-12 for the fused kernels) are detected when
    the operand planes are built and evaluated by the library GEMMs instead."""
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    model = workloads.torchani_like_model(n_models=1, seed=31)
    for net in model.neural_networks[0].values():
        net[2].weight.data *= factor
    species = np.array([0, 0, 3, 1], dtype=np.int32)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=DEV)
    fused = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout=layout).to(DEV)
    grouped = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="grouped").to(DEV)
    assert not fused[0].fused_ok
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.rand(1, len(species), 1008, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    torch.testing.assert_close(fused((sp, aev)).energies, grouped((sp, aev)).energies, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("factor", [10.0, 100.0, 500.0])
def test_fused_networks_pick_the_activation_scale_from_the_weights(factor):
    """Weights whose crude activation bound exceeds what a 1/16 scale holds in fp16 (real ANI-2x members do) used to fall back to
    the library GEMMs, ~9x slower; the kernels now take the scale as an argument (nnpops_hip.h: act_scale_log2) and the module
    picks the smallest one that holds the bound.  Energies and AEV gradients against the float64 networks, the usual bars."""
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    model = workloads.torchani_like_model(n_models=2, seed=37)
    for ens in model.neural_networks:
        for net in ens.values():
            net[2].weight.data *= factor
    species = np.array([0, 0, 3, 1, 2, 0, 1, 1, 0, 3], dtype=np.int32)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=DEV)
    fused = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="fused").to(DEV)
    assert fused[0].fused_ok and 4 < fused[0].act_scale_log2 <= 12, fused[0].act_scale_log2
    exact = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout="grouped").to(DEV).double()
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.rand(1, len(species), 1008, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    a1, a2 = aev.clone().requires_grad_(True), aev.double().requires_grad_(True)
    e1, e2 = fused((sp, a1)).energies, exact((sp, a2)).energies
    assert e1.dtype == torch.float32
    assert abs(float(e1) - float(e2)) <= 1e-5 * abs(float(e2)) + 1e-5 * factor, (float(e1), float(e2))
    e1.sum().backward()
    e2.sum().backward()
    err = float((a1.grad.double() - a2.grad).abs().max()) / float(a2.grad.abs().max())
    assert err <= 1e-4, err


def test_gemm_row_maps():
    """a_rows / c_rows: the GEMM reads row m of A from a_rows[m] and writes row m of C to c_rows[m] (the atoms of a species
    are used where they lie)."""
    from nnpops_amd import capi
    a, w = _rand((500, 256), 1.0, 11), _rand((96, 256), 1.0 / 16, 12)
    pick = torch.randperm(500, generator=torch.Generator().manual_seed(13))[:321].to(torch.int32).to(DEV)
    planes = capi.split_planes(w)
    out = capi.gemm_split(a, planes, a_rows=pick, rows=321)
    ref = a[pick.long()].double() @ w.double().t()
    assert (out.double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    scattered = torch.zeros((500, 96), device=DEV)
    capi.gemm_split(a[pick.long()].contiguous(), planes, out=scattered, c_rows=pick)
    assert (scattered[pick.long()].double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()


def _golden_case(golden_dir, k):
    from nnpops_amd import workloads
    g = np.load(f"{golden_dir}/batched_nn_ref.npz")
    c = {name[len(f"c{k}_"):]: g[name] for name in g.files if name.startswith(f"c{k}_")}
    model = workloads.torchani_like_model(n_models=int(c["n_models"]), seed=int(c["model_seed"]))
    return c, model


@pytest.mark.parametrize("layout", ["fused", "gemm", "grouped", "reference"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_reference_batched_linear_goldens(golden_dir, k, layout):
    """Energies and dE/dAEV produced by the REFERENCE's BatchedLinear CPU op (src/pytorch/BatchedNN.cpp:30-42) in the
    reference's composition (BatchedNN.py:97-119), committed as tests/golden/batched_nn_ref.npz, against every layout of
    TorchANIBatchedNN: the species-grouped split-fp16 GEMMs (default), the same grouping on the library GEMMs, and the
    reference's per-atom replicated weights through torch.ops.NNPOpsBatchedNN.BatchedLinear."""
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    c, model = _golden_case(golden_dir, k)
    species = c["species"]
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]])
    nn = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers, layout=layout).to(DEV)
    sp = torch.tensor(species, device=DEV).unsqueeze(0)
    aev = torch.tensor(c["aev"], device=DEV).requires_grad_(True)
    energy = nn((sp, aev)).energies
    energy.sum().backward()
    scale = float(c["energy_scale"])
    assert abs(float(energy) - float(c["energy"][0])) <= 1e-5 * scale, (float(energy), float(c["energy"][0]), scale)
    gref = torch.tensor(c["aev_grad"], device=DEV)
    assert float((aev.grad - gref).abs().max()) <= 1e-4 * float(gref.abs().max())


def test_batched_linear_op_matches_reference_first_layer(golden_dir):
    """torch.ops.NNPOpsBatchedNN.BatchedLinear itself (same schema as the reference's op) on the packed first layer."""
    import NNPOps  # noqa: F401  (registers the op)
    from nnpops_amd import workloads
    from NNPOps.BatchedNN import TorchANIBatchedNN
    c, model = _golden_case(golden_dir, 0)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in c["species"]]])
    nn = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers, layout="reference").to(DEV)
    v = torch.tensor(c["aev"], device=DEV).unsqueeze(-2).unsqueeze(-1)
    y = torch.ops.NNPOpsBatchedNN.BatchedLinear(v, nn[0].layer0_weights, nn[0].layer0_biases)
    ref = torch.tensor(c["first_layer_atom0"], device=DEV)
    torch.testing.assert_close(y[0, 0, :, :, 0], ref, rtol=2e-5, atol=2e-5)


def test_python_packer_equals_the_c_abi_packer():
    """nnpops_amd/BatchedNN.py::_pack_fragments (torch ops; what a module runs at construction, on any device) writes the
    planes nnpops_mlp_pack (the C ABI's device kernel) writes, bit for bit."""
    from nnpops_amd.BatchedNN import _pack_fragments
    from nnpops_amd.capi import mlp_pack
    gen = torch.Generator().manual_seed(7)
    for rows, cols in ((192, 1008), (160, 256), (40, 72), (1008, 512)):
        w = torch.randn((rows, cols), generator=gen) / np.sqrt(cols)
        for permute in (False, True):
            assert torch.equal(_pack_fragments(w, permute), mlp_pack(w.to(DEV), rows, cols, permute=permute).cpu())
            assert torch.equal(_pack_fragments(w.t(), permute), mlp_pack(w.to(DEV), cols, rows, transpose=True, permute=permute).cpu())
