"""Host-side logic that needs no GPU: workload generators, argument validation, roofline arithmetic."""
import numpy as np
import pytest
import torch

from nnpops_amd import workloads


def test_ani2x_function_tables():
    rf, af = workloads.ani2x_functions()
    assert rf.shape == (16, 2) and af.shape == (32, 4)
    # order of the binding's loop nest (SymmetryFunctions.cpp:110-120): thetas fastest
    assert np.allclose(af[:4, 3], [(2 * i + 1) * np.pi / 8 for i in range(4)])
    assert np.allclose(af[::4, 1], [0.8 + 0.3375 * i for i in range(8)])
    assert 7 * 16 + 28 * 32 == 1008


def test_workloads_are_deterministic_and_well_formed():
    p1, s1, b1 = workloads.random_box(500, seed=3)
    p2, s2, b2 = workloads.random_box(500, seed=3)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2) and np.array_equal(b1, b2)
    assert abs(500 / b1[0, 0] ** 3 - 0.1) < 1e-3
    d = p1[:, None, :] - p1[None, :, :]
    d -= np.round(d / b1[0, 0]) * b1[0, 0]
    r = np.sqrt((d ** 2).sum(-1)) + np.eye(500) * 10
    assert r.min() > 0.5
    pw, sw, bw = workloads.water_box(50, seed=1)
    assert pw.shape == (150, 3) and sw.tolist()[:3] == [3, 0, 0]
    oh = np.linalg.norm(pw[1] - pw[0])
    assert abs(oh - 0.96) < 1e-4
    pc, sc = workloads.conformer(60, seed=2)
    dc = np.linalg.norm(pc[:, None] - pc[None], axis=-1) + np.eye(60) * 10
    assert dc.min() >= 0.9 - 1e-6 and sc.max() <= 3


def test_cpu_tensors_are_rejected_loudly():
    """There is no CPU fallback in the product path."""
    from nnpops_amd import capi
    with pytest.raises(ValueError, match="no CPU path"):
        capi.neighbor_pairs_forward(torch.zeros(4, 3), 1.0)
    with pytest.raises(ValueError, match="no CPU path"):
        capi._dev_f32(torch.zeros(4, 3), "positions")


def test_creating_an_evaluator_without_a_gpu_fails_loudly():
    from nnpops_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rf, af = workloads.ani2x_functions()
    with pytest.raises(capi.NNPOpsHipError):
        capi.AniSymmetryFunctions(7, 5.1, 3.5, np.zeros(5, np.int32), rf, af)


def test_species_grouped_nn_equals_per_atom_evaluation_cpu():
    """Host-side logic of the species-grouped BatchedNN (pure tensor algebra, device agnostic): grouping,
    padding of unequal layer widths, molecule batching and the model mean, against the obvious per-atom loop."""
    from types import SimpleNamespace
    from torch import nn
    from nnpops_amd.BatchedNN import _SpeciesGroupedNN

    class Conv:
        def __call__(self, x):
            return SimpleNamespace(species=x[0].unsqueeze(0))

    torch.manual_seed(0)
    widths = [(16, 12, 8), (14, 12, 8), (10, 8, 6)]
    ensemble = nn.ModuleList([nn.ModuleList([nn.Sequential(nn.Linear(20, a), nn.CELU(0.1), nn.Linear(a, b), nn.CELU(0.1),
                                                           nn.Linear(b, c), nn.CELU(0.1), nn.Linear(c, 1))
                                             for a, b, c in widths]) for _ in range(3)])
    species = torch.tensor([2, 0, 0, 1, 2, 2, 0])
    module = _SpeciesGroupedNN(Conv(), ensemble, species)
    assert module.group_sizes == [3, 1, 3] and module.atom_order.tolist() == [1, 2, 6, 3, 0, 4, 5]
    aev = torch.randn(2, 7, 20)
    got = module((species, aev)).energies
    want = torch.stack([sum(model[int(s)](aev[b, i]).sum() for model in ensemble for i, s in enumerate(species)) / 3
                        for b in range(2)])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_batched_nn_golden_weights_regenerate():
    """tests/golden/batched_nn_ref.npz was produced by the reference's BatchedLinear CPU op on seeded networks that the GPU
    tests rebuild from the seed: the rebuilt weights must be the ones the fixture was made with."""
    import os
    import numpy as np
    from nnpops_amd import workloads
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "batched_nn_ref.npz"))
    assert int(g["num_cases"]) == 3
    for k in range(3):
        model = workloads.torchani_like_model(n_models=int(g[f"c{k}_n_models"]), seed=int(g[f"c{k}_model_seed"]))
        checksum = sum(float(p.detach().double().abs().sum()) for net in model.neural_networks for p in net.parameters())
        assert abs(checksum - float(g[f"c{k}_weights_checksum"])) <= 1e-9 * checksum
        assert g[f"c{k}_aev"].shape == (1, len(g[f"c{k}_species"]), 1008) and g[f"c{k}_aev_grad"].shape == g[f"c{k}_aev"].shape
