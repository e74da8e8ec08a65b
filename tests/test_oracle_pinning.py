"""Pin the oracle (CPU, no GPU needed): against the golden vectors the reference's own tests hold,
against the reference's own sources compiled in place (oracle/_ref, when present), and against
fixtures produced by the reference CPU torch op (tests/golden/neighbors_ref.npz)."""
import numpy as np
import pytest

import oracle
from nnpops_amd import workloads
from oracle import (AniOracle, CFConvNeighborsOracle, CFConvOracle, neighbor_pairs_backward_oracle,
                    neighbor_pairs_oracle)

needs_ref = pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


# ---------------------------------------------------------------- ANI
@pytest.mark.parametrize("tag", ["nonperiodic", "periodic", "triclinic"])
def test_ani_oracle_vs_torchani_golden(golden_dir, tag):
    """src/ani/TestANISymmetryFunctions.h:111-252 (values computed with TorchANI)."""
    g = np.load(f"{golden_dir}/ani_water18.npz")
    box = g[f"{tag}_box"] if tag != "nonperiodic" else None
    o = AniOracle(2, 4.5, 3.5, g["species"], g["radial_functions"], g["angular_functions"], periodic=box is not None)
    r, a = o.forward(g["positions"], box)
    # the reference asserts |diff| <= 1e-4 OR rel <= 1e-3 (:8-12,100,102); we require both-sided closeness
    np.testing.assert_allclose(r, g[f"{tag}_radial"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, g[f"{tag}_angular"], rtol=1e-5, atol=1e-6)


def _fd_check(forward, backward_dot, pos, step=1e-3):
    """Finite-difference check in the spirit of validateDerivatives (TestANISymmetryFunctions.h:14-58)."""
    grad = backward_dot()
    norm = np.linalg.norm(grad)
    d = grad / norm * step
    e_plus, e_minus = forward(pos + d), forward(pos - d)
    est = (e_plus - e_minus) / (2 * step)
    assert abs(est - norm) <= 5e-3 * norm + 1e-5


@pytest.mark.parametrize("torchani", [True, False])
def test_ani_oracle_gradient_is_consistent(golden_dir, torchani):
    g = np.load(f"{golden_dir}/ani_water18.npz")
    o = AniOracle(2, 4.5, 3.5, g["species"], g["radial_functions"], g["angular_functions"], torchani=torchani)
    rng = np.random.default_rng(0)
    r0, a0 = o.forward(g["positions"])
    wr, wa = rng.standard_normal(r0.shape).astype(np.float32), rng.standard_normal(a0.shape).astype(np.float32)

    def energy(p):
        r, a = o.forward(p.astype(np.float32))
        return float((r.astype(np.float64) * wr).sum() + (a.astype(np.float64) * wa).sum())

    def grad():
        o.forward(g["positions"])
        return o.backward(wr, wa).astype(np.float64)

    _fd_check(energy, grad, g["positions"].astype(np.float64))


@needs_ref
@pytest.mark.parametrize("kind", ["vacuum", "cubic", "triclinic"])
@pytest.mark.parametrize("torchani", [True, False])
def test_ani_oracle_bitwise_equals_reference(kind, torchani):
    rf, af = workloads.ani2x_functions()
    if kind == "vacuum":
        pos, species = workloads.conformer(50, 0)
        box = None
    elif kind == "cubic":
        pos, species, box = workloads.random_box(200, seed=1)
    else:
        pos, species, box = workloads.triclinic_box(180, seed=2)
    a = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=box is not None, torchani=torchani)
    b = oracle.RefAni(7, 5.1, 3.5, species, rf, af, periodic=box is not None, torchani=torchani)
    ra, aa = a.forward(pos, box)
    rb, ab = b.forward(pos, box)
    assert np.array_equal(ra, rb) and np.array_equal(aa, ab)
    rng = np.random.default_rng(3)
    wr, wa = rng.standard_normal(ra.shape).astype(np.float32), rng.standard_normal(aa.shape).astype(np.float32)
    assert np.array_equal(a.backward(wr, wa), b.backward(wr, wa))


# ---------------------------------------------------------------- CFConv
@pytest.mark.parametrize("tag", ["nonperiodic_ssp", "periodic_ssp", "triclinic_ssp", "nonperiodic_tanh"])
def test_cfconv_oracle_vs_schnetpack_golden(golden_dir, tag):
    """src/schnet/TestCFConv.h:142-247 (values computed with SchNetPack)."""
    g = np.load(f"{golden_dir}/cfconv_water18.npz")
    box = g[f"{tag}_box"] if f"{tag}_box" in g else None
    nb = CFConvNeighborsOracle(18, 2.0, box is not None)
    nb.build(g["positions"], box)
    cf = CFConvOracle(18, 8, 5, 2.0, 0.5, tag.split("_")[1], g["w1"], g["b1"], g["w2"], g["b2"], periodic=box is not None)
    y = cf.forward(nb, g["positions"], g["x"], box)
    np.testing.assert_allclose(y, g[f"{tag}_output"], rtol=2e-6, atol=1e-6 * np.abs(y).max())


@needs_ref
@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("act", ["ssp", "tanh"])
def test_cfconv_oracle_bitwise_equals_reference(periodic, act):
    pos, _, box = workloads.random_box(150, seed=5)
    box = box if periodic else None
    rng = np.random.default_rng(6)
    W, G = 16, 9
    w1, w2 = rng.standard_normal((W, G)).astype(np.float32) * 0.3, rng.standard_normal((W, W)).astype(np.float32) * 0.2
    b1, b2 = rng.standard_normal(W).astype(np.float32), rng.standard_normal(W).astype(np.float32)
    x, gy = rng.standard_normal((150, W)).astype(np.float32), rng.standard_normal((150, W)).astype(np.float32)
    res = []
    for NB, CF in ((CFConvNeighborsOracle, CFConvOracle), (oracle.RefCFConvNeighbors, oracle.RefCFConv)):
        nb = NB(150, 4.0, periodic)
        nb.build(pos, box)
        cf = CF(150, W, G, 4.0, 0.3, act, w1, b1, w2, b2, periodic=periodic)
        res.append((cf.forward(nb, pos, x, box),) + cf.backward(nb, pos, x, gy, box) + nb.export())
    for a, b in zip(*res):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------- getNeighborPairs
def test_neighbor_oracle_vs_reference_cpu_op(golden_dir):
    """96 cases produced by torch.ops.neighbors.getNeighborPairs of the reference on CPU."""
    g = np.load(f"{golden_dir}/neighbors_ref.npz")
    for k in range(int(g["num_cases"])):
        pos, cutoff, mnp, box = g[f"c{k}_positions"], float(g[f"c{k}_cutoff"]), int(g[f"c{k}_max_num_pairs"]), g[f"c{k}_box"]
        nb, dl, ds, n = neighbor_pairs_oracle(pos, cutoff, mnp, box if box.size else None, device_semantics=False)
        assert np.array_equal(nb, g[f"c{k}_neighbors"]), k
        tol = 1e-6 if pos.dtype == np.float32 else 1e-13
        np.testing.assert_allclose(dl, g[f"c{k}_deltas"], rtol=tol, atol=tol, equal_nan=True)
        np.testing.assert_allclose(ds, g[f"c{k}_distances"], rtol=tol, atol=tol, equal_nan=True)
        assert n == int(g[f"c{k}_num_pairs"][0]), k


def test_neighbor_oracle_device_semantics():
    pos = np.zeros((4, 3), np.float32)
    pos[:, 0] = np.arange(4) * 0.1
    nb, dl, ds, n = neighbor_pairs_oracle(pos, 1.0, 4, device_semantics=True)
    assert nb.shape == (2, 4) and n == 6
    nb, dl, ds, n = neighbor_pairs_oracle(pos, 1.0, 4, device_semantics=False)
    assert nb.shape == (2, 6) and n == 6          # CPU reference: not truncated (getNeighborPairsCPU.cpp:86-98)


def test_neighbor_backward_oracle_matches_autograd():
    import torch
    rng = np.random.default_rng(1)
    pos = rng.standard_normal((12, 3))
    nb, dl, ds, _ = neighbor_pairs_oracle(pos, 100.0, -1)
    gd, gs = rng.standard_normal(dl.shape), rng.standard_normal(ds.shape)
    tp = torch.tensor(pos, requires_grad=True)
    d = tp[torch.tensor(nb[0], dtype=torch.long)] - tp[torch.tensor(nb[1], dtype=torch.long)]
    ((d * torch.tensor(gd)).sum() + (d.norm(dim=1) * torch.tensor(gs)).sum()).backward()
    np.testing.assert_allclose(neighbor_pairs_backward_oracle(12, nb, dl, ds, gd, gs), tp.grad.numpy(), rtol=1e-10, atol=1e-12)


# ---------------------------------------------------------------- the reference's own test molecules
MOLECULES = ["1hvj", "1hvk", "2iuz", "3hkw", "3hky", "3lka", "3o99", "water"]


def molecule_weights(shape, k):
    """The upstream gradients of tests/golden/make_golden_molecules.py::weights (a formula, not stored)."""
    i, j = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    return (np.round(np.cos(0.37 * i + 1.3 * j + 0.5 + k) * 64) / 64).astype(np.float32)


@pytest.mark.parametrize("name", MOLECULES)
def test_ani_oracle_on_the_reference_test_molecules(golden_dir, name):
    """src/pytorch/TestSymmetryFunctions.py:37-105: the seven ligands and the 306-atom water box the reference tests
    itself on; expected values = the reference CPU implementation compiled in place (make_golden_molecules.py).
    The restatement is the same arithmetic in the same order: bit for bit."""
    g = np.load(f"{golden_dir}/molecules_ref.npz")
    k = MOLECULES.index(name)
    cell = g[f"{name}_cell"] if f"{name}_cell" in g else None
    rf, af = workloads.ani2x_functions()
    o = AniOracle(7, 5.1, 3.5, g[f"{name}_species"], rf, af, periodic=cell is not None)
    r, a = o.forward(g[f"{name}_positions"], cell)
    assert np.array_equal(r, g[f"{name}_radial"]) and np.array_equal(a, g[f"{name}_angular"])
    grad = o.backward(molecule_weights(r.shape, k), molecule_weights(a.shape, k + 100))
    assert np.array_equal(grad, g[f"{name}_grad"])
