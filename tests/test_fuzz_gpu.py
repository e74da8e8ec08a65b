"""Seeded differential fuzzing: HIP (through the C ABI) against the oracle on configurations nobody hand-picked.

Each case draws its own shape of problem -- species count, radial / angular function sets (any set that factors as
{(eta, Rs)} x {(zeta, theta_s)}, the only kind the torch binding can build), cutoffs, torchani or paper mode,
vacuum / cubic / triclinic box, atom count and density -- and is checked to the same tolerances as the fixed tests.
The point is the corners the fixed tests do not name: one species, a single angular function, nR not a power of two,
padded factor counts (3 x 5, 7 x 3 ...), atoms with zero / one / many neighbours in the same system.
"""
import os

import numpy as np
import pytest
import torch

from nnpops_amd import workloads
from oracle import AniOracle, CFConvNeighborsOracle, CFConvOracle
from oracle.neighbors_oracle import neighbor_pairs_oracle

pytestmark = pytest.mark.gpu

# more seeds for a soak run:  NNPOPS_FUZZ_SCALE=10 python -m pytest tests/test_fuzz_gpu.py -m gpu
SCALE = max(1, int(os.environ.get("NNPOPS_FUZZ_SCALE", "1")))


def _random_geometry(rng, n, kind, density):
    """-> positions, box (or None)"""
    if kind == "vacuum":
        pos, _ = workloads.conformer(n, seed=int(rng.integers(1 << 30)))
        if n > 6:                                            # a few far-away atoms: empty and one-neighbour rows
            pos[-1] += np.float32(40.0)
            pos[-2] = pos[-3] + np.array([1.1, 0, 0], np.float32)
            pos[-3:-1] += np.float32(25.0)
        return pos.astype(np.float32), None
    if kind == "cubic":
        pos, _, box = workloads.random_box(n, density=density, seed=int(rng.integers(1 << 30)))
    else:
        pos, _, box = workloads.triclinic_box(n, seed=int(rng.integers(1 << 30)), density=density)
    shift = rng.uniform(-30, 30, size=3).astype(np.float32)   # atoms need not sit in the primary cell
    return (pos + shift).astype(np.float32), box


@pytest.mark.parametrize("seed", list(range(52)) + list(range(100, 100 + 52 * (SCALE - 1))))        # seeds 40..51 (and 1 in 4 of the soak seeds): dense systems (row / record capacities grow, >32 angular neighbours)
def test_ani_random_configuration(seed, monkeypatch):
    from nnpops_amd.capi import AniSymmetryFunctions
    # systems this small take the fused build + forward kernel by default: every other seed keeps the two-launch path covered
    monkeypatch.setenv("NNPOPS_ANI_FUSE", str(seed % 2))
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(1, 9))
    n_eta_r, n_shf_r = int(rng.integers(1, 3)), int(rng.integers(1, 13))
    n_fr, n_fz = int(rng.integers(1, 9)), int(rng.integers(1, 6))          # angular factor counts, padded to 4/8/16 x 4/8
    rcr = float(rng.uniform(3.5, 5.5))
    rca = float(rng.uniform(2.5, min(3.8, rcr)))
    funcs = workloads.expand_functions(
        EtaR=list(rng.uniform(4.0, 20.0, n_eta_r)), ShfR=list(np.linspace(0.8, rcr - 0.4, n_shf_r)),
        EtaA=[float(rng.uniform(4.0, 14.0))], Zeta=[float(rng.choice([1.0, 2.0, 8.0, 14.1, 32.0]))],
        ShfA=list(np.linspace(0.8, rca - 0.3, n_fr)), ShfZ=list((np.arange(n_fz) + 0.5) * np.pi / n_fz))
    rf, af = funcs
    kind = ["vacuum", "cubic", "triclinic"][seed % 3]
    torchani = bool(rng.integers(0, 2))
    n = int(rng.integers(2, 60)) if kind == "vacuum" else int(rng.integers(150, 500))
    dense = 40 <= seed < 52 or (seed >= 100 and seed % 4 == 0)
    density = float(rng.uniform(0.16, 0.24)) if dense else float(rng.uniform(0.04, 0.11))
    pos, box = _random_geometry(rng, n, kind, density)
    n = len(pos)
    species = rng.integers(0, S, size=n).astype(np.int32)
    periodic = box is not None
    if periodic and min(box[0, 0], box[1, 1], box[2, 2]) < 2.05 * rcr:
        pytest.skip("box below the reference's own 2 x cutoff requirement")
    oracle = AniOracle(S, rcr, rca, species, rf, af, periodic=periodic, torchani=torchani)
    r_ref, a_ref = oracle.forward(pos, box)
    wr = rng.standard_normal(r_ref.shape).astype(np.float32)
    wa = rng.standard_normal(a_ref.shape).astype(np.float32)
    g_ref = oracle.backward(wr, wa)
    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(S, rcr, rca, species, rf, af, periodic=periodic, torchani=torchani)
    roomy = periodic and min(box[0, 0], box[1, 1], box[2, 2]) > 3.3 * rcr      # forcing the cell grid needs 3 cells per axis
    sym.set_neighbor_algorithm(int(rng.integers(0, 3)) if roomy else int(rng.integers(0, 2)))
    radial, angular = sym.compute(torch.tensor(pos, device=dev), torch.tensor(box, device=dev) if periodic else None)
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
    r, a, g = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()
    np.testing.assert_allclose(r, r_ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a, a_ref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(a_ref).max())))
    e_ref, e_gpu = float(r_ref.astype(np.float64).sum() + a_ref.astype(np.float64).sum()), float(r.astype(np.float64).sum() + a.astype(np.float64).sum())
    assert abs(e_gpu - e_ref) <= 1e-5 * abs(e_ref) + 1e-12            # E = sum of the AEV, 1e-5 relative
    fmax = max(float(np.abs(g_ref).max()), 1e-6)
    err = float(np.abs(g - g_ref).max())
    if err > 1e-4 * fmax:
        # North_star's gate is 1e-4 of the largest force.  The only cases ever seen beyond it are dense PAPER-mode systems
        # (torchani=False, reachable through the core API only): without TorchANI's 0.95 damping the angle gradient
        # carries 1 / sin(theta) and is ill conditioned for nearly collinear legs -- for the fp32 reference too.  Such a
        # case is judged against the algorithm evaluated in DOUBLE precision (oracle.AniOracle64): the HIP result must be
        # within 1e-4 of the exact forces, or at least as close to them as the fp32 reference manages to be.
        assert not torchani, (err, fmax)
        _PAPER_MODE_ESCAPES.append((seed, err / fmax))             # counted and reported by the last test of this file
        from oracle import AniOracle64
        o64 = AniOracle64(S, rcr, rca, species, rf, af, periodic=periodic, torchani=torchani)
        o64.forward(pos, box)
        g64 = o64.backward(wr, wa)
        err_ref, err_gpu = float(np.abs(g_ref - g64).max()), float(np.abs(g - g64).max())
        assert err_gpu <= max(1e-4 * fmax, 1.5 * err_ref), (err_gpu, err_ref, fmax)


_PAPER_MODE_ESCAPES = []        # (seed, error / largest force) of every ANI fuzz case that left north_star's 1e-4 bar for the float64 judge


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_cfconv_random_configuration(seed):
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    rng = np.random.default_rng(2000 + seed)
    W = int(rng.choice([1, 3, 8, 16, 24, 32, 48, 64, 80, 96, 100, 112, 128, 160, 256]))
    G = int(rng.integers(2, 65))
    act = ["ssp", "tanh"][seed % 2]
    cutoff = float(rng.uniform(2.5, 6.0))
    sigma = float(rng.uniform(0.08, 0.6))
    kind = ["vacuum", "cubic", "triclinic"][seed % 3]
    n = int(rng.integers(2, 80)) if kind == "vacuum" else int(rng.integers(150, 450))
    pos, box = _random_geometry(rng, n, kind, float(rng.uniform(0.04, 0.1)))
    n = len(pos)
    periodic = box is not None
    if periodic and min(box[0, 0], box[1, 1], box[2, 2]) < 2.05 * cutoff:
        pytest.skip("box below 2 x cutoff")
    w1 = (0.3 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.3 * rng.standard_normal((W, W)) / np.sqrt(W)).astype(np.float32)
    b1 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    onb = CFConvNeighborsOracle(n, cutoff, periodic)
    onb.build(pos, box)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    y_ref = ocf.forward(onb, pos, x, box)
    xg_ref, pg_ref = ocf.backward(onb, pos, x, gy, box)
    dev = torch.device("cuda:0")
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev) if periodic else None
    nb = CFConvNeighbors(n, cutoff, periodic=periodic)
    nb.build(tpos, tbox, check=True)
    cf = CFConv(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    y = torch.empty_like(tx)
    cf.compute(nb, tpos, tx, tbox, y)
    gx, gpos = cf.backprop(nb, tpos, tx, tg, tbox)
    scale_y = max(float(np.abs(y_ref).max()), 1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=5e-5, atol=5e-6 * scale_y)
    np.testing.assert_allclose(gx.cpu().numpy(), xg_ref, rtol=5e-5, atol=5e-6 * max(float(np.abs(xg_ref).max()), 1e-6))
    assert np.abs(gpos.cpu().numpy() - pg_ref).max() <= 1e-4 * max(float(np.abs(pg_ref).max()), 1e-6)


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_cfconv_random_cell_grid_configuration(seed):
    """Systems large enough for the cell-grid neighbour search (>= 1024 atoms), matrix-core widths: the pair slots behind
    the rows, rows that outgrow their capacity at the higher densities, empty space around a non-periodic cloud."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    rng = np.random.default_rng(7000 + seed)
    W = int(rng.choice([16, 32, 48, 64]))
    G = int(rng.integers(4, 21))
    act = ["ssp", "tanh"][seed % 2]
    cutoff = float(rng.uniform(3.5, 5.5))
    sigma = float(rng.uniform(0.2, 0.6))
    kind = ["cubic", "triclinic", "open"][seed % 3]
    n = int(rng.integers(1024, 2200))
    density = float(rng.choice([0.06, 0.1, 0.2]))
    pos, box = _random_geometry(rng, n, "triclinic" if kind == "triclinic" else "cubic", density)
    if kind == "open":
        box = None
    periodic = box is not None
    if periodic and min(box[0, 0], box[1, 1], box[2, 2]) < 3.05 * cutoff:
        pytest.skip("box below 3 cells")
    w1 = (0.3 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.3 * rng.standard_normal((W, W)) / np.sqrt(W)).astype(np.float32)
    b1 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    onb = CFConvNeighborsOracle(n, cutoff, periodic)
    onb.build(pos, box)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    y_ref = ocf.forward(onb, pos, x, box)
    xg_ref, pg_ref = ocf.backward(onb, pos, x, gy, box)
    dev = torch.device("cuda:0")
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev) if periodic else None
    nb = CFConvNeighbors(n, cutoff, periodic=periodic)
    nb.build(tpos, tbox, check=True)
    cf = CFConv(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    y = torch.empty_like(tx)
    cf.compute(nb, tpos, tx, tbox, y)
    gx, gpos = cf.backprop(nb, tpos, tx, tg, tbox)
    atoms, _ = nb.export()
    start, other, _ = onb.export()
    assert atoms.shape[1] == len(other)
    scale_y = max(float(np.abs(y_ref).max()), 1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=5e-5, atol=5e-6 * scale_y)
    np.testing.assert_allclose(gx.cpu().numpy(), xg_ref, rtol=5e-5, atol=5e-6 * max(float(np.abs(xg_ref).max()), 1e-6))
    assert np.abs(gpos.cpu().numpy() - pg_ref).max() <= 1e-4 * max(float(np.abs(pg_ref).max()), 1e-6)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_neighbor_pairs_random_configuration(seed):
    from nnpops_amd.capi import neighbor_pairs_forward
    rng = np.random.default_rng(3000 + seed)
    dtype = [torch.float32, torch.float64][seed % 2]
    npdt = np.float32 if dtype == torch.float32 else np.float64
    n = int(rng.integers(1, 400))
    cutoff = float(rng.uniform(1.0, 6.0))
    kind = ["vacuum", "cubic", "triclinic"][seed % 3]
    L = float(rng.uniform(2.2 * cutoff, 4.0 * cutoff))
    pos = (rng.random((n, 3)) * L - L / 3).astype(npdt)
    if kind == "vacuum":
        box = None
    elif kind == "cubic":
        box = np.eye(3, dtype=npdt) * npdt(L)
    else:
        box = np.array([[L, 0, 0], [0.2 * L, 1.1 * L, 0], [-0.1 * L, 0.15 * L, 1.2 * L]], dtype=npdt)
    total = n * (n - 1) // 2
    mode = seed % 4
    max_pairs = -1 if mode == 0 else (max(1, total) if mode == 1 else max(1, int(rng.integers(1, max(2, total)))))
    ref_nb, ref_dl, ref_ds, ref_n = neighbor_pairs_oracle(pos, cutoff, max_pairs, box)
    dev = torch.device("cuda:0")
    nb, dl, ds, num = neighbor_pairs_forward(torch.tensor(pos, device=dev), cutoff, max_pairs,
                                             torch.tensor(box, device=dev) if box is not None else None)
    nb, dl, ds, num = nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), int(num.item())
    found = int(np.count_nonzero(ref_nb[0] >= 0)) if max_pairs == -1 else ref_n
    assert nb.shape == ref_nb.shape and dl.shape == ref_dl.shape
    if max_pairs == -1:
        assert np.array_equal(nb, ref_nb)
        np.testing.assert_allclose(ds, ref_ds, rtol=2e-5 if npdt == np.float32 else 1e-12, equal_nan=True)
    else:
        # compacted: same SET of pairs when they all fit, same count always (surplus pairs are dropped, never invented)
        true_pairs = set(map(tuple, np.stack(neighbor_pairs_oracle(pos, cutoff, max(1, total), box)[0], 1)[
            neighbor_pairs_oracle(pos, cutoff, max(1, total), box)[0][0] >= 0].tolist())) if total else set()
        assert num == len(true_pairs)
        got = [tuple(p) for p in np.stack(nb, 1).tolist() if p[0] >= 0]
        assert len(got) == min(num, nb.shape[1]) and len(set(got)) == len(got) and set(got) <= true_pairs
        if num <= nb.shape[1]:
            assert set(got) == true_pairs
        # (round 6) the backward pass as a gather over the list's transposed index: the oracle's gradient, whatever the size, the
        # padding, the truncation and the search that made the list
        from nnpops_amd.capi import neighbor_pairs_backward_indexed, neighbor_pairs_build_index
        from oracle import neighbor_pairs_backward_oracle
        t_nb, t_dl, t_ds, _ = neighbor_pairs_forward(torch.tensor(pos, device=dev), cutoff, max_pairs,
                                                     torch.tensor(box, device=dev) if box is not None else None)
        gd = rng.standard_normal(dl.shape).astype(npdt)
        gs = rng.standard_normal(ds.shape).astype(npdt)
        index = neighbor_pairs_build_index(n, t_nb)
        gp = neighbor_pairs_backward_indexed(n, t_nb, t_dl, t_ds, torch.tensor(gd, device=dev), torch.tensor(gs, device=dev), index).cpu().numpy()
        gp_ref = neighbor_pairs_backward_oracle(n, nb, dl, ds, gd, gs)
        tol = 1e-5 if npdt == np.float32 else 1e-12
        np.testing.assert_allclose(gp, gp_ref, rtol=tol, atol=tol * max(float(np.abs(gp_ref).max()), 1e-30))


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_ani_batched_molecules_random(seed):
    """nnpops_ani_set_molecules: a random batch of independent molecules in one handle equals the oracle molecule by
    molecule (atoms of different molecules must never see each other, whatever their coordinates)."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rng = np.random.default_rng(4000 + seed)
    rf, af = workloads.ani2x_functions()
    sizes = [int(rng.integers(1, 70)) for _ in range(int(rng.integers(2, 12)))]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pos = np.concatenate([workloads.conformer(k, seed=int(rng.integers(1 << 30)))[0] for k in sizes]).astype(np.float32)
    species = rng.integers(0, 7, size=len(pos)).astype(np.int32)       # all molecules sit on top of each other in space
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=False)
    sym.set_molecules(offsets)
    dev = torch.device("cuda:0")
    radial, angular = sym.compute(torch.tensor(pos, device=dev), None)
    wr = rng.standard_normal(tuple(radial.shape)).astype(np.float32)
    wa = rng.standard_normal(tuple(angular.shape)).astype(np.float32)
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
    r, a, g = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()
    for m in range(len(sizes)):
        lo, hi = offsets[m], offsets[m + 1]
        oracle = AniOracle(7, 5.1, 3.5, species[lo:hi], rf, af, periodic=False)
        r_ref, a_ref = oracle.forward(pos[lo:hi], None)
        g_ref = oracle.backward(wr[lo:hi], wa[lo:hi])
        np.testing.assert_allclose(r[lo:hi], r_ref, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(a[lo:hi], a_ref, rtol=2e-5, atol=2e-6)
        assert np.abs(g[lo:hi] - g_ref).max() <= 1e-4 * max(float(np.abs(g_ref).max()), 1e-6)


def test_zz_report_paper_mode_escapes():
    """the float64 escape of test_ani_random_configuration is an EXCEPTION to north_star's 1e-4 force bar (dense
    paper-mode systems only, unreachable from the torch surface) -- it must be visible, not silent.  Runs last in this file: says how
    many seeds took the escape (a warning, shown in the pytest summary and so in the GPUTEST tail) and fails if it is more than a
    handful (3 % of the ANI seeds), which would mean the HIP arithmetic has drifted, not that a few frames are ill conditioned."""
    import warnings
    n_seeds = 52 * SCALE
    if _PAPER_MODE_ESCAPES:
        worst = max(e for _, e in _PAPER_MODE_ESCAPES)
        warnings.warn(f"ANI fuzz: {len(_PAPER_MODE_ESCAPES)} of {n_seeds} seeds (paper mode, dense) exceeded 1e-4 of the largest force against the "
                      f"fp32 oracle and were judged against float64 instead: seeds {[s for s, _ in _PAPER_MODE_ESCAPES]}, worst {worst:.2e}")
    assert len(_PAPER_MODE_ESCAPES) <= max(2, int(0.03 * n_seeds)), _PAPER_MODE_ESCAPES
