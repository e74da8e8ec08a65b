"""BASELINE.json's full sizes, checked through size-independent properties (the O(N^2) oracle cannot reach them
in test time): config 5 = 100 000 atoms (getNeighborPairs at 5.2 A + ANI-2x AEV), config 3 = 10 000-atom CFConv.

Properties used
  * the cell-grid search and the all-pairs search (the reference's algorithm, run on the GPU) are independent
    code paths: same pair set, same AEV within the parity tolerance;
  * Newton's third law: the position gradient of any function of the AEV / of the convolution output sums to
    zero over the atoms (translation invariance);
  * the pair list is row-grouped, ascending, col < row, |delta| == distance <= cutoff, and every row sampled
    agrees with a brute-force numpy scan of that row (bit-exact index sets).
"""
import numpy as np
import pytest
import torch

from nnpops_amd import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def box100k():
    pos, species, box = workloads.random_box(100000, density=0.1, seed=6)
    return pos, species, box


def test_neighbor_pairs_100k(box100k):
    from nnpops_amd.capi import neighbor_pairs_forward
    pos, _, box = box100k
    dev = torch.device("cuda:0")
    cutoff, max_pairs = 5.2, 3_200_000
    nb, dl, ds, n = neighbor_pairs_forward(torch.tensor(pos, device=dev), cutoff, max_pairs, torch.tensor(box, device=dev))
    nb, dl, ds, n = nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), int(n.item())
    valid = nb[0] >= 0
    assert n == int(valid.sum()) and 2_800_000 < n < max_pairs           # ~2.94e6 expected
    assert np.all(valid[:n]) and not valid[n:].any()                     # compacted, padded with -1
    assert np.all(np.isnan(ds[n:])) and np.all(nb[1][n:] == -1)
    rows, cols = nb[0][:n], nb[1][:n]
    assert np.all(rows > cols) and np.all(np.diff(rows) >= 0)
    np.testing.assert_allclose(np.sqrt((dl[:n].astype(np.float64) ** 2).sum(1)), ds[:n], rtol=1e-6)
    assert ds[:n].max() <= cutoff
    # pair count per atom is symmetric information: every atom's degree from the list == brute force on samples
    L = float(box[0, 0])
    starts = np.searchsorted(rows, np.arange(100001))
    rng = np.random.default_rng(0)
    for row in rng.choice(100000, 300, replace=False):
        d = pos[row].astype(np.float32) - pos[:row]
        d -= np.round(d / np.float32(L)) * np.float32(L)
        r = np.sqrt((d * d).sum(1))
        want = np.nonzero(r <= np.float32(cutoff))[0]
        got = np.sort(cols[starts[row]:starts[row + 1]])
        assert np.array_equal(want, got), row


def test_ani_100k_cells_equal_allpairs_and_forces_sum_to_zero(box100k):
    from nnpops_amd.capi import AniSymmetryFunctions
    pos, species, box = box100k
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    out = {}
    for algorithm in (2, 1):                  # cell grid, then the reference's all-pairs scan on the GPU
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
        sym.set_neighbor_algorithm(algorithm)
        radial, angular = sym.compute(tpos, tbox)
        if not out:
            g_r = torch.randn(radial.shape, device=dev, generator=gen)
            g_a = torch.randn(angular.shape, device=dev, generator=gen)
        grad = sym.backprop(g_r, g_a)
        out[algorithm] = (radial.clone(), angular.clone(), grad.clone())
        del sym
    (r2, a2, f2), (r1, a1, f1) = out[2], out[1]
    assert torch.isfinite(r2).all() and torch.isfinite(a2).all() and torch.isfinite(f2).all()
    torch.testing.assert_close(r2, r1, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(a2, a1, rtol=2e-5, atol=2e-6)
    fmax = float(f1.abs().max())
    assert float((f2 - f1).abs().max()) <= 1e-4 * fmax
    # translation invariance: the net force vanishes (to accumulated rounding of 1e5 fp32 terms)
    net = f2.double().sum(0).abs().max().item()
    assert net <= 1e-3 * fmax, (net, fmax)
    # every atom of a liquid at this density has neighbours: no empty AEV rows
    assert bool((r2.abs().sum(1) > 0).all())


def test_cfconv_10k_forces_sum_to_zero_and_match_vector_kernels(monkeypatch):
    """Config 3 at full size: matrix-core kernels vs the vector kernels (independent code), and Newton's third law."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    n, W, G, cutoff = 10000, 128, 50, 5.0
    pos, _, box = workloads.random_box(n, density=0.1, seed=3)
    rng = np.random.default_rng(4)
    w1 = (0.1 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.1 * rng.standard_normal((W, W))).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    x = torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=dev)
    gy = torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=dev)
    nbrs = CFConvNeighbors(n, cutoff, periodic=True)
    nbrs.build(tpos, tbox, check=True)
    assert 250_000 < nbrs.num_pairs() < 275_000                         # ~2.6e5 half pairs
    res = {}
    for valu in ("0", "1"):
        monkeypatch.setenv("NNPOPS_CFCONV_VALU", valu)
        cf = CFConv(n, W, G, cutoff, 0.1, "ssp", w1, b1, w2, b2, periodic=True)
        y = torch.empty_like(x)
        cf.compute(nbrs, tpos, x, tbox, y)
        gx, gpos = cf.backprop(nbrs, tpos, x, gy, tbox)
        res[valu] = (y.clone(), gx.clone(), gpos.clone())
    (y0, gx0, gp0), (y1, gx1, gp1) = res["0"], res["1"]
    torch.testing.assert_close(y0, y1, rtol=1e-4, atol=1e-4 * float(y1.abs().max()))
    torch.testing.assert_close(gx0, gx1, rtol=1e-4, atol=1e-4 * float(gx1.abs().max()))
    fmax = float(gp1.abs().max())
    assert float((gp0 - gp1).abs().max()) <= 1e-4 * fmax
    assert gp0.double().sum(0).abs().max().item() <= 1e-3 * fmax


def test_cfconv_10k_against_the_oracle_at_full_size():
    """BASELINE config 3 at full size (10 000 atoms periodic, W=128, G=50, N(0, 0.1^2) weights, ssp) against the CPU ORACLE
    (~15 s of single-core work), not against this repository's own vector kernels: the split-fp16 dense layers the handle
    picks by default must hold the same bars as everything else -- outputs and input gradients 2e-5 of the largest entry
    (element-wise), "energy" <gy, y> 1e-5 relative to the sum of its absolute terms, position gradients 1e-4 of the largest
    component."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    from oracle import CFConvNeighborsOracle, CFConvOracle
    n, W, G, cutoff, sigma = 10000, 128, 50, 5.0, 0.1
    pos, _, box = workloads.random_box(n, density=0.1, seed=3)
    rng = np.random.default_rng(4)
    w1 = (0.1 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.1 * rng.standard_normal((W, W))).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    onb = CFConvNeighborsOracle(n, cutoff, True)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True)
    onb.build(pos, box)
    y_ref = ocf.forward(onb, pos, x, box)
    xg_ref, pg_ref = ocf.backward(onb, pos, x, gy, box)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    nb = CFConvNeighbors(n, cutoff, periodic=True)
    nb.build(tpos, tbox, check=True)
    assert nb.num_pairs() == onb.num_pairs()
    cf = CFConv(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True)
    y = torch.empty_like(tx)
    cf.compute(nb, tpos, tx, tbox, y)
    gx, gpos = cf.backprop(nb, tpos, tx, tg, tbox)
    y, gx, gpos = y.cpu().numpy(), gx.cpu().numpy(), gpos.cpu().numpy()
    np.testing.assert_allclose(y, y_ref, rtol=2e-5, atol=2e-5 * np.abs(y_ref).max())
    np.testing.assert_allclose(gx, xg_ref, rtol=2e-5, atol=2e-5 * np.abs(xg_ref).max())
    e_ref, e = float((y_ref.astype(np.float64) * gy).sum()), float((y.astype(np.float64) * gy).sum())
    assert abs(e - e_ref) <= 1e-5 * float(np.abs(y_ref.astype(np.float64) * gy).sum())
    assert np.abs(gpos - pg_ref).max() <= 1e-4 * np.abs(pg_ref).max()


def test_headline_workload_against_the_oracle_at_full_size():
    """The benchmark's own frame (10 000 atoms, periodic, 7 species uniform, seed 100) element by element against the
    CPU oracle (O(N^2), a few seconds): AEV rtol 2e-5 / atol 2e-6, energy 1e-5, forces 1e-4 of the largest component."""
    from nnpops_amd.capi import AniSymmetryFunctions
    from oracle import AniOracle
    pos, species, box = workloads.random_box(10000, density=0.1, seed=100, n_species=7)
    rf, af = workloads.ani2x_functions()
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    r_ref, a_ref = oracle.forward(pos, box)
    rng = np.random.default_rng(0)
    wr = rng.standard_normal(r_ref.shape).astype(np.float32)
    wa = rng.standard_normal(a_ref.shape).astype(np.float32)
    g_ref = oracle.backward(wr, wa)
    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    radial, angular = sym.compute(torch.tensor(pos, device=dev), torch.tensor(box, device=dev))
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
    r, a, g = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()
    np.testing.assert_allclose(r, r_ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a, a_ref, rtol=2e-5, atol=2e-6)
    e_ref = float((r_ref.astype(np.float64) * wr).sum() + (a_ref.astype(np.float64) * wa).sum())
    e = float((r.astype(np.float64) * wr).sum() + (a.astype(np.float64) * wa).sum())
    scale = float(np.abs(r_ref.astype(np.float64) * wr).sum() + np.abs(a_ref.astype(np.float64) * wa).sum())
    assert abs(e - e_ref) <= 1e-5 * scale
    assert np.abs(g - g_ref).max() <= 1e-4 * np.abs(g_ref).max()


def test_ani_one_million_atoms_sampled_against_local_clusters():
    """1 000 000 atoms in one periodic box (10 GB of neighbour data: the layout is N x capacity, nothing is N^2).  Checked
    (i) by cutting the Rcr-neighbourhood of sampled atoms out of the box -- minimum-image shifts applied -- and running
    the oracle on that small vacuum cluster: the central atom's AEV row must equal the row of the big system (an atom's
    AEV depends on nothing else); (ii) net force zero, everything finite."""
    from nnpops_amd.capi import AniSymmetryFunctions
    from oracle import AniOracle
    n = 1_000_000
    pos, species, box = workloads.random_box(n, density=0.1, seed=77)
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    radial, angular = sym.compute(torch.tensor(pos, device=dev), torch.tensor(box, device=dev))
    gen = torch.Generator(device=dev).manual_seed(5)
    g_r = torch.randn(radial.shape, device=dev, generator=gen)
    g_a = torch.randn(angular.shape, device=dev, generator=gen)
    grad = sym.backprop(g_r, g_a)
    assert bool(torch.isfinite(radial).all()) and bool(torch.isfinite(angular).all()) and bool(torch.isfinite(grad).all())
    fmax = float(grad.abs().max())
    assert float(grad.double().sum(0).abs().max()) <= 1e-2 * fmax          # 1e6 fp32 terms
    L = float(box[0, 0])
    rng = np.random.default_rng(3)
    sample = rng.choice(n, 24, replace=False)
    rows_r = radial[torch.tensor(sample, device=dev)].cpu().numpy()
    rows_a = angular[torch.tensor(sample, device=dev)].cpu().numpy()
    for k, i in enumerate(sample):
        d = pos - pos[i]
        d -= np.round(d / np.float32(L)) * np.float32(L)
        near = np.nonzero((d * d).sum(1) < np.float32(5.1 * 5.1 * 1.02))[0]
        near = np.concatenate([[i], near[near != i]])
        cluster = (pos[i] + d[near]).astype(np.float32)                       # the neighbourhood, unwrapped around atom i
        oracle = AniOracle(7, 5.1, 3.5, species[near], rf, af, periodic=False)
        r_ref, a_ref = oracle.forward(cluster, None)
        np.testing.assert_allclose(rows_r[k], r_ref[0], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(rows_a[k], a_ref[0], rtol=2e-5, atol=2e-6)


def test_neighbor_pairs_one_million_atoms():
    """getNeighborPairs at 1 000 000 atoms (29.5 M pairs, 0.7 GB of output): count, ordering, distances and 100 rows
    brute-forced with numpy.  The reference stops at ~65 000 atoms (int32 pair index)."""
    from nnpops_amd.capi import neighbor_pairs_forward
    n, cutoff, max_pairs = 1_000_000, 5.2, 32_000_000
    pos, _, box = workloads.random_box(n, density=0.1, seed=78)
    dev = torch.device("cuda:0")
    nb, dl, ds, num = neighbor_pairs_forward(torch.tensor(pos, device=dev), cutoff, max_pairs, torch.tensor(box, device=dev))
    num = int(num.item())
    assert 28_000_000 < num < max_pairs
    rows, cols = nb[0, :num], nb[1, :num]
    assert bool((rows > cols).all()) and bool((rows[1:] >= rows[:-1]).all()) and bool((nb[0, num:] == -1).all())
    assert bool(torch.isnan(ds[num:]).all()) and float(ds[:num].max()) <= cutoff
    torch.testing.assert_close(dl[:num].double().pow(2).sum(1).sqrt().float(), ds[:num], rtol=1e-6, atol=0)
    starts = torch.searchsorted(rows.contiguous(), torch.arange(n + 1, device=dev, dtype=rows.dtype)).cpu().numpy()
    cols_h = cols.cpu().numpy()
    L = np.float32(box[0, 0])
    rng = np.random.default_rng(4)
    for row in rng.choice(n, 100, replace=False):
        d = pos[row] - pos[:row]
        d -= np.round(d / L) * L
        want = np.nonzero(np.sqrt((d * d).sum(1)) <= np.float32(cutoff))[0]
        assert np.array_equal(want, np.sort(cols_h[starts[row]:starts[row + 1]])), row


def _torchani_reference(model, species, aev_ref):
    """Energy and dE/dAEV of a TorchANI-shaped ensemble in float64 on the host: the per-species Sequential networks
    applied to the given AEV rows, summed over atoms, averaged over the members (reference BatchedNN.py:100-111 is the
    same function in another layout)."""
    import copy
    syms = list(workloads.ANI2X_WIDTHS)
    aev = torch.tensor(aev_ref, dtype=torch.float64, requires_grad=True)
    sp = torch.tensor(species, dtype=torch.long)
    total = torch.zeros((), dtype=torch.float64)
    members = list(model.neural_networks)
    for member in members:
        for s in sorted(set(species.tolist())):
            net = copy.deepcopy(member[syms[s]]).double()
            total = total + net(aev[sp == s]).sum()
    energy = total / len(members)
    energy.backward()
    return float(energy.detach()), aev.grad.numpy().astype(np.float32)


@pytest.mark.parametrize("layout", ["fused", "fused-composition", "gemm", "grouped", "reference"])
def test_config2_optimized_torchani_against_the_oracle_at_full_size(layout):
    """BASELINE config 2 in its own shape -- OptimizedTorchANI (species converter + HIP AEV + BatchedNN + shifter) on the
    2 001-atom periodic water box with an 8-member ANI-2x-shaped ensemble -- against the oracle pipeline: the CPU oracle's
    AEV (O(N^2), 0.4 s) pushed through the same networks in float64 on the host, forces by the oracle's backward of the
    float64 dE/dAEV.  north_star's bars: energy 1e-5 relative, forces 1e-4 of the largest component, for every layout of
    the networks (split-fp16 GEMMs / library GEMMs / the reference's per-atom weights through BatchedLinear)."""
    from NNPOps import OptimizedTorchANI
    from NNPOps.BatchedNN import TorchANIBatchedNN
    from oracle import AniOracle
    from test_torch_surface_gpu import _numbers
    dev = torch.device("cuda:0")
    model = workloads.torchani_like_model(n_models=8, seed=2, self_energies=[-0.5, -38.0, -54.7, -75.2, -398.1, -99.8, -460.1])
    pos, species, box = workloads.water_box(667, seed=1)
    assert len(species) == 2001
    numbers = _numbers(species)
    if layout == "reference":                 # 21.6 GB of per-atom weights: assemble them on the device
        torch.set_default_device(dev)
    try:
        opt = OptimizedTorchANI(model, numbers.cpu(), nn_layout=layout.split("-")[0], fused_step=layout != "fused-composition")
    finally:
        torch.set_default_device("cpu")
    # 'fused': AEV + networks as ONE autograd node (torch.ops.NNPOpsANISymmetryFunctions.energy); the others: the reference's
    # four-module composition with the networks in the named layout
    assert (type(opt).__name__ == "FusedOptimizedTorchANI") == (layout == "fused")
    opt = opt.to(dev)
    tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
    cell, pbc = torch.tensor(box, device=dev), torch.tensor([True, True, True], device=dev)
    energy = opt((numbers, tpos), cell, pbc).energies
    energy.sum().backward()
    forces = tpos.grad[0].cpu().numpy()

    rf, af = workloads.ani2x_functions()
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True)
    r_ref, a_ref = oracle.forward(pos, box)
    e_nn, g_aev = _torchani_reference(model, species, np.concatenate([r_ref, a_ref], axis=1))
    sae = np.array([-0.5, -38.0, -54.7, -75.2, -398.1, -99.8, -460.1])
    e_shift = float(sae[species].sum())
    f_ref = oracle.backward(np.ascontiguousarray(g_aev[:, :112]), np.ascontiguousarray(g_aev[:, 112:]))
    # the shifter adds a constant five orders of magnitude larger than the network part: gate the NETWORK energy
    e_net = float(energy.double().item()) - e_shift
    assert abs(e_net - e_nn) <= 1e-5 * abs(e_nn), (layout, e_net, e_nn)
    assert abs(float(energy.double().item()) - (e_nn + e_shift)) <= 1e-5 * abs(e_nn + e_shift)
    err = np.abs(forces - f_ref).max() / np.abs(f_ref).max()
    assert err <= 1e-4, (layout, err)
