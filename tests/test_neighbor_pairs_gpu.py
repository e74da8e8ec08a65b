"""getNeighborPairs through the C ABI vs the numpy oracle (bit-exact indices, allclose floats).

Mirrors the reference's test matrix (src/pytorch/neighbors/TestNeighbors.py:32-90, 92-140, 143-168,
209-270): sizes 1..1000, cutoffs 1/10/100, fp32/fp64, both output modes, gradients, overflow
semantics, triclinic boxes; plus the large-system cell-grid path the reference cannot run."""
import numpy as np
import pytest
import torch

from oracle import neighbor_pairs_oracle, neighbor_pairs_backward_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(pos, cutoff, max_num_pairs, box, dtype):
    from nnpops_amd.capi import neighbor_pairs_forward
    tp = torch.tensor(pos, dtype=dtype, device=DEV)
    tb = None if box is None else torch.tensor(box, dtype=dtype, device=DEV)
    nb, dl, ds, npairs = neighbor_pairs_forward(tp, cutoff, max_num_pairs, tb)
    torch.cuda.synchronize()
    return nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), int(npairs.item())


def _sorted(nb, dl, ds):
    order = np.lexsort(nb)[::-1]
    return nb[:, order], dl[order], ds[order]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("num_atoms", [1, 2, 3, 4, 5, 10, 100, 1000])
@pytest.mark.parametrize("cutoff", [1, 10, 100])
@pytest.mark.parametrize("all_pairs", [True, False])
def test_values(dtype, num_atoms, cutoff, all_pairs):
    rng = np.random.default_rng(num_atoms * 7 + cutoff)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    pos = (10 * rng.standard_normal((num_atoms, 3))).astype(npdt)
    ref_all = neighbor_pairs_oracle(pos, cutoff, -1)
    found = int(np.count_nonzero(ref_all[0][0] >= 0))
    max_num_pairs = -1 if all_pairs else max(found, 1)
    ref_nb, ref_dl, ref_ds, ref_n = neighbor_pairs_oracle(pos, cutoff, max_num_pairs, device_semantics=True)
    nb, dl, ds, n = _run(pos, cutoff, max_num_pairs, None, dtype)
    assert nb.dtype == np.int32 and dl.dtype == npdt and ds.dtype == npdt
    assert np.array_equal(nb, ref_nb)                       # same slots, same order as the reference CPU op
    np.testing.assert_allclose(dl, ref_dl, rtol=1e-6 if npdt == np.float32 else 1e-12, equal_nan=True)
    np.testing.assert_allclose(ds, ref_ds, rtol=1e-6 if npdt == np.float32 else 1e-12, equal_nan=True)
    assert n == found


def test_docstring_examples():
    """The four worked examples of the reference docstring (getNeighborPairs.py:104-138)."""
    pos = np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]], np.float32)
    nb, dl, ds, n = _run(pos, 3.0, -1, None, torch.float32)
    assert nb.tolist() == [[1, 2, 2], [0, 0, 1]] and ds.tolist() == [1.0, 2.0, 1.0] and n == 3
    nb, dl, ds, n = _run(pos, 1.5, -1, None, torch.float32)
    assert nb.tolist() == [[1, -1, 2], [0, -1, 1]] and np.isnan(ds[1]) and n == 2      # CUDA semantics: true count
    nb, dl, ds, n = _run(pos, 3.0, 6, None, torch.float32)
    assert nb.tolist() == [[1, 2, 2, -1, -1, -1], [0, 0, 1, -1, -1, -1]] and n == 3
    nb, dl, ds, n = _run(pos, 1.5, 6, None, torch.float32)
    assert nb.tolist() == [[1, 2, -1, -1, -1, -1], [0, 1, -1, -1, -1, -1]] and n == 2


def test_too_many_neighbors_are_dropped_not_truncated_silently():
    """Overflow semantics of the device path (TestNeighbors.py:143-168, CUDA.cu:68-78)."""
    pos = np.zeros((4, 3), np.float32)
    pos[:, 0] = np.arange(4) * 0.1
    nb, dl, ds, n = _run(pos, 1.0, 4, None, torch.float32)
    assert n == 6 and np.all(nb >= 0) and nb.shape == (2, 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("num_atoms", [2, 5, 100, 1000])
@pytest.mark.parametrize("grad", ["deltas", "distances", "combined"])
def test_backward(dtype, num_atoms, grad):
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_forward
    rng = np.random.default_rng(num_atoms)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    pos = (10 * rng.standard_normal((num_atoms, 3))).astype(npdt)
    tp = torch.tensor(pos, device=DEV)
    nb, dl, ds, _ = neighbor_pairs_forward(tp, 15.0, -1)
    gd = torch.tensor(rng.standard_normal(tuple(dl.shape)).astype(npdt), device=DEV)
    gs = torch.tensor(rng.standard_normal(tuple(ds.shape)).astype(npdt), device=DEV)
    if grad == "deltas":
        gs.zero_()
    elif grad == "distances":
        gd.zero_()
    gp = neighbor_pairs_backward(num_atoms, nb, dl, ds, gd, gs).cpu().numpy()
    ref = neighbor_pairs_backward_oracle(num_atoms, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(),
                                         gd.cpu().numpy(), gs.cpu().numpy())
    tol = 1e-3 if npdt == np.float32 else 1e-9
    np.testing.assert_allclose(gp, ref, rtol=tol, atol=tol * np.abs(ref).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_backward_bitwise_reproducible(dtype):
    """The backward pass adds the pair forces as fixed-point integers (no float atomics, nnpops_hip.h): the same inputs give the
    same bits every time -- also with the pair list in another order -- where the reference's atomicAdd scatter
    (getNeighborPairsCUDA.cu:96-100) gives a different rounding from run to run."""
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_forward
    rng = np.random.default_rng(5)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    n = 3000
    pos = (12 * rng.random((n, 3))).astype(npdt)
    tp = torch.tensor(pos, device=DEV)
    nb, dl, ds, cnt = neighbor_pairs_forward(tp, 3.0, 400000)
    assert 0 < int(cnt) < 400000
    gd = torch.tensor(rng.standard_normal(tuple(dl.shape)).astype(npdt), device=DEV)
    gs = torch.tensor(rng.standard_normal(tuple(ds.shape)).astype(npdt), device=DEV)
    first = neighbor_pairs_backward(n, nb, dl, ds, gd, gs)
    for _ in range(5):
        assert torch.equal(neighbor_pairs_backward(n, nb, dl, ds, gd, gs), first)
    perm = torch.randperm(nb.shape[1], device=DEV)                       # the same pairs, shuffled
    again = neighbor_pairs_backward(n, nb[:, perm].contiguous(), dl[perm].contiguous(), ds[perm].contiguous(), gd[perm].contiguous(),
                                    gs[perm].contiguous())
    assert torch.equal(again, first)
    ref = neighbor_pairs_backward_oracle(n, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), gd.cpu().numpy(), gs.cpu().numpy())
    tol = 1e-4 if npdt == np.float32 else 1e-10
    np.testing.assert_allclose(first.cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())


def test_backward_keeps_float64_resolution_beside_one_huge_contribution():
    """One pair in near contact (a gradient 1e9 times the others') sets the scale of the fixed-point sums; the second word keeps the
    other atoms' float64 gradients to 1e-13 of their own size (one word, round 4: 1e-12 of the LARGEST contribution = 1e-3 of theirs)."""
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_forward
    rng = np.random.default_rng(15)
    n = 400
    pos = 9 * rng.random((n, 3))
    nb, dl, ds, cnt = neighbor_pairs_forward(torch.tensor(pos, device=DEV), 3.0, 40000)
    gd = rng.standard_normal(tuple(dl.shape))
    gs = rng.standard_normal(tuple(ds.shape))
    k = int(torch.nonzero(nb[0] >= 0)[7])
    gd[k] *= 1.0e9
    got = neighbor_pairs_backward(n, nb, dl, ds, torch.tensor(gd, device=DEV), torch.tensor(gs, device=DEV)).cpu().numpy()
    ref = neighbor_pairs_backward_oracle(n, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), gd, gs)
    a, b = int(nb[0, k]), int(nb[1, k])
    others = np.setdiff1d(np.arange(n), [a, b])
    small = np.abs(ref[others]).max()
    assert small < 1e3 and np.abs(ref[[a, b]]).max() > 1e8
    assert np.abs(got[others] - ref[others]).max() <= 1e-12 * small
    np.testing.assert_allclose(got[[a, b]], ref[[a, b]], rtol=1e-13)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_backward_nan_stays_with_the_two_atoms_of_its_pair(dtype):
    """A pair at distance zero (grad_distance / 0) poisons its two atoms in the reference (getNeighborPairsCUDA.cu:96-100: the
    atomicAdds of a NaN); every other atom's gradient is what it would be without that pair."""
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_forward
    rng = np.random.default_rng(16)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    n = 300
    pos = (8 * rng.random((n, 3))).astype(npdt)
    nb, dl, ds, cnt = neighbor_pairs_forward(torch.tensor(pos, device=DEV), 3.0, 40000)
    gd = torch.tensor(rng.standard_normal(tuple(dl.shape)).astype(npdt), device=DEV)
    gs = torch.tensor(rng.standard_normal(tuple(ds.shape)).astype(npdt), device=DEV)
    k = int(torch.nonzero(nb[0] >= 0)[11])
    a, b = int(nb[0, k]), int(nb[1, k])
    ds_bad, dl_bad = ds.clone(), dl.clone()
    ds_bad[k] = 0
    dl_bad[k] = 0
    got = neighbor_pairs_backward(n, nb, dl_bad, ds_bad, gd, gs).cpu().numpy()
    nb_without = nb.clone()
    nb_without[:, k] = -1
    clean = neighbor_pairs_backward(n, nb_without, dl, ds, gd, gs).cpu().numpy()
    assert np.isnan(got[[a, b]]).all()
    others = np.setdiff1d(np.arange(n), [a, b])
    assert np.isfinite(got[others]).all()
    tol = 1e-5 if npdt == np.float32 else 1e-12
    np.testing.assert_allclose(got[others], clean[others], rtol=tol, atol=tol * np.abs(clean).max())


def _indexed_case(n, cutoff, max_pairs, periodic, npdt, seed):
    from nnpops_amd.capi import neighbor_pairs_forward
    rng = np.random.default_rng(seed)
    edge = (n / 0.1) ** (1.0 / 3.0)                                       # liquid density: ~0.1 atoms per cubic Angstrom
    pos = (edge * rng.random((n, 3))).astype(npdt)
    box = torch.tensor(np.diag([edge] * 3).astype(npdt), device=DEV) if periodic else None
    nb, dl, ds, cnt = neighbor_pairs_forward(torch.tensor(pos, device=DEV), cutoff, max_pairs, box)
    gd = torch.tensor(rng.standard_normal(tuple(dl.shape)).astype(npdt), device=DEV)
    gs = torch.tensor(rng.standard_normal(tuple(ds.shape)).astype(npdt), device=DEV)
    return nb, dl, ds, int(cnt), gd, gs


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("case", ["allpairs_1000", "cells_12000_periodic", "cells_9000_vacuum", "truncated_list", "two_atoms"])
def test_backward_indexed(dtype, case):
    """Round 6: the backward pass of a list the forward op emitted as an owner-computes gather over the list's transposed index
    (no atomics; nnpops_neighbor_pairs_build_index / _backward_indexed) -- the oracle's gradient (getNeighborPairsCUDA.cu:80-101), the
    fixed-point path's gradient to rounding, the same bits on every call; for all three searches of the forward op (all pairs,
    cell grid periodic / vacuum), a list cut short by max_num_pairs, and the smallest system."""
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_backward_indexed, neighbor_pairs_build_index
    npdt = np.float32 if dtype == torch.float32 else np.float64
    n, cutoff, max_pairs, periodic = {"allpairs_1000": (1000, 4.0, 40000, False), "cells_12000_periodic": (12000, 5.0, 400000, True),
                                      "cells_9000_vacuum": (9000, 5.0, 300000, False), "truncated_list": (2000, 4.0, 20000, True),
                                      "two_atoms": (2, 50.0, 4, False)}[case]
    nb, dl, ds, found, gd, gs = _indexed_case(n, cutoff, max_pairs, periodic, npdt, seed=len(case))
    assert (found > max_pairs) == (case == "truncated_list") and found > 0
    index = neighbor_pairs_build_index(n, nb)
    got = neighbor_pairs_backward_indexed(n, nb, dl, ds, gd, gs, index)
    for _ in range(3):
        assert torch.equal(neighbor_pairs_backward_indexed(n, nb, dl, ds, gd, gs, neighbor_pairs_build_index(n, nb)), got)
    ref = neighbor_pairs_backward_oracle(n, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), gd.cpu().numpy(), gs.cpu().numpy())
    tol = 1e-5 if npdt == np.float32 else 1e-12
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())
    fixed = neighbor_pairs_backward(n, nb, dl, ds, gd, gs).cpu().numpy()
    np.testing.assert_allclose(got.cpu().numpy(), fixed, rtol=tol, atol=tol * np.abs(ref).max())
    # the index itself: `order` lists every used slot exactly once, sorted by neighbors[1], ascending slot inside a group
    slots = nb.shape[1]
    order = index[:slots].cpu().numpy()
    cols = nb[1].cpu().numpy()
    used = np.nonzero(cols >= 0)[0]
    assert np.array_equal(order[:len(used)], used[np.argsort(cols[used], kind="stable")])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_backward_indexed_nan_stays_with_the_two_atoms_of_its_pair(dtype):
    """The gather adds every atom's own terms only: a pair at distance zero poisons its two atoms (getNeighborPairsCUDA.cu:96-100)
    and nobody else -- bit for bit nobody else."""
    from nnpops_amd.capi import neighbor_pairs_backward_indexed, neighbor_pairs_build_index
    npdt = np.float32 if dtype == torch.float32 else np.float64
    n = 1500
    nb, dl, ds, found, gd, gs = _indexed_case(n, 4.0, 60000, True, npdt, seed=77)
    index = neighbor_pairs_build_index(n, nb)
    k = int(torch.nonzero(nb[0] >= 0)[123])
    a, b = int(nb[0, k]), int(nb[1, k])
    ds_bad, dl_bad = ds.clone(), dl.clone()
    ds_bad[k] = 0
    dl_bad[k] = 0
    got = neighbor_pairs_backward_indexed(n, nb, dl_bad, ds_bad, gd, gs, index).cpu().numpy()
    clean = neighbor_pairs_backward_indexed(n, nb, dl, ds, gd, gs, index).cpu().numpy()
    assert np.isnan(got[[a, b]]).all()
    others = np.setdiff1d(np.arange(n), [a, b])
    assert np.array_equal(got[others], clean[others])


def test_backward_indexed_keeps_float64_resolution_beside_one_huge_contribution():
    from nnpops_amd.capi import neighbor_pairs_backward_indexed, neighbor_pairs_build_index
    n = 400
    nb, dl, ds, found, gd, gs = _indexed_case(n, 3.0, 40000, False, np.float64, seed=15)
    gdn = gd.cpu().numpy().copy()
    k = int(torch.nonzero(nb[0] >= 0)[7])
    gdn[k] *= 1.0e9
    got = neighbor_pairs_backward_indexed(n, nb, dl, ds, torch.tensor(gdn, device=DEV), gs, neighbor_pairs_build_index(n, nb)).cpu().numpy()
    ref = neighbor_pairs_backward_oracle(n, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), gdn, gs.cpu().numpy())
    a, b = int(nb[0, k]), int(nb[1, k])
    others = np.setdiff1d(np.arange(n), [a, b])
    small = np.abs(ref[others]).max()
    assert np.abs(got[others] - ref[others]).max() <= 1e-12 * small
    np.testing.assert_allclose(got[[a, b]], ref[[a, b]], rtol=1e-13)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("box", [[[10, 0, 0], [0, 10, 0], [0, 0, 10]], [[10, 0, 0], [2, 12, 0], [0, 1, 11]],
                                 [[10, 0, 0], [-2, 12, 0], [0, -1, 11]]])
@pytest.mark.parametrize("all_pairs", [True, False])
def test_periodic(dtype, box, all_pairs):
    """Triclinic minimum image (TestNeighbors.py:209-270)."""
    npdt = np.float32 if dtype == torch.float32 else np.float64
    rng = np.random.default_rng(3)
    pos = (rng.random((100, 3)) * 30 - 15).astype(npdt)
    box = np.array(box, npdt)
    max_num_pairs = -1 if all_pairs else 4000
    ref_nb, ref_dl, ref_ds, ref_n = neighbor_pairs_oracle(pos, 5.0, max_num_pairs, box)
    nb, dl, ds, n = _run(pos, 5.0, max_num_pairs, box, dtype)
    # an exactly half-box component may round either way between numpy (half-even) and the device (half-away)
    assert np.array_equal(nb, ref_nb)
    np.testing.assert_allclose(ds, ref_ds, rtol=2e-5 if npdt == np.float32 else 1e-12, equal_nan=True)
    np.testing.assert_allclose(dl, ref_dl, rtol=2e-5 if npdt == np.float32 else 1e-12, atol=1e-5 if npdt == np.float32 else 1e-12,
                               equal_nan=True)
    assert n == int(np.count_nonzero(ref_nb[0] >= 0)) or not all_pairs


@pytest.mark.parametrize("periodic", [False, True])
def test_large_system_cell_grid(periodic):
    """20 000 atoms: the cell-grid path.  Compared as a SET with the oracle restricted to a slab of rows
    (the full O(N^2) numpy reference would need ~5 GB)."""
    from nnpops_amd import workloads
    pos, _, box = workloads.random_box(20000, seed=21)
    cutoff = 5.2
    nb, dl, ds, n = _run(pos, cutoff, 700000, box if periodic else None, torch.float32)
    valid = nb[0] >= 0
    assert n == int(valid.sum()) and n < 700000
    assert np.all(nb[0][valid] > nb[1][valid])
    # deterministic grouping by row, ascending
    assert np.all(np.diff(nb[0][valid]) >= 0)
    # brute-force check of 400 random rows
    rng = np.random.default_rng(0)
    L = float(box[0, 0])
    for row in rng.choice(20000, 400, replace=False):
        d = pos[row] - pos[:row]
        if periodic:
            d -= np.round(d / L) * L
        r = np.sqrt((d * d).sum(1))
        want = set(np.nonzero(r <= cutoff)[0].tolist())
        got = set(nb[1][valid & (nb[0] == row)].tolist())
        assert want == got, row
    # distances consistent with deltas
    np.testing.assert_allclose(np.sqrt((dl[valid] ** 2).sum(1)), ds[valid], rtol=1e-6)


def test_large_system_cell_grid_float64_matches_float32_pairs():
    """The fp64 instance of the cell-grid path (gathers positions instead of reading the grid's fp32 copy): same pair
    list as the fp32 instance on the same frame, distances to double precision of the double positions."""
    from nnpops_amd import workloads
    pos, _, box = workloads.random_box(12000, seed=23)
    nb32, _, _, n32 = _run(pos, 5.0, 400000, box, torch.float32)
    nb64, dl64, ds64, n64 = _run(pos.astype(np.float64), 5.0, 400000, box.astype(np.float64), torch.float64)
    assert n32 == n64 and np.array_equal(nb32, nb64)              # no pair sits within fp32 rounding of the cutoff here
    L = float(box[0, 0])
    d = pos[nb64[0][:n64]].astype(np.float64) - pos[nb64[1][:n64]].astype(np.float64)
    d -= np.round(d / L) * L
    np.testing.assert_allclose(dl64[:n64], d, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ds64[:n64], np.sqrt((d * d).sum(1)), rtol=1e-14)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_minimum_image_ties_take_the_division(monkeypatch, dtype):
    """The cell-grid path rounds the scaled displacement by reciprocal multiply + round-to-nearest-even and redoes a batch
    with the reference's round(d / L) when a candidate sits at a half-integer (neighbor_pairs.hip).  A lattice whose
    spacing divides half the box, with the cutoff exactly half the box, is made of such candidates -- and keeps them
    (d2 <= cutoff2): the deltas, signs included, must equal those of the all-division build bit for bit, and the oracle's."""
    npdt = np.float32 if dtype == torch.float32 else np.float64
    m, a = 22, 2.0                                             # 22^3 = 10 648 atoms (above the all-pairs threshold), L = 44
    g = np.arange(m, dtype=npdt) * npdt(a)
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(npdt)
    rng = np.random.default_rng(9)
    pos[rng.choice(len(pos), 3000, replace=False)] += rng.normal(0, 0.05, (3000, 3)).astype(npdt)   # and ordinary candidates
    box = np.diag([m * a] * 3).astype(npdt)
    cutoff = 4.0
    fast = _run(pos, cutoff, 2000000, box, dtype)
    monkeypatch.setenv("NNPOPS_PAIRS_DIVIDE", "1")
    exact = _run(pos, cutoff, 2000000, box, dtype)
    assert fast[3] == exact[3] and fast[3] > 100000
    for x, y in zip(fast[:3], exact[:3]):
        assert np.array_equal(x, y, equal_nan=True)
    # ties that are KEPT: the cutoff is exactly half the box along x, and columns of four atoms share their (y, z): the pairs
    # half a box apart have d2 == cutoff2 exactly, and the sign of their delta is the reference's round-half-away
    monkeypatch.delenv("NNPOPS_PAIRS_DIVIDE")
    rc = npdt(5.0)
    yz = rng.uniform(0, 40, (2048, 2)).astype(npdt)
    cols = np.concatenate([np.concatenate([np.full((2048, 1), x, npdt), yz], 1) for x in (0.0, 2.5, 5.0, 7.5)]).astype(npdt)
    cols = cols[rng.permutation(len(cols))]
    cbox = np.diag([10.0, 40.0, 40.0]).astype(npdt)
    f2 = _run(cols, float(rc), 3000000, cbox, dtype)
    monkeypatch.setenv("NNPOPS_PAIRS_DIVIDE", "1")
    e2 = _run(cols, float(rc), 3000000, cbox, dtype)
    assert f2[3] == e2[3]
    for x, y in zip(f2[:3], e2[:3]):
        assert np.array_equal(x, y, equal_nan=True)
    n = f2[3]
    assert int(np.count_nonzero(np.abs(f2[1][:n, 0]) == 5.0)) >= 4096          # the two tie pairs of every column were kept
    ref = neighbor_pairs_oracle(cols, float(rc), -1, cbox)
    got = _sorted(f2[0][:, :n], f2[1][:n], f2[2][:n])
    ok = ref[0][0] >= 0
    want = _sorted(ref[0][:, ok], ref[1][ok], ref[2][ok])
    assert n == int(ok.sum()) and np.array_equal(got[0], want[0])
    # (the numpy oracle rounds half to even like the reference's CPU op, the device code half away from zero like its CUDA
    #  kernel, oracle/neighbors_oracle.py: at the exact ties the two pick opposite images -- same length, opposite sign)
    tie = np.abs(want[1][:, 0]) == 5.0
    assert np.array_equal(got[1][~tie], want[1][~tie]) and np.array_equal(np.abs(got[1][tie]), np.abs(want[1][tie]))
    np.testing.assert_allclose(got[2], want[2], rtol=1e-6 if dtype == torch.float32 else 1e-14)


def test_division_free_cell_split_is_exact_on_grids_of_millions_of_cells(tmp_path):
    """celllist.h::split_cell (the cell index -> (cx, cy, cz) of every stencil walk, no integer division) against c / n over
    whole grids of up to 16.7 M cells -- sizes the per-kernel tests cannot reach (round-2 advisor finding: the uncorrected
    float floor fails above ~2.7 M cells and would silently walk the wrong stencil).  The checker is compiled here with
    hipcc from the same header the library is built from."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "split_cell_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(root, "nnpops_amd", "csrc"),
                           os.path.join(root, "tools", "ubench", "split_cell_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("triclinic", [False, True])
def test_fine_grid_and_no_tie_test_change_nothing(monkeypatch, dtype, triclinic):
    """Round 5: half-width cells walked whole (no per-cell binary search) and the reciprocal minimum image without its tie test
    (cutoff below 0.49 of the shortest edge) must give the list of the round-4 path -- full-width cells, prefixes, tie test --
    pair for pair and bit for bit, in a cubic and in a triclinic box, in both precisions; and both must be the oracle's list."""
    from nnpops_amd import workloads
    n, cutoff = 12000, 5.2                                              # (above the all-pairs threshold: the cell-grid path)
    if triclinic:
        pos, _, box = workloads.triclinic_box(n, seed=11, density=0.1)
    else:
        pos, _, box = workloads.random_box(n, density=0.1, seed=11, n_species=7)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    pos, box = pos.astype(npdt), box.astype(npdt)
    max_pairs = 40 * n
    monkeypatch.setenv("NNPOPS_PAIRS_FINE_GRID", "0")
    monkeypatch.setenv("NNPOPS_PAIRS_TIE_TEST", "1")
    old = _run(pos, cutoff, max_pairs, box, dtype)
    monkeypatch.delenv("NNPOPS_PAIRS_FINE_GRID")
    monkeypatch.delenv("NNPOPS_PAIRS_TIE_TEST")
    new = _run(pos, cutoff, max_pairs, box, dtype)
    assert old[3] == new[3] and 0 < new[3] < max_pairs
    k = new[3]
    a, b = _sorted(old[0][:, :k], old[1][:k], old[2][:k]), _sorted(new[0][:, :k], new[1][:k], new[2][:k])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    rows = new[0][0, :k]
    assert np.all(np.diff(rows) >= 0) and np.all(new[0][1, :k] < rows)   # rows ascending, column below row: the layout of the cell path
    assert np.all(new[0][:, k:] == -1) and np.all(np.isnan(new[2][k:]))
    # the oracle on a sample of rows (brute force over all columns, the reference's arithmetic)
    rng = np.random.default_rng(5)
    starts = np.searchsorted(rows, np.arange(n + 1))
    inv = np.linalg.inv(box.astype(np.float64))
    for row in rng.integers(1, n, size=40):
        d = pos[row].astype(np.float64) - pos[:row].astype(np.float64)
        for axis in (2, 1, 0):                                          # the reference's z, y, x single-round rule
            d -= np.round(d[:, axis] / box[axis, axis])[:, None] * box[axis].astype(np.float64)
        want = np.nonzero((d * d).sum(1) <= cutoff * cutoff)[0]
        got = np.sort(new[0][1, starts[row]:starts[row + 1]])
        near = np.abs(np.sqrt((d * d).sum(1)) - cutoff) < 1e-4            # (pairs within rounding of the cutoff may go either way)
        assert set(want[~near[want]]) <= set(got) <= set(want) | set(np.nonzero(near)[0]), row
