"""Parity of the HIP ANI symmetry functions (through the C ABI) against the oracle.

Tolerances (BASELINE.json north_star): 1e-5 relative on energies, 1e-4 on forces.  "Energy" here
is a fixed random linear functional of the AEV (E = <w, aev>), so that its position gradient
exercises backprop with a dense upstream gradient.  Element-wise AEV agreement is checked too.
"""
import numpy as np
import pytest
import torch

from nnpops_amd import workloads
from oracle import AniOracle

pytestmark = pytest.mark.gpu

ENERGY_RTOL = 1e-5
FORCE_RTOL = 1e-4      # relative to the largest force component of the system
AEV_ATOL, AEV_RTOL = 2e-6, 2e-5


def _run_case(n_species, rcr, rca, species, rf, af, pos, box, torchani=True, seed=0, algorithm=0):
    from nnpops_amd.capi import AniSymmetryFunctions
    periodic = box is not None
    oracle = AniOracle(n_species, rcr, rca, species, rf, af, periodic=periodic, torchani=torchani)
    r_ref, a_ref = oracle.forward(pos, box)
    rng = np.random.default_rng(seed)
    wr = rng.standard_normal(r_ref.shape).astype(np.float32)
    wa = rng.standard_normal(a_ref.shape).astype(np.float32)
    g_ref = oracle.backward(wr, wa)

    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(n_species, rcr, rca, species, rf, af, periodic=periodic, torchani=torchani)
    sym.set_neighbor_algorithm(algorithm)
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev) if periodic else None
    radial, angular = sym.compute(tpos, tbox)
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
    torch.cuda.synchronize()
    r, a, g = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()

    assert np.all(np.isfinite(r)) and np.all(np.isfinite(a)) and np.all(np.isfinite(g))
    np.testing.assert_allclose(r, r_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    np.testing.assert_allclose(a, a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    e_ref = float((r_ref.astype(np.float64) * wr).sum() + (a_ref.astype(np.float64) * wa).sum())
    e = float((r.astype(np.float64) * wr).sum() + (a.astype(np.float64) * wa).sum())
    scale = float(np.abs(r_ref.astype(np.float64) * wr).sum() + np.abs(a_ref.astype(np.float64) * wa).sum())
    assert abs(e - e_ref) <= ENERGY_RTOL * scale, (e, e_ref, scale)
    # north_star's energy gate in its plainest form: E = sum of the AEV (all terms >= 0, nothing cancels), 1e-5 relative
    e_sum_ref = float(r_ref.astype(np.float64).sum() + a_ref.astype(np.float64).sum())
    e_sum = float(r.astype(np.float64).sum() + a.astype(np.float64).sum())
    assert abs(e_sum - e_sum_ref) <= ENERGY_RTOL * abs(e_sum_ref), (e_sum, e_sum_ref)
    fmax = np.abs(g_ref).max()
    assert np.abs(g - g_ref).max() <= FORCE_RTOL * fmax, (np.abs(g - g_ref).max(), fmax)
    return r, a, g


@pytest.mark.parametrize("tag", ["nonperiodic", "periodic", "triclinic"])
@pytest.mark.parametrize("torchani", [True, False])
def test_water18_golden(golden_dir, tag, torchani):
    """The reference's own fixture (src/ani/TestANISymmetryFunctions.h:63-252)."""
    g = np.load(f"{golden_dir}/ani_water18.npz")
    box = g[f"{tag}_box"] if tag != "nonperiodic" else None
    r, a, _ = _run_case(2, 4.5, 3.5, g["species"], g["radial_functions"], g["angular_functions"], g["positions"], box,
                        torchani=torchani)
    if torchani:   # TorchANI-generated expected values, same tolerance form as the reference test but strict
        np.testing.assert_allclose(r, g[f"{tag}_radial"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(a, g[f"{tag}_angular"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("n_atoms,seed", [(50, 0), (21, 1), (116, 2)])
def test_ani2x_conformer(n_atoms, seed):
    """BASELINE config 1 (50-atom molecule in vacuum) and ligand-sized neighbours."""
    pos, species = workloads.conformer(n_atoms, seed)
    rf, af = workloads.ani2x_functions()
    _run_case(7, 5.1, 3.5, species, rf, af, pos, None)


def test_ani2x_periodic_box_small():
    pos, species, box = workloads.random_box(600, seed=3)
    rf, af = workloads.ani2x_functions()
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


def test_ani2x_water_box():
    """BASELINE config 2 geometry (periodic water), sized so the oracle finishes in seconds."""
    pos, species, box = workloads.water_box(300, seed=1)
    rf, af = workloads.ani2x_functions()
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


def test_ani2x_triclinic_box():
    pos, species, box = workloads.triclinic_box(500, seed=4)
    rf, af = workloads.ani2x_functions()
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


@pytest.mark.parametrize("algorithm", [1, 2])
@pytest.mark.parametrize("kind", ["cubic", "triclinic", "vacuum"])
def test_neighbor_algorithms_agree_with_oracle(algorithm, kind):
    """All-pairs scan (the reference's search) and the cell grid must give the same physics."""
    rf, af = workloads.ani2x_functions()
    if kind == "cubic":
        pos, species, box = workloads.random_box(700, seed=8)
        pos = pos + np.float32(40.0)          # atoms far outside the primary cell: wrapping must cope
    elif kind == "triclinic":
        pos, species, box = workloads.triclinic_box(650, seed=9)
    else:
        pos, species = workloads.conformer(300, seed=10)
        box = None
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box, algorithm=algorithm)


def test_cell_grid_falls_back_when_box_too_small(monkeypatch):
    """A system large enough for the cell grid (the threshold is lowered to this one's size), but a 14.6 A box has < 3 cells
    per axis at Rcr 5.1: the handle must notice on the device, switch to the all-pairs search and still be right (also grows
    the rows)."""
    monkeypatch.setenv("NNPOPS_ANI_CELL_ATOMS", "1024")
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1100, density=0.35, seed=12, min_dist=0.5)
    assert box[0, 0] < 3 * 5.1
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


def test_cell_bins_grow_on_overflow(monkeypatch):
    """The two-kernel grid build drops atom ids into fixed-capacity per-cell bins; a cell with more atoms than
    a bin must be noticed on the device and the bins grown by check() (here forced: 4 ids per bin, ~13 per cell)."""
    monkeypatch.setenv("NNPOPS_CELL_BIN_CAP", "4")
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1500, seed=21)
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box, algorithm=2)


def test_repeated_builds_reuse_clean_histogram():
    """Ten evaluations on moving atoms through one handle: the cell histogram is cleared by the consumer kernel,
    never by a memset, so stale counts would show up as wrong neighbours from the second call on."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1400, seed=22)
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    sym.set_neighbor_algorithm(2)
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    tbox = torch.tensor(box, device=dev)
    for it in range(10):
        pos = (pos + rng.normal(0, 0.05, pos.shape)).astype(np.float32)
        radial, angular = sym.compute(torch.tensor(pos, device=dev), tbox)
        if it in (0, 1, 9):
            r_ref, a_ref = oracle.forward(pos, box)
            np.testing.assert_allclose(radial.cpu().numpy(), r_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
            np.testing.assert_allclose(angular.cpu().numpy(), a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)


def test_forces_bitwise_reproducible():
    """The backward pass has no atomics (legs are gathered by the owner atom), so two evaluations of the same
    frame must agree to the last bit -- something the reference's CUDA path (atomicAdd scatter) cannot promise."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1300, seed=23)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    gen = torch.Generator(device=dev).manual_seed(3)
    runs = []
    for _ in range(3):
        radial, angular = sym.compute(tpos, tbox)
        if not runs:
            g_r = torch.randn(radial.shape, device=dev, generator=gen)
            g_a = torch.randn(angular.shape, device=dev, generator=gen)
        grad = sym.backprop(g_r, g_a)
        runs.append((radial.cpu().numpy().copy(), angular.cpu().numpy().copy(), grad.cpu().numpy().copy()))
    for r, a, g in runs[1:]:
        assert np.array_equal(r, runs[0][0]) and np.array_equal(a, runs[0][1]) and np.array_equal(g, runs[0][2])


def test_denser_frame_after_the_last_check_is_still_exact():
    """check() sizes the angular backward's LDS pair matrix to the busiest atom it has seen (compact layout).  A later
    frame evaluated WITHOUT a check (graph replay, check=False) may hold a busier atom: the kernel must notice per
    atom and fall back to tile pairs inside the space it has, not corrupt anything."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1500, seed=24)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    dev = torch.device("cuda:0")
    sym.compute(torch.tensor(pos, device=dev), torch.tensor(box, device=dev))          # with check: sizes the matrix
    _, max_a = sym.neighbor_stats()
    shrink = np.float32(0.93)                                                            # ~24 % denser: busier atoms
    pos2, box2 = pos * shrink, box * shrink
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    r_ref, a_ref = oracle.forward(pos2, box2)
    busiest = int((np.linalg.norm(a_ref.reshape(len(pos), -1), axis=1) > 0).sum())      # (sanity: everyone has triples)
    assert busiest == len(pos)
    rng = np.random.default_rng(2)
    wr = rng.standard_normal(r_ref.shape).astype(np.float32)
    wa = rng.standard_normal(a_ref.shape).astype(np.float32)
    g_ref = oracle.backward(wr, wa)
    radial, angular = sym.compute(torch.tensor(pos2, device=dev), torch.tensor(box2, device=dev), check=False)
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
    _, max_b = sym.neighbor_stats()
    assert max_b > max_a and max_b <= 32, (max_a, max_b)       # the scenario really happened, rows did not overflow
    np.testing.assert_allclose(angular.cpu().numpy(), a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    assert np.abs(grad.cpu().numpy() - g_ref).max() <= FORCE_RTOL * np.abs(g_ref).max()


def test_capacity_check_in_two_halves():
    """nnpops_ani_check_begin / _end (include/nnpops_hip.h): not deferrable before the capacities have been fitted; then _begin
    queues the copy, consumers may be launched, _end says OK; a frame dense enough to overflow the fitted rows makes _end grow
    them and return ERR_CAPACITY, after which compute() gives the oracle's numbers."""
    from nnpops_amd.capi import AniSymmetryFunctions, OK, ERR_CAPACITY
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1200, seed=31)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    sym.compute(tpos, tbox, check=False)
    assert sym.check_begin() is False                       # first frame: the full check has capacities to fit
    sym.compute(tpos, tbox, check=True)
    ref = [t.clone() for t in sym.compute(tpos, tbox, check=True)]
    r, a = sym.compute(tpos, tbox, check=False)
    assert sym.check_begin() is True
    grad = sym.backprop(torch.ones_like(r), torch.ones_like(a))     # a consumer launched between the halves
    assert sym.check_end() == OK
    assert torch.equal(r, ref[0]) and torch.equal(a, ref[1]) and bool(torch.isfinite(grad).all())
    shrink = np.float32(0.72)                               # 2.7 x the density: rows outgrow max_row * 1.25 + 8
    pos2, box2 = pos * shrink, box * shrink
    tpos2, tbox2 = torch.tensor(pos2, device=dev), torch.tensor(box2, device=dev)
    sym.compute(tpos2, tbox2, check=False)
    assert sym.check_begin() is True
    sym.backprop(torch.ones_like(r), torch.ones_like(a))            # harmless on the clamped rows
    assert sym.check_end() == ERR_CAPACITY
    r2, a2 = sym.compute(tpos2, tbox2, check=True)
    r_ref, a_ref = AniOracle(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=True).forward(pos2, box2)
    np.testing.assert_allclose(r2.cpu().numpy(), r_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    np.testing.assert_allclose(a2.cpu().numpy(), a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)


@pytest.mark.parametrize("kernel", ["0", "1", "2"])
@pytest.mark.parametrize("kind", ["water", "seven_species", "dense"])
def test_both_angular_forward_kernels(monkeypatch, kernel, kind):
    """The handle picks one of two angular forward kernels from the species composition (chunked view of the triple
    list for few well-filled species pairs, run merging for many equally likely species); $NNPOPS_ANI_FORWARD forces
    either, and both must pass on every kind of system."""
    monkeypatch.setenv("NNPOPS_ANI_FORWARD", kernel)
    monkeypatch.setenv("NNPOPS_ANI_FUSE", "0")                 # (the stand-alone kernels; test_fused_build_and_forward covers the fused one)
    rf, af = workloads.ani2x_functions()
    if kind == "water":
        pos, species, box = workloads.water_box(350, seed=31)
    elif kind == "seven_species":
        pos, species, box = workloads.random_box(1100, seed=32)
    else:
        pos, species, box = workloads.random_box(900, density=0.2, seed=33)       # > 32 angular neighbours: records grow
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


@pytest.mark.parametrize("kind", ["water", "seven_species", "dense", "vacuum", "triclinic", "tiny_box"])
def test_fused_build_and_forward(monkeypatch, kind):
    """$NNPOPS_ANI_FUSE=1: neighbour build, radial and angular AEV of an atom in one workgroup (ani_build_forward.h; the
    default for systems of up to 4096 atoms, where a launch less is worth 6-13 % of a step).  Same answers on every kind
    of system, including the all-pairs search (vacuum) and a box too small for the cell stencil (the handle falls back
    and recomputes)."""
    monkeypatch.setenv("NNPOPS_ANI_FUSE", "1")
    monkeypatch.setenv("NNPOPS_ANI_CELL_ATOMS", "1024")        # (the periodic cases below are meant to take the cell grid)
    rf, af = workloads.ani2x_functions()
    box = None
    if kind == "water":
        pos, species, box = workloads.water_box(350, seed=31)
    elif kind == "seven_species":
        pos, species, box = workloads.random_box(1100, seed=32)
    elif kind == "dense":
        pos, species, box = workloads.random_box(900, density=0.2, seed=33)       # > 32 angular neighbours: records grow
    elif kind == "vacuum":
        pos, species = workloads.conformer(120, seed=34)
    elif kind == "triclinic":
        pos, species, box = workloads.triclinic_box(1200, seed=35)
    else:
        pos, species, box = workloads.random_box(1100, density=0.35, seed=12, min_dist=0.5)
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


@pytest.mark.parametrize("fuse", ["0", "1"])
@pytest.mark.parametrize("dyn", ["0", "1"])
@pytest.mark.parametrize("kind", ["water", "seven_species", "dense", "organic", "one_species"])
def test_forward_quads_dealt_out_per_atom(monkeypatch, kind, dyn, fuse):
    """$NNPOPS_ANI_FWD_DYN: the matrix-core forward with its quads dealt out per atom from the atom's bucket sizes (a bucket larger
    than the piece length shared by consecutive quads of one wave, pairs without triples no quad at all) against the fixed quad
    set per species pair -- the handle picks by composition (on for H/C/N/O molecules, off for water and for seven equally
    likely species); both must give the oracle's AEV on every kind of system, in the stand-alone and in the fused kernel."""
    monkeypatch.setenv("NNPOPS_ANI_FWD_DYN", dyn)
    monkeypatch.setenv("NNPOPS_ANI_FUSE", fuse)
    rf, af = workloads.ani2x_functions()
    box = None
    if kind == "water":
        pos, species, box = workloads.water_box(350, seed=31)
    elif kind == "seven_species":
        pos, species, box = workloads.random_box(1100, seed=32)
    elif kind == "dense":
        pos, species, box = workloads.random_box(900, density=0.2, seed=33)       # > 32 angular neighbours: several chunks, records grow
    elif kind == "organic":
        pos, species = workloads.conformer(140, seed=36)                          # H / C / N / O: ten buckets, very uneven
    else:
        pos, species, box = workloads.random_box(1000, seed=37, n_species=1)      # one bucket shared by all 32 quads
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box)


@pytest.mark.parametrize("fine", ["0", "1"])
def test_large_cluster_in_vacuum_uses_the_cell_grid(monkeypatch, fine):
    """A non-periodic system of more than 1024 atoms takes the five-kernel grid build over its bounding box (no wrap, cells
    at the faces have fewer stencil ranges), with half-cutoff or full-width cells ($NNPOPS_ANI_FINE_GRID)."""
    monkeypatch.setenv("NNPOPS_ANI_FINE_GRID", fine)
    rf, af = workloads.ani2x_functions()
    pos, species, _ = workloads.random_box(2600, seed=57)      # the same lattice gas, without its box
    pos = pos - np.float32(13.0)                               # (negative coordinates: the grid origin is the bounding box's corner)
    _run_case(7, 5.1, 3.5, species, rf, af, pos, None, algorithm=2)


def test_kernel_timing_brackets_and_stride(monkeypatch):
    """nnpops_ani_enable_timing / _set_timing_stride / _get_timing: HIP events around the kernels a benchmark selects, on
    every k-th launch; counters reset by get_timing; results unchanged by the brackets.  (Two launches for build and
    angular forward: a system this small would otherwise take the fused kernel, which is timed as the build.)"""
    from nnpops_amd.capi import AniSymmetryFunctions
    monkeypatch.setenv("NNPOPS_ANI_FUSE", "0")
    monkeypatch.setenv("NNPOPS_ANI_CELL_ATOMS", "1024")        # (the cell-grid kernels are among the bracketed ones)
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1500, seed=71)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    radial, angular = sym.compute(tpos, tbox)
    ref = sym.backprop(torch.ones_like(radial), torch.ones_like(angular)).clone()
    sym.enable_timing(True)
    for _ in range(6):
        sym.compute(tpos, tbox, radial, angular, check=False)
        grad = sym.backprop(torch.ones_like(radial), torch.ones_like(angular))
    t = sym.get_timing()
    assert t["radial_forward"][1] == 0                         # the radial AEV is written by the neighbour kernel
    for k in ("neighbors", "angular_forward", "radial_backward", "angular_backward", "cell_grid"):
        ms, launches = t[k]
        assert launches == 6 and 0.0 < ms / launches < 5.0, (k, t[k])
    assert torch.equal(grad, ref)
    assert all(c == 0 for _, c in sym.get_timing().values())  # reset by the previous call
    sym.enable_timing(True, only=["angular_forward"], every=4)
    for _ in range(8):
        sym.compute(tpos, tbox, radial, angular, check=False)
    t = sym.get_timing()
    assert t["angular_forward"][1] == 2 and t["neighbors"][1] == 0, t
    assert sym.timing_overhead() >= 0.0
    sym.enable_timing(False)


def test_single_atom_and_isolated_atoms():
    rf, af = workloads.ani2x_functions()
    pos = np.array([[0, 0, 0], [30, 0, 0], [0, 30, 0]], dtype=np.float32)
    r, a, g = _run_case(7, 5.1, 3.5, np.array([0, 1, 3], np.int32), rf, af, pos, None)
    assert not r.any() and not a.any() and not g.any()
    _run_case(7, 5.1, 3.5, np.array([2], np.int32), rf, af, pos[:1], None)


def test_backprop_uses_last_compute():
    """Stateful contract (reference ANISymmetryFunctions.h:83-84)."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    pos1, species = workloads.conformer(40, 5)
    pos2 = pos1 + np.random.default_rng(6).normal(scale=0.05, size=pos1.shape).astype(np.float32)
    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af)
    oracle = AniOracle(7, 5.1, 3.5, species, rf, af)
    t1, t2 = torch.tensor(pos1, device=dev), torch.tensor(pos2, device=dev)
    sym.compute(t1)
    r, a = sym.compute(t2)
    wr, wa = torch.ones_like(r), torch.ones_like(a)
    g = sym.backprop(wr, wa).cpu().numpy()
    oracle.forward(pos2)
    g_ref = oracle.backward(wr.cpu().numpy(), wa.cpu().numpy())
    assert np.abs(g - g_ref).max() <= FORCE_RTOL * np.abs(g_ref).max()


def test_batched_molecules_match_per_molecule_oracle():
    """BASELINE config 4 in miniature: a batch of independent conformers evaluated by ONE handle
    (nnpops_ani_set_molecules) must equal the oracle run molecule by molecule, even when the molecules
    overlap in space (they are independent systems, not a cluster)."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    rng = np.random.default_rng(11)
    sizes = rng.integers(20, 70, size=24)
    mols = [workloads.conformer(int(n), seed=100 + k) for k, n in enumerate(sizes)]
    pos = np.concatenate([m[0] for m in mols]).astype(np.float32)
    species = np.concatenate([m[1] for m in mols]).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dev = torch.device("cuda:0")
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af)
    sym.set_molecules(offsets)
    radial, angular = sym.compute(torch.tensor(pos, device=dev))
    wr = rng.standard_normal(tuple(radial.shape)).astype(np.float32)
    wa = rng.standard_normal(tuple(angular.shape)).astype(np.float32)
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev)).cpu().numpy()
    r, a = radial.cpu().numpy(), angular.cpu().numpy()
    for k in range(len(sizes)):
        lo, hi = offsets[k], offsets[k + 1]
        o = AniOracle(7, 5.1, 3.5, species[lo:hi], rf, af)
        r_ref, a_ref = o.forward(pos[lo:hi])
        np.testing.assert_allclose(r[lo:hi], r_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
        np.testing.assert_allclose(a[lo:hi], a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
        g_ref = o.backward(np.ascontiguousarray(wr[lo:hi]), np.ascontiguousarray(wa[lo:hi]))
        assert np.abs(grad[lo:hi] - g_ref).max() <= FORCE_RTOL * np.abs(g_ref).max()


def test_handles_release_their_device_memory():
    """200 create / compute / backprop / (capacity growth) / destroy cycles of the ANI and CFConv handles leave the
    device's free memory where it was (the handles own ~20 device buffers each, some reallocated by check())."""
    import gc
    import os
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("reads the device's free memory: meaningless while other test processes allocate on the same GPU")
    from nnpops_amd.capi import AniSymmetryFunctions, CFConv, CFConvNeighbors
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1200, density=0.2, seed=41)          # dense: rows and records grow
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    w1 = np.zeros((32, 8), np.float32); w2 = np.zeros((32, 32), np.float32); b = np.zeros(32, np.float32)
    x = torch.zeros((len(pos), 32), device=dev)

    def cycle():
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
        radial, angular = sym.compute(tpos, tbox)
        sym.backprop(torch.ones_like(radial), torch.ones_like(angular))
        nb = CFConvNeighbors(len(pos), 5.0, periodic=True)
        nb.build(tpos, tbox, check=True)
        cf = CFConv(len(pos), 32, 8, 5.0, 0.3, "ssp", w1, b, w2, b, periodic=True)
        y = torch.empty_like(x)
        cf.compute(nb, tpos, x, tbox, y)
        del sym, nb, cf

    for _ in range(3):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(200):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)          # allow allocator granularity, not 200 leaked handles


@pytest.mark.parametrize("pad", [8, 5])
def test_strided_rows_write_one_aev_array(pad):
    """nnpops_ani_compute_strided / _backprop_strided: radial and angular parts written into (and their gradients read
    from) ONE [N, W_r + W_a + padding] array, bit-identical to the dense calls (rows that stay 16-byte aligned) or equal
    to rounding (odd strides take the scalar-load kernels); strides below the row width refused."""
    from nnpops_amd.capi import AniSymmetryFunctions, lib, _ptr, _check, NNPOpsHipError
    rf, af = workloads.ani2x_functions()
    pos, species, box = workloads.random_box(1300, seed=61)
    dev = torch.device("cuda:0")
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    radial, angular = sym.compute(tpos, tbox)
    gen = torch.Generator(device=dev).manual_seed(1)
    g_r = torch.randn(radial.shape, device=dev, generator=gen)
    g_a = torch.randn(angular.shape, device=dev, generator=gen)
    grad = sym.backprop(g_r, g_a).clone()
    wr, wa = radial.shape[1], angular.shape[1]
    ld = wr + wa + pad
    aev = torch.full((len(pos), ld), float("nan"), device=dev)
    L = lib()
    _check(L.nnpops_ani_compute_strided(sym._h, _ptr(tpos), _ptr(tbox), aev.data_ptr(), ld, aev.data_ptr() + 4 * wr, ld))
    torch.cuda.synchronize()
    assert torch.equal(aev[:, :wr], radial) and bool(torch.isnan(aev[:, wr + wa:]).all())
    if ld % 4 == 0:
        assert torch.equal(aev[:, wr:wr + wa], angular)
    else:            # (rows that are not 16-byte aligned take the kernel without the LDS row assembly: another order of the same sums)
        assert float((aev[:, wr:wr + wa] - angular).abs().max()) <= 2e-6 * float(angular.abs().max())
    g = torch.zeros((len(pos), ld), device=dev)
    g[:, :wr], g[:, wr:wr + wa] = g_r, g_a
    grad2 = torch.empty_like(grad)
    _check(L.nnpops_ani_backprop_strided(sym._h, g.data_ptr(), ld, g.data_ptr() + 4 * wr, ld, _ptr(grad2)))
    torch.cuda.synchronize()
    if ld % 4 == 0:
        assert torch.equal(grad2, grad)
    else:
        assert float((grad2 - grad).abs().max()) <= 2e-6 * float(grad.abs().max())
    with pytest.raises(NNPOpsHipError, match="row strides"):
        _check(L.nnpops_ani_compute_strided(sym._h, _ptr(tpos), _ptr(tbox), aev.data_ptr(), wr - 1, aev.data_ptr(), ld))


@pytest.mark.parametrize("torchani", [True, False])
@pytest.mark.parametrize("kind", ["irregular", "too_many_factors", "forced_grid", "off_grid_shifts", "other_grid"])
def test_arbitrary_angular_function_lists(monkeypatch, kind, torchani):
    """The reference core evaluates ANY vector of AngularFunction records one by one (CpuANISymmetryFunctions.cpp:153-194).
    Lists that are not a full {(eta,rs)} x {(zeta,thetas)} grid -- here: a grid with holes, a duplicate and shuffled order;
    and a grid with more distinct (zeta, thetas) factors than the factored kernels take -- run on the generic kernels
    instead of being refused; `forced_grid` pushes the ANI-2x grid itself through them ($NNPOPS_ANI_GENERIC)."""
    rng = np.random.default_rng(77)
    rf, af = workloads.ani2x_functions()
    if kind == "irregular":
        keep = rng.permutation(len(af))[:19]
        af = np.concatenate([af[keep], af[keep[:1]]]).astype(np.float32)          # 20 functions, one of them twice
        af[3, 0] = 9.5                                                             # a (eta, rs) pair nobody else has
    elif kind == "too_many_factors":
        zs = [(float(z), float(t)) for z in (1.0, 4.0, 14.1) for t in np.linspace(0.3, 2.8, 4)]      # 12 (zeta, thetas) factors
        af = np.array([[12.5, r, z, t] for r in (0.8, 1.9, 3.0) for z, t in zs], dtype=np.float32)    # 36 functions
    elif kind == "off_grid_shifts":
        # a full 8 x 4 grid of one eta whose eight shifts are NOT equally spaced: the matrix-core forward kernel must not take its
        # recurrence for the radial factors (ani_kernels.h: radial_factors_geo8) but evaluate them one by one
        shifts = (0.8, 1.1, 1.5, 1.8125, 2.2, 2.4875, 2.9, 3.1625)
        af = np.array([[12.5, r, 14.1, t] for r in shifts for t in (0.3927, 1.1781, 1.9635, 2.7489)], dtype=np.float32)
    elif kind == "other_grid":
        # ... and an equally spaced grid with other constants than ANI-2x's (sharper, wider apart, descending): the recurrence's
        # constants come from the list, not from the model
        shifts = 3.3 - 0.36 * np.arange(8)
        af = np.array([[19.0, r, 9.0, t] for r in shifts for t in (0.3927, 1.1781, 1.9635, 2.7489)], dtype=np.float32)
    else:
        monkeypatch.setenv("NNPOPS_ANI_GENERIC", "1")
    pos, species, box = workloads.random_box(420, seed=91)
    _, a_default, _ = _run_case(7, 5.1, 3.5, species, rf, af, pos, box, torchani=torchani)
    mol, sp = workloads.conformer(45, seed=92)
    _run_case(7, 5.1, 3.5, sp, rf, af, mol, None, torchani=torchani)
    if kind in ("off_grid_shifts", "other_grid"):
        # which forward arithmetic ran: against the factor-by-factor kernel ($NNPOPS_ANI_FWD_UNI=0) the off-grid list must give the
        # same bits (it IS that kernel), the grid a different rounding of the same numbers (the recurrence)
        monkeypatch.setenv("NNPOPS_ANI_FWD_UNI", "0")
        _, a_plain, _ = _run_case(7, 5.1, 3.5, species, rf, af, pos, box, torchani=torchani)
        assert np.array_equal(a_default, a_plain) == (kind == "off_grid_shifts")
        monkeypatch.delenv("NNPOPS_ANI_FWD_UNI")
        from nnpops_amd.capi import AniSymmetryFunctions
        what = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=torchani).describe()
        assert what["uniform"] == "1" and what["literal"] == "0" and what["grid"] == ("0" if kind == "off_grid_shifts" else "1")


@pytest.mark.parametrize("fuse", ["0", "1"])
@pytest.mark.parametrize("kind", ["seven_species", "organic"])
def test_published_ani2x_constants_run_the_literal_kernels(monkeypatch, kind, fuse):
    """The ANI-2x parameter set selects forward kernels that carry its derived constants as literals (ani_kernels.h: Ani2xAngular;
    chosen only when every constant nnpops_ani_create derives equals the compiled-in one bit for bit).  Same arithmetic on the same
    numbers: the AEV must equal what the constants-in-registers kernels give ($NNPOPS_ANI_FWD_LITERAL=0) to the last bit or two,
    and both pass the oracle; a set that differs in one parameter must not take them."""
    from nnpops_amd.capi import AniSymmetryFunctions
    monkeypatch.setenv("NNPOPS_ANI_FUSE", fuse)
    rf, af = workloads.ani2x_functions()
    if kind == "seven_species":
        pos, species, box = workloads.random_box(700, seed=5)
    else:
        pos, species, box = workloads.random_box(700, seed=6, n_species=7, species_probs=[0.5, 0.3, 0.1, 0.1, 0, 0, 0])
    assert AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True).describe()["literal"] == "1"
    _, a_lit, _ = _run_case(7, 5.1, 3.5, species, rf, af, pos, box)
    assert AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True, torchani=False).describe()["literal"] == "1"
    _run_case(7, 5.1, 3.5, species, rf, af, pos, box, torchani=False)         # (the paper's angle: its own instantiations of the literal kernels)
    monkeypatch.setenv("NNPOPS_ANI_FWD_LITERAL", "0")
    assert AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True).describe()["literal"] == "0"
    _, a_reg, _ = _run_case(7, 5.1, 3.5, species, rf, af, pos, box)
    np.testing.assert_allclose(a_lit, a_reg, rtol=2e-6, atol=1e-9)
    monkeypatch.delenv("NNPOPS_ANI_FWD_LITERAL")
    af2 = af.copy()
    af2[:, 2] = 14.0                                           # another zeta: not the published set
    assert AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af2, periodic=True).describe()["literal"] == "0"
    _run_case(7, 5.1, 3.5, species, rf, af2, pos, box)


# ---------------------------------------------------------------- the reference's own test molecules
REFERENCE_MOLECULES = ["1hvj", "1hvk", "2iuz", "3hkw", "3hky", "3lka", "3o99", "water"]


def _molecule_weights(shape, k):
    """tests/golden/make_golden_molecules.py::weights (a formula, so the fixture does not carry them)."""
    i, j = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    return (np.round(np.cos(0.37 * i + 1.3 * j + 0.5 + k) * 64) / 64).astype(np.float32)


@pytest.mark.parametrize("surface", ["c_abi", "torch_module"])
@pytest.mark.parametrize("name", REFERENCE_MOLECULES)
def test_reference_test_molecules(golden_dir, name, surface):
    """The inputs of the reference's own Python tests (src/pytorch/TestSymmetryFunctions.py:37-105): seven drug-like
    ligands (real bonded geometries, H C N O S F) in vacuum and the 306-atom water box with its 15 A cell.  Expected
    AEVs and position gradients come from the reference CPU implementation itself (tests/golden/molecules_ref.npz,
    make_golden_molecules.py).  The reference compares energies to 5e-7 and gradients to 5e-3..7.5e-3 RELATIVE PER
    COMPONENT against TorchANI; here: north_star's bars (1e-5 on the energy, 1e-4 of the largest force component),
    element-wise AEV agreement, and the reference's own per-component bar on top."""
    g = np.load(f"{golden_dir}/molecules_ref.npz")
    k = REFERENCE_MOLECULES.index(name)
    pos, species = g[f"{name}_positions"], g[f"{name}_species"]
    cell = g[f"{name}_cell"] if f"{name}_cell" in g else None
    r_ref, a_ref, g_ref = g[f"{name}_radial"], g[f"{name}_angular"], g[f"{name}_grad"]
    wr, wa = _molecule_weights(r_ref.shape, k), _molecule_weights(a_ref.shape, k + 100)
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    if surface == "c_abi":
        from nnpops_amd.capi import AniSymmetryFunctions
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=cell is not None)
        radial, angular = sym.compute(torch.tensor(pos, device=dev), torch.tensor(cell, device=dev) if cell is not None else None)
        grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev))
        torch.cuda.synchronize()
        r, a, gr = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()
    else:
        from test_torch_surface_gpu import FakeConverter, fake_aev_computer, _numbers
        from NNPOps.SymmetryFunctions import TorchANISymmetryFunctions
        numbers = _numbers(species)
        module = TorchANISymmetryFunctions(FakeConverter(), fake_aev_computer(), numbers.cpu()).to(dev)
        tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
        tcell = torch.tensor(cell, device=dev) if cell is not None else None
        pbc = torch.tensor([True, True, True], device=dev) if cell is not None else None
        _, aev = module((torch.tensor(species, device=dev).unsqueeze(0), tpos), tcell, pbc)
        w = torch.tensor(np.concatenate([wr, wa], axis=1), device=dev).unsqueeze(0)
        (aev * w).sum().backward()
        full = aev[0].detach().cpu().numpy()
        r, a, gr = full[:, :r_ref.shape[1]], full[:, r_ref.shape[1]:], tpos.grad[0].cpu().numpy()
    np.testing.assert_allclose(r, r_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    np.testing.assert_allclose(a, a_ref, rtol=AEV_RTOL, atol=AEV_ATOL)
    e_ref = float(r_ref.astype(np.float64).sum() + a_ref.astype(np.float64).sum())
    e = float(r.astype(np.float64).sum() + a.astype(np.float64).sum())
    assert abs(e - e_ref) <= ENERGY_RTOL * abs(e_ref)
    fmax = np.abs(g_ref).max()
    assert np.abs(gr - g_ref).max() <= FORCE_RTOL * fmax, (np.abs(gr - g_ref).max(), fmax)
    # the reference's own bar (TestSymmetryFunctions.py:66-70,102-105): relative error of every gradient component
    big = np.abs(g_ref) > 1e-3 * fmax
    assert np.max(np.abs((gr - g_ref)[big] / g_ref[big])) < 5e-3


def test_backward_classes_regroup_when_an_atom_outgrows_its_class(monkeypatch):
    """The angular backward runs one launch per class of atoms (pair matrix of 32 / 48 / all record slots, nnpops_ani_check).  A later
    frame in which an atom of the 32-slot class has more angular neighbours than its class allows is flagged by the neighbour
    build (overflow bit 3), reported by check() as NNPOPS_ERR_CAPACITY and evaluated again with new classes: forces of BOTH
    frames must be the oracle's, with the classes on and off."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    # many loose molecules (at most ~28 angular neighbours) and ONE compact one that makes the records 64 slots wide
    mols, species = [], []
    for m in range(40):
        p, s = workloads.conformer(60, seed=100 + m)
        mols.append(1.35 * p if m else p)                             # molecule 0 stays compact
        species.append(s)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in mols])]).astype(np.int32)
    species = np.concatenate(species)
    frame_a = np.concatenate(mols).astype(np.float32)
    frame_b = frame_a.copy()
    lo, hi = offsets[7], offsets[8]
    frame_b[lo:hi] = (mols[7] / 1.35).astype(np.float32)              # molecule 7 shrinks: its atoms leave the 32-slot class
    results = {}
    for classes in ("1", "0"):
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASSES", classes)
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASS_MIN", "0")
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASS_ATOMS", "0")          # (by default only systems of 16 384+ atoms are launched by class)
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=False)
        sym.set_molecules(offsets)
        out = []
        for frame in (frame_a, frame_b, frame_a):
            tp = torch.tensor(frame, device=dev)
            radial, angular = sym.compute(tp, None)
            gen = torch.Generator(device=dev).manual_seed(3)
            g_r = torch.randn(radial.shape, device=dev, generator=gen)
            g_a = torch.randn(angular.shape, device=dev, generator=gen)
            out.append((frame, g_r.cpu().numpy(), g_a.cpu().numpy(), sym.backprop(g_r, g_a).cpu().numpy()))
        results[classes] = out
    assert sym.neighbor_stats()[1] > 32                               # (the records really are wider than the 32-slot class)
    for classes, out in results.items():
        for frame, wr, wa, grad in out:
            ref = np.zeros_like(grad)
            for m in range(len(offsets) - 1):                         # per-molecule oracle: molecules never interact
                a, b = offsets[m], offsets[m + 1]
                o = AniOracle(7, 5.1, 3.5, species[a:b], rf, af, periodic=False)
                o.forward(frame[a:b], None)
                ref[a:b] = o.backward(wr[a:b], wa[a:b])
            assert np.abs(grad - ref).max() <= 1e-4 * np.abs(ref).max(), classes
    for (_, _, _, g1), (_, _, _, g0) in zip(results["1"], results["0"]):
        assert np.array_equal(g1, g0)                                 # same kernel arithmetic per atom, whatever the launch it ran in
    # ... and WITHOUT any check between the frames (graph replays, check intervals > 1: ADVICE r04): classes cut for frame A, frame B
    # evaluated unchecked -- the atoms that outgrew their class are taken by the clean-up launch behind the classes, bit 3 of the
    # overflow word says that it happened
    monkeypatch.setenv("NNPOPS_ANI_BWD_CLASSES", "1")
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=False)
    sym.set_molecules(offsets)
    sym.compute(torch.tensor(frame_a, device=dev), None)              # (checked: capacities fitted, classes cut for frame A)
    assert int(sym.describe()["classes"]) >= 2, sym.describe()
    tp = torch.tensor(frame_b, device=dev)
    radial, angular = sym.compute(tp, None, check=False)
    _, wr, wa, checked = results["0"][1]
    grad = sym.backprop(torch.tensor(wr, device=dev), torch.tensor(wa, device=dev)).cpu().numpy()
    assert sym.overflow_word() & 8                                    # an atom did outgrow its class (the builders say so) ...
    assert np.array_equal(grad, checked)                              # ... and its forces are those of the checked evaluation


def test_scattered_leg_forces_stay_in_bounds_on_an_overflowed_frame(monkeypatch):
    """the two-wave angular backward stores leg forces in the RECEIVING atom's row at a slot it looks up in that
    atom's id row.  In a frame that overflows cap_angular the rows are cut short, the pair relation is no longer symmetric, and the
    look-up can miss: the slot must then read "no receiver" (nothing stored), not the bits of fc as an index ~16 GB out of bounds.
    backprop() does run on such frames before the host sees the overflow word (deferred checks, check intervals, graph replays).  Canary:
    the frame is evaluated unchecked with NNPOPS_ANI_SCATTER=1, the device survives, the overflow is reported, and the evaluation
    repeated after check() has grown the rows gives the per-molecule oracle's forces."""
    from nnpops_amd.capi import AniSymmetryFunctions
    monkeypatch.setenv("NNPOPS_ANI_SCATTER", "1")
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    mols, species = [], []
    for m in range(48):
        p, s = workloads.conformer(60, seed=500 + m)
        mols.append(p)
        species.append(s)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in mols])]).astype(np.int32)
    species = np.concatenate(species)
    loose = np.concatenate([1.5 * p for p in mols]).astype(np.float32)         # capacities are fitted to this frame ...
    tight = np.concatenate([0.8 * p for p in mols]).astype(np.float32)         # ... and this one overflows them (1.9 x the radius ratio)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=False)
    sym.set_molecules(offsets)
    r, a = sym.compute(torch.tensor(loose, device=dev), None)                  # checked: rows fitted to the loose frame
    gen = torch.Generator(device=dev).manual_seed(5)
    g_r = torch.randn(r.shape, device=dev, generator=gen)
    g_a = torch.randn(a.shape, device=dev, generator=gen)
    sym.backprop(g_r, g_a)
    cap_before = sym.neighbor_stats()[1]
    tp = torch.tensor(tight, device=dev)
    sym.compute(tp, None, check=False)                                          # overflowed, nobody has looked yet
    sym.backprop(g_r, g_a)                                                      # must not fault or write out of bounds
    torch.cuda.synchronize()
    assert sym.overflow_word() & 1, (sym.overflow_word(), cap_before)          # the scenario happened: an atom outgrew its records
    r, a = sym.compute(tp, None)                                                # check() grows the rows, evaluates again
    grad = sym.backprop(g_r, g_a).cpu().numpy()
    wr, wa = g_r.cpu().numpy(), g_a.cpu().numpy()
    ref = np.zeros_like(grad)
    for m in range(len(offsets) - 1):
        lo, hi = offsets[m], offsets[m + 1]
        o = AniOracle(7, 5.1, 3.5, species[lo:hi], rf, af, periodic=False)
        o.forward(tight[lo:hi], None)
        ref[lo:hi] = o.backward(wr[lo:hi], wa[lo:hi])
    assert np.abs(grad - ref).max() <= FORCE_RTOL * np.abs(ref).max()


def test_pair_walk_of_the_builders_does_not_change_the_lists(monkeypatch):
    """Round 5: the builders' triple loop walks the pairs of an atom as a folded rectangle (decode_pair_folded) or row-major (handles
    whose lists exceed the Infinity Cache); either walk writes the SAME bucket-major list, so AEV and forces must agree bit for bit --
    on a liquid (cell-grid builder) and on a batch of molecules (all-pairs builder, atoms with even and odd neighbour counts)."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    pos, species, box = workloads.random_box(3000, density=0.1, seed=21, n_species=7)
    mols = [workloads.conformer(50 + m, seed=300 + m) for m in range(12)]
    mpos = np.concatenate([m[0] for m in mols]).astype(np.float32)
    mspecies = np.concatenate([m[1] for m in mols]).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(m[0]) for m in mols])]).astype(np.int32)
    results = {}
    for walk in ("0", "1"):
        monkeypatch.setenv("NNPOPS_ANI_TRI_ROW_MAJOR", walk)
        out = []
        for p, s, b, off in ((pos, species, box, None), (mpos, mspecies, None, offsets)):
            sym = AniSymmetryFunctions(7, 5.1, 3.5, s, rf, af, periodic=b is not None)
            if off is not None:
                sym.set_molecules(off)
            tp = torch.tensor(p, device=dev)
            tb = torch.tensor(b, device=dev) if b is not None else None
            radial, angular = sym.compute(tp, tb)
            gen = torch.Generator(device=dev).manual_seed(9)
            g_r = torch.randn(radial.shape, device=dev, generator=gen)
            g_a = torch.randn(angular.shape, device=dev, generator=gen)
            out.append((radial.cpu().numpy(), angular.cpu().numpy(), sym.backprop(g_r, g_a).cpu().numpy()))
        results[walk] = out
    for a, b in zip(results["0"], results["1"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert np.abs(results["0"][0][1]).max() > 0 and np.abs(results["0"][1][2]).max() > 0


def test_leg_forces_in_the_receivers_rows(monkeypatch):
    """Round 5: where every launch of the angular backward runs two waves per atom (dense systems), the second wave looks up the slot
    of its centre atom in the records of the atom on every leg and the leg force is stored in the RECEIVING atom's row, which the
    radial backward reads instead of searching its neighbours' id rows ($NNPOPS_ANI_SCATTER forces either way).  Same forces as the
    gathering path up to the order of one fp32 sum, the oracle's forces within the usual bar, bitwise reproducible -- on a batch of
    molecules (records of 64 slots, atoms with a single angular neighbour among them) and with the backward launched by classes."""
    from nnpops_amd.capi import AniSymmetryFunctions
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda:0")
    mols = [workloads.conformer(40 + 3 * m, seed=500 + m) for m in range(10)]
    mols.append((np.array([[0.0, 0, 0], [1.1, 0, 0], [9.0, 0, 0]], np.float32), np.array([1, 0, 3], np.int32)))      # two bonded atoms + a loner
    pos = np.concatenate([m[0] for m in mols]).astype(np.float32)
    species = np.concatenate([m[1] for m in mols]).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(m[0]) for m in mols])]).astype(np.int32)
    tp = torch.tensor(pos, device=dev)
    grads = {}
    for scatter, classes in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("NNPOPS_ANI_SCATTER", scatter)
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASSES", classes)
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASS_MIN", "0")
        monkeypatch.setenv("NNPOPS_ANI_BWD_CLASS_ATOMS", "0")
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=False)
        sym.set_molecules(offsets)
        sym.compute(tp, None)
        radial, angular = sym.compute(tp, None)                      # (classes and the two-wave decision are in place after a check)
        gen = torch.Generator(device=dev).manual_seed(4)
        g_r = torch.randn(radial.shape, device=dev, generator=gen)
        g_a = torch.randn(angular.shape, device=dev, generator=gen)
        a = sym.backprop(g_r, g_a).cpu().numpy()
        b = sym.backprop(g_r, g_a).cpu().numpy()
        assert np.array_equal(a, b)                                   # reproducible
        assert sym.describe()["scatter"] == scatter, sym.describe()   # (the path asked for is the one that ran)
        grads[(scatter, classes)] = a
        wr, wa = g_r.cpu().numpy(), g_a.cpu().numpy()
    ref = np.zeros_like(grads[("0", "0")])
    for m in range(len(offsets) - 1):
        lo, hi = offsets[m], offsets[m + 1]
        o = AniOracle(7, 5.1, 3.5, species[lo:hi], rf, af, periodic=False)
        o.forward(pos[lo:hi], None)
        ref[lo:hi] = o.backward(wr[lo:hi], wa[lo:hi])
    fmax = np.abs(ref).max()
    for key, g in grads.items():
        assert np.abs(g - ref).max() <= 1e-4 * fmax, key
    assert np.abs(grads[("1", "0")] - grads[("0", "0")]).max() <= 2e-6 * fmax
    assert np.array_equal(grads[("1", "0")], grads[("1", "1")])      # same arithmetic per atom, whatever launch it ran in
