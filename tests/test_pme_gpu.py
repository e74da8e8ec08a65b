"""Direct-space PME on the MI355X (C ABI and torch surface) against the oracle and the reference-made vectors."""
import numpy as np
import pytest
import torch

from nnpops_amd import workloads
from oracle import pme_direct_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sorted_excl(excl):
    return -np.sort(-excl.astype(np.int64), axis=1) if excl.shape[1] else excl.astype(np.int64)


def test_c_abi_reproduces_the_reference_vectors(golden_dir):
    from nnpops_amd import capi
    g = np.load(f"{golden_dir}/pme_ref.npz")
    for k in range(int(g["num_cases"])):
        c = {name[len(f"c{k}_"):]: g[name] for name in g.files if name.startswith(f"c{k}_")}
        t = lambda a, dt=torch.float32: torch.tensor(a, dtype=dt, device=DEV)
        e, pd, cd = capi.pme_direct(t(c["positions"]), t(c["charges"]), t(c["neighbors"], torch.int32), t(c["deltas"]), t(c["distances"]),
                                    t(_sorted_excl(c["exclusions"]), torch.int32), float(c["alpha"]), float(c["coulomb"]))
        assert abs(float(e) - float(c["energy"])) <= 1e-5 * max(abs(float(c["energy"])), 1.0), k
        assert np.abs(pd.cpu().numpy() - c["pos_grad"]).max() <= 1e-4 * np.abs(c["pos_grad"]).max()
        assert np.abs(cd.cpu().numpy() - c["charge_grad"]).max() <= 1e-4 * np.abs(c["charge_grad"]).max()


def _salt_box(n, seed, max_excl):
    rng = np.random.default_rng(seed)
    pos, _, box = workloads.random_box(n, density=0.05, seed=seed)            # 0.05 atoms/A^3, in Angstrom
    charges = rng.choice([-1.0, 1.0], size=n).astype(np.float32) * rng.uniform(0.2, 1.0, n).astype(np.float32)
    excl = -np.ones((n, max_excl), np.int64)
    fill = np.zeros(n, int)
    order = rng.permutation(n)
    for a, b in zip(order[0::2], order[1::2]):                              # disjoint pairs + chains: symmetric by construction
        for i, j in ((a, b), (b, a)):
            if fill[i] < max_excl:
                excl[i, fill[i]] = j; fill[i] += 1
    return pos, charges, box, _sorted_excl(excl)


@pytest.mark.parametrize("n,compact", [(1500, False), (4000, True)])
def test_torch_surface_against_the_oracle(n, compact):
    """PME.compute_direct on device tensors (getNeighborPairs + pme_direct + autograd) against the numpy oracle fed with the
    SAME pair list: energy 1e-5, derivatives 1e-4 of the largest component."""
    from NNPOps.pme import PME
    from NNPOps.neighbors import getNeighborPairs
    pos, charges, box, excl = _salt_box(n, seed=5 + n, max_excl=2)
    cutoff, alpha, coulomb = 9.0, 0.35, 332.063713
    tpos = torch.tensor(pos, device=DEV, requires_grad=True)
    tq = torch.tensor(charges, device=DEV, requires_grad=True)
    tbox = torch.tensor(box, device=DEV)
    pme = PME(32, 32, 32, 5, alpha, coulomb, torch.tensor(excl, dtype=torch.int32))
    mnp = int(2 * n * 0.05 * 4.19 * cutoff ** 3 / 2) if compact else -1      # ~2x the expected number of pairs
    e = pme.compute_direct(tpos, tq, cutoff, tbox, mnp)
    e.backward()
    nb, dl, ds, found = getNeighborPairs(tpos.detach(), cutoff, mnp, tbox)
    assert not compact or int(found) <= mnp
    e_ref, pd_ref, cd_ref = pme_direct_oracle(pos, charges, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(), excl, alpha, coulomb)
    terms = float(np.abs(cd_ref * charges).sum())              # sum of |pair energies| scale: the energy itself cancels
    assert abs(float(e) - e_ref) <= 1e-5 * max(terms, abs(e_ref))
    assert np.abs(tpos.grad.cpu().numpy() - pd_ref).max() <= 1e-4 * np.abs(pd_ref).max()
    assert np.abs(tq.grad.cpu().numpy() - cd_ref).max() <= 1e-4 * np.abs(cd_ref).max()


def test_full_size_properties_100k_atoms():
    """BASELINE config 5's box (100 000 atoms, 5.2 A list, ~2.9 M pairs): Newton's third law, a bitwise reproducible energy
    and pair-order independence (the reversed list must give the same physics)."""
    from nnpops_amd import capi
    n = 100000
    pos, _, box = workloads.random_box(n, density=0.1, seed=6)
    rng = np.random.default_rng(2)
    q = torch.tensor(rng.normal(0, 0.3, n).astype(np.float32), device=DEV)
    tpos, tbox = torch.tensor(pos, device=DEV), torch.tensor(box, device=DEV)
    nb, dl, ds, found = capi.neighbor_pairs_forward(tpos, 5.2, 32 * n, tbox)
    excl = torch.full((n, 1), -1, dtype=torch.int32, device=DEV)
    runs = [capi.pme_direct(tpos, q, nb, dl, ds, excl, 0.6, 332.063713) for _ in range(2)]
    assert torch.equal(runs[0][0], runs[1][0])                                   # energy: fixed summation order
    e, pd, cd = runs[0]
    fmax = float(pd.abs().max())
    assert float(pd.double().sum(0).abs().max()) <= 1e-3 * fmax                  # forces sum to zero
    k = int(found)
    idx = torch.arange(k - 1, -1, -1, device=DEV)
    e2, pd2, cd2 = capi.pme_direct(tpos, q, nb[:, :k][:, idx].contiguous(), dl[:k][idx].contiguous(), ds[:k][idx].contiguous(), excl, 0.6, 332.063713)
    assert abs(float(e2) - float(e)) <= 1e-6 * abs(float(e)) + 1e-3
    assert float((pd2 - pd).abs().max()) <= 1e-4 * fmax and float((cd2 - cd).abs().max()) <= 1e-4 * float(cd.abs().max())


@pytest.mark.parametrize("n,max_excl", [(3000, 2), (12000, 0)])
def test_indexed_path_matches_the_oracle_and_the_delivering_path(n, max_excl):
    """Round 6: with the list's transposed index (neighbor_pairs_build_index) the direct-space sums are one streaming pass + an
    owner-computes gather (nnpops_pme_direct_indexed): the oracle's numbers under the same bars, the delivering path's (one returning
    atomic per pair, order-independent quantised sums) to rounding, the same bits on every call; through the torch surface the op picks
    the index up from the getNeighborPairs call that made the list (same storage, same version) -- bit for bit the C ABI's indexed
    result -- and a list it does not know takes the path that assumes nothing."""
    from nnpops_amd import capi
    from NNPOps.neighbors import getNeighborPairs
    pos, charges, box, excl = _salt_box(n, seed=70 + n, max_excl=max_excl)
    cutoff, alpha, coulomb = 8.0, 0.35, 332.063713
    tpos, tq, tbox = torch.tensor(pos, device=DEV), torch.tensor(charges, device=DEV), torch.tensor(box, device=DEV)
    texcl = torch.tensor(excl, dtype=torch.int32, device=DEV) if max_excl else torch.zeros((n, 0), dtype=torch.int32, device=DEV)
    slots = int(2 * n * 0.05 * 4.19 * cutoff ** 3 / 2)
    nb, dl, ds, found = capi.neighbor_pairs_forward(tpos, cutoff, slots, tbox)
    assert 0 < int(found) < slots
    index = capi.neighbor_pairs_build_index(n, nb)
    e_i, pd_i, cd_i = capi.pme_direct(tpos, tq, nb, dl, ds, texcl, alpha, coulomb, index=index)
    for _ in range(2):
        again = capi.pme_direct(tpos, tq, nb, dl, ds, texcl, alpha, coulomb, index=index)
        assert all(torch.equal(a, b) for a, b in zip(again, (e_i, pd_i, cd_i)))
    e_d, pd_d, cd_d = capi.pme_direct(tpos, tq, nb, dl, ds, texcl, alpha, coulomb)
    e_ref, pd_ref, cd_ref = pme_direct_oracle(pos, charges, nb.cpu().numpy(), dl.cpu().numpy(), ds.cpu().numpy(),
                                              excl if max_excl else np.zeros((n, 0), np.int64), alpha, coulomb)
    terms = float(np.abs(cd_ref * charges).sum())
    assert abs(float(e_i) - e_ref) <= 1e-5 * max(terms, abs(e_ref)) and abs(float(e_i) - float(e_d)) <= 1e-6 * max(terms, abs(e_ref))
    assert np.abs(pd_i.cpu().numpy() - pd_ref).max() <= 1e-4 * np.abs(pd_ref).max()
    assert np.abs(cd_i.cpu().numpy() - cd_ref).max() <= 1e-4 * np.abs(cd_ref).max()
    assert float((pd_i - pd_d).abs().max()) <= 2e-6 * float(pd_d.abs().max()) and float((cd_i - cd_d).abs().max()) <= 2e-6 * float(cd_d.abs().max())
    # the torch surface: the list of a differentiable getNeighborPairs call carries its index to pme_direct
    p = tpos.clone().requires_grad_(True)
    q = tq.clone().requires_grad_(True)
    t_nb, t_dl, t_ds, _ = getNeighborPairs(p, cutoff, slots, tbox)
    assert torch.equal(t_nb, nb)
    energy = torch.ops.pme.pme_direct(p, q, t_nb, t_dl, t_ds, texcl, alpha, coulomb)
    energy.backward()
    # (pme_direct's own derivatives only: the list's deltas / distances enter it as data, exactly as in the reference, pme.cpp)
    assert torch.equal(energy.detach().reshape(()), e_i.reshape(())) and torch.equal(q.grad, cd_i)
    shuffled = torch.randperm(int(found), device=DEV)
    nb_s = nb.clone()
    nb_s[:, :int(found)] = nb[:, :int(found)][:, shuffled]
    dl_s, ds_s = dl.clone(), ds.clone()
    dl_s[:int(found)] = dl[:int(found)][shuffled]
    ds_s[:int(found)] = ds[:int(found)][shuffled]
    e_s = torch.ops.pme.pme_direct(tpos, tq, nb_s, dl_s, ds_s, texcl, alpha, coulomb)       # a list of unknown origin: correct all the same
    assert abs(float(e_s) - e_ref) <= 1e-5 * max(terms, abs(e_ref))


def _nine_charges():
    rng = np.random.default_rng(11)
    pos = torch.tensor((3 * rng.random((9, 3)) - 1).astype(np.float32), device=DEV)
    charges = torch.tensor([(i - 4) * 0.1 for i in range(9)], dtype=torch.float32, device=DEV)
    box = torch.tensor([[1, 0, 0], [0, 1.1, 0], [0, 0, 1.2]], dtype=torch.float32, device=DEV)
    return pos, charges, box


def test_double_derivative_raises():
    """Second derivatives are not implemented and must say so (reference pme/TestPme.py:297-318, direct-space half)."""
    from NNPOps.pme import PME
    pos, charges, box = _nine_charges()
    pos.requires_grad_(); charges.requires_grad_()
    pme = PME(14, 16, 15, 5, 5.0, 138.935, torch.zeros(9, 0, dtype=torch.int32))
    edir = pme.compute_direct(pos, charges, 0.5, box)
    ddir = torch.autograd.grad(edir, pos, retain_graph=True)              # the reference's own sequence: first derivative ...
    with pytest.raises(Exception):
        torch.autograd.grad(ddir[0].sum(), pos, retain_graph=True)      # ... is a leaf, nothing to differentiate
    with pytest.raises(Exception):
        torch.autograd.grad(ddir[0].sum(), charges, retain_graph=True)
    # a RECORDED backward pass (force matching, Hessians) would silently treat the PME part of d(force)/dx as zero: refused
    with pytest.raises(RuntimeError, match="second derivatives are not implemented"):
        torch.autograd.grad(edir, pos, retain_graph=True, create_graph=True)
    with pytest.raises(RuntimeError, match="second derivatives are not implemented"):
        torch.autograd.grad(edir, charges, retain_graph=True, create_graph=True)
    with pytest.raises(RuntimeError, match="reciprocal"):
        pme.compute_reciprocal(pos, charges, box)


def test_scripted_module_calls_the_direct_space_op():
    """torch.jit.script of a module around getNeighborPairs + torch.ops.pme.pme_direct (reference pme/TestPme.py test_jit,
    direct-space half): same energy and forces as the Python class."""
    from NNPOps.pme import PME
    from NNPOps.neighbors import getNeighborPairs
    pos, charges, box = _nine_charges()
    excl = torch.sort(torch.tensor([[1], [0], [-1], [-1], [-1], [-1], [-1], [-1], [-1]], dtype=torch.int32), descending=True)[0].to(DEV)

    class Direct(torch.nn.Module):
        def __init__(self, exclusions: torch.Tensor, alpha: float, coulomb: float):
            super().__init__()
            self.exclusions, self.alpha, self.coulomb = exclusions, alpha, coulomb

        def forward(self, positions: torch.Tensor, charges: torch.Tensor, box_vectors: torch.Tensor) -> torch.Tensor:
            neighbors, deltas, distances, _ = getNeighborPairs(positions, 0.5, -1, box_vectors)
            return torch.ops.pme.pme_direct(positions, charges, neighbors, deltas, distances, self.exclusions, self.alpha, self.coulomb)

    scripted = torch.jit.script(Direct(excl, 5.0, 138.935))
    p1 = pos.clone().requires_grad_(); p2 = pos.clone().requires_grad_()
    e1 = scripted(p1, charges, box); e1.backward()
    e2 = PME(14, 16, 15, 5, 5.0, 138.935, excl.cpu()).compute_direct(p2, charges, 0.5, box); e2.backward()
    assert torch.equal(e1, e2) and torch.allclose(p1.grad, p2.grad, rtol=1e-6, atol=1e-6)


def test_direct_space_replays_as_a_hip_graph():
    """compute_direct + backward captured into one HIP graph and replayed on moved atoms (reference test_cuda_graph)."""
    from NNPOps.pme import PME
    pos, charges, box = _nine_charges()
    pme = PME(14, 16, 15, 5, 5.0, 138.935, torch.zeros(9, 0, dtype=torch.int32))
    static_pos = pos.clone().requires_grad_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                 # warm-up outside the capture
        for _ in range(2):
            e = pme.compute_direct(static_pos, charges, 0.5, box, max_num_pairs=64)
            e.backward()
            static_pos.grad = None
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        e = pme.compute_direct(static_pos, charges, 0.5, box, max_num_pairs=64)
        e.backward()
    moved = pos + 0.05
    with torch.no_grad():
        static_pos.copy_(moved)
    graph.replay()
    torch.cuda.synchronize()
    ref_pos = moved.clone().requires_grad_()
    e_ref = pme.compute_direct(ref_pos, charges, 0.5, box, max_num_pairs=64)
    e_ref.backward()
    assert torch.allclose(e, e_ref, rtol=1e-6, atol=1e-6) and torch.allclose(static_pos.grad, ref_pos.grad, rtol=1e-5, atol=1e-5)
