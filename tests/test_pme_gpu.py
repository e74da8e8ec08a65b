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
