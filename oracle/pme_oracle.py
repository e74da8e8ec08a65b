"""oracle/pme_oracle.py -- TEST INFRASTRUCTURE ONLY (parity checker).

numpy restatement of the reference's direct-space PME (reference src/pytorch/pme/pmeCPU.cpp:75-163): for every listed pair
that is not excluded the erfc() Coulomb term, for every excluded pair (un-wrapped, once) minus the erf() term, and the
derivatives with respect to positions and charges.  float32 arithmetic per pair like the reference, double accumulation of
the energy.  Parity status: PINNED against outputs of the reference's own CPU op and the OpenMM numbers held by the
reference's tests (tests/golden/pme_ref.npz, made by tests/golden/make_golden_pme.py).
"""
import numpy as np
from scipy.special import erf, erfc

TWO_OVER_SQRT_PI = np.float32(1.12837916709551257390)


def pme_direct_oracle(positions, charges, neighbors, deltas, distances, exclusions, alpha, coulomb):
    """-> (energy float64, dE/dpositions [N,3] float32, dE/dcharges [N] float32).  `exclusions` rows sorted descending."""
    pos = np.asarray(positions, np.float32)
    q = np.asarray(charges, np.float32)
    nb = np.asarray(neighbors, np.int64)
    dl, ds = np.asarray(deltas, np.float32), np.asarray(distances, np.float32)
    ex = np.asarray(exclusions, np.int64).reshape(len(pos), -1)
    alpha, coulomb = np.float32(alpha), np.float32(coulomb)
    pd = np.zeros((len(pos), 3), np.float64)
    cd = np.zeros(len(pos), np.float64)
    a1, a2 = nb[0], nb[1]
    keep = a1 > -1                                                         # ref :103
    if ex.shape[1]:
        keep &= ~(ex[np.maximum(a1, 0)] == a2[:, None]).any(1)              # ref :104-107
    a1, a2, r, d = a1[keep], a2[keep], ds[keep], dl[keep]
    inv_r, ar = np.float32(1) / r, alpha * r
    er, ex2 = erfc(ar).astype(np.float32), np.exp(-ar * ar).astype(np.float32)
    pre = coulomb * inv_r
    energy = float(np.sum((pre * er * q[a1] * q[a2]).astype(np.float64)))   # ref :115
    np.add.at(cd, a1, pre * er * q[a2])
    np.add.at(cd, a2, pre * er * q[a1])
    dedr = pre * q[a1] * q[a2] * (er + ar * ex2 * TWO_OVER_SQRT_PI) * inv_r * inv_r      # ref :118
    np.add.at(pd, a1, -dedr[:, None] * d)
    np.add.at(pd, a2, dedr[:, None] * d)
    for i in range(len(pos)):                                               # ref :128-150
        for j in ex[i]:
            if j <= i:
                break
            dr = pos[i] - pos[j]
            r = np.float32(np.sqrt(np.sum(dr * dr, dtype=np.float32)))
            inv_r, ar = np.float32(1) / r, alpha * r
            er, ex2 = np.float32(erf(ar)), np.float32(np.exp(-ar * ar))
            pre = coulomb * inv_r
            energy -= float(pre * er * q[i] * q[j])
            cd[i] -= pre * er * q[j]
            cd[j] -= pre * er * q[i]
            dedr = pre * q[i] * q[j] * (er - ar * ex2 * TWO_OVER_SQRT_PI) * inv_r * inv_r
            pd[i] += dedr * dr
            pd[j] -= dedr * dr
    return energy, pd.astype(np.float32), cd.astype(np.float32)
