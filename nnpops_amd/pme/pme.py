"""PME -- the direct-space (short-range) term of Particle Mesh Ewald on the HIP neighbour-pair list.

Mirrors the reference class (src/pytorch/pme/pme.py:5-165): same constructor arguments and checks, same
``compute_direct(positions, charges, cutoff, box_vectors, max_num_pairs)`` contract, same unit convention (the value of
Coulomb's constant sets the units), same exclusion semantics (only the un-wrapped copy of an excluded pair is left out, and
the erf() part that reciprocal space cannot leave out is subtracted here).  ``compute_direct`` = getNeighborPairs +
``torch.ops.pme.pme_direct``, differentiable w.r.t. positions and charges (first derivatives only).

The reciprocal-space term (charge spreading onto a grid + 3-D FFTs, src/pytorch/pme/pmeCUDA.cu:102-235) is outside the
scope of this build: ``compute_reciprocal`` raises.
"""
import torch

from ..neighbors import getNeighborPairs


class PME:
    def __init__(self, gridx: int, gridy: int, gridz: int, order: int, alpha: float, coulomb: float, exclusions: torch.Tensor):
        # the reference's argument checks (pme.py:75-85)
        if gridx < 1 or gridy < 1 or gridz < 1:
            raise ValueError('The grid dimensions must be positive')
        if order < 1:
            raise ValueError('order must be positive')
        if alpha <= 0:
            raise ValueError('alpha must be positive')
        if coulomb <= 0:
            raise ValueError('coulomb must be positive')
        if exclusions.dim() != 2:
            raise ValueError('exclusions must be 2D')
        self.gridx, self.gridy, self.gridz, self.order = gridx, gridy, gridz, order
        self.alpha, self.coulomb = alpha, coulomb
        # The table must be symmetric -- j in row i exactly when i is in row j (the reference documents it, pme.py:66-73, and its
        # kernel visits an excluded pair from the row of the higher index only).  The owner-computes HIP kernel has every atom take
        # the terms of its excluded pairs from ITS OWN row (nnpops_hip.h: nnpops_pme_direct): a one-sided table would give
        # derivatives that differ between the device and the host path without any error.  Checked here, once.
        ex = exclusions.to(torch.int64).cpu()
        n, width = ex.shape
        if width > 0 and n > 0:
            if bool(((ex >= n) | (ex < -1)).any()):
                raise ValueError('exclusions must hold atom indices or -1')
            rows = torch.arange(n).unsqueeze(1).expand(n, width)
            valid = ex >= 0
            pairs = torch.stack([rows[valid], ex[valid]], dim=1)
            keys = set((pairs[:, 0] * n + pairs[:, 1]).tolist())
            if any((j * n + i) not in keys for i, j in pairs.tolist()):
                raise ValueError('exclusions must be symmetric: if atom j is excluded from atom i, atom i must be excluded from atom j')
        # rows sorted in descending order: the kernels stop scanning a row at the first entry below the partner (pme.py:93)
        self.exclusions, _ = torch.sort(exclusions.to(torch.int32), descending=True)

    def compute_direct(self, positions: torch.Tensor, charges: torch.Tensor, cutoff: float, box_vectors: torch.Tensor,
                       max_num_pairs: int = -1):
        """Energy of the direct-space term (a 0-dim tensor)."""
        if positions.dim() != 2 or positions.shape[1] != 3:
            raise ValueError('positions must have shape (atoms, 3)')
        if charges.dim() != 1:
            raise ValueError('charges must be 1D')
        if positions.shape[0] != self.exclusions.shape[0] or charges.shape[0] != self.exclusions.shape[0]:
            raise ValueError('positions, charges, and exclusions must all have the same length')
        if box_vectors.dim() != 2 or box_vectors.shape[0] != 3 or box_vectors.shape[1] != 3:
            raise ValueError('box_vectors must have shape (3, 3)')
        if cutoff <= 0:
            raise ValueError('cutoff must be positive')
        neighbors, deltas, distances, _ = getNeighborPairs(positions, cutoff, max_num_pairs, box_vectors)
        self.exclusions = self.exclusions.to(positions.device)
        return torch.ops.pme.pme_direct(positions, charges, neighbors, deltas, distances, self.exclusions, self.alpha, self.coulomb)

    def compute_reciprocal(self, positions: torch.Tensor, charges: torch.Tensor, box_vectors: torch.Tensor):
        raise RuntimeError("the reciprocal-space term of PME (charge spreading + FFT) is not part of this build: only the "
                           "direct-space term, the consumer of getNeighborPairs, is (see DESIGN.md, scope)")
