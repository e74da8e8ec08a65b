"""getNeighborPairs -- pairs of atoms closer than a cutoff (reference src/pytorch/neighbors/getNeighborPairs.py:8-147)."""
from typing import Optional, Tuple

import torch
from torch import Tensor, empty

from .. import torch_binding

torch_binding.load()


def getNeighborPairs(positions: Tensor, cutoff: float, max_num_pairs: int = -1, box_vectors: Optional[Tensor] = None,
                     check_errors: bool = False) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Returns ``(neighbors, deltas, distances, number_found_pairs)``.

    positions [num_atoms, 3] float32/float64 on the GPU; ``max_num_pairs = -1`` returns one slot per pair of
    the lower triangle (slots beyond the cutoff hold -1 / NaN / NaN), ``max_num_pairs > 0`` a compacted list of
    that many slots; ``box_vectors`` [3, 3] (rows a, b, c in reduced form, each at least 2*cutoff wide) enables
    the triclinic minimum-image wrap; ``deltas`` point from ``neighbors[1]`` to ``neighbors[0]``.
    ``number_found_pairs`` is the true number of pairs within the cutoff and may exceed ``max_num_pairs``; with
    ``check_errors=True`` that raises instead (this synchronises and cannot be captured in a graph).

    Unlike the reference's GPU path the compacted list comes out in a deterministic order (grouped by
    ``neighbors[0]``, ascending).
    """
    if box_vectors is None:
        box_vectors = empty((0, 0), device=positions.device, dtype=positions.dtype)
    return torch.ops.neighbors.getNeighborPairs(positions, cutoff, max_num_pairs, box_vectors, check_errors)
