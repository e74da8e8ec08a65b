"""numpy restatement of ``neighbors::getNeighborPairs`` -- TEST INFRASTRUCTURE ONLY.

Follows the reference CPU implementation (src/pytorch/neighbors/getNeighborPairsCPU.cpp:19-100)
for the values, and the reference CUDA implementation (getNeighborPairsCUDA.cu:31-78,103-164)
for the *device* bookkeeping, which differs in two documented ways:

  * pair k of the lower triangle <-> (row, col<row), ``tril_indices(N, -1)`` order
    (CPU.cpp:57-61 closed form; here via numpy tril_indices, same order);
  * delta = pos[row] - pos[col]  (direction neighbors[1] -> neighbors[0]);
  * triclinic wrap, one round() per axis in z, y, x order using the *diagonal* element of each
    box vector (CPU.cpp:66-68);
  * membership: distance <= cutoff (CPU.cpp:81; CUDA rejects distance^2 > cutoff^2, CUDA.cu:66);
  * max_num_pairs == -1: every one of the N(N-1)/2 slots is kept, non-members become
    (-1, NaN, NaN) (CPU.cpp:72-78);
  * max_num_pairs  > 0 : members are compacted in pair order and padded with (-1, NaN, NaN)
    up to max_num_pairs (CPU.cpp:80-96).
      - ``device_semantics=False`` (CPU reference): nothing is truncated when more members are
        found than max_num_pairs, and num_pairs reports the *padded* length (CPU.cpp:97-98);
      - ``device_semantics=True`` (CUDA reference, the one the HIP path mirrors): outputs always
        have exactly max_num_pairs slots, surplus members are dropped, and num_pairs is the true
        number found (CUDA.cu:68-78,163).

Rounding note: torch::round on CPU is half-to-even while the CUDA kernel's round() is
half-away-from-zero; the two differ only for components that are exactly (k+1/2) box lengths,
a measure-zero case the tests avoid.  This file follows numpy/torch CPU (half-to-even).

Parity status: PINNED against fixtures generated from the reference CPU op itself
(tests/golden/neighbors_*.npz, script tests/golden/make_golden_torch_ref.py) and against the four
worked examples in the reference docstring (src/pytorch/neighbors/getNeighborPairs.py:104-138).
"""
import numpy as np


def _wrap(deltas, box):
    box = np.asarray(box, dtype=deltas.dtype)
    for axis in (2, 1, 0):
        scale = np.round(deltas[:, axis] / box[axis, axis])
        deltas = deltas - np.outer(scale, box[axis])
    return deltas


def neighbor_pairs_oracle(positions, cutoff, max_num_pairs=-1, box=None, device_semantics=True):
    """-> neighbors int32[2,P], deltas[P,3], distances[P], num_pairs (python int)."""
    pos = np.asarray(positions)
    assert pos.ndim == 2 and pos.shape[1] == 3
    n = pos.shape[0]
    rows, cols = np.tril_indices(n, -1)
    rows = rows.astype(np.int32)
    cols = cols.astype(np.int32)
    deltas = pos[rows] - pos[cols]
    if box is not None and np.asarray(box).size:
        deltas = _wrap(deltas, box)
    dist = np.sqrt((deltas * deltas).sum(axis=1)).astype(pos.dtype)
    cutoff = pos.dtype.type(cutoff)
    neighbors = np.vstack([rows, cols]).astype(np.int32).reshape(2, -1)
    if max_num_pairs == -1:
        out = dist > cutoff
        neighbors = neighbors.copy()
        neighbors[:, out] = -1
        deltas = deltas.copy()
        deltas[out] = np.nan
        dist = dist.copy()
        dist[out] = np.nan
        return neighbors, deltas, dist, int(dist.shape[0])
    keep = dist <= cutoff
    found = int(keep.sum())
    neighbors, deltas, dist = neighbors[:, keep], deltas[keep], dist[keep]
    if device_semantics and found > max_num_pairs:
        neighbors, deltas, dist = neighbors[:, :max_num_pairs], deltas[:max_num_pairs], dist[:max_num_pairs]
    pad = max_num_pairs - dist.shape[0]
    if pad > 0:
        neighbors = np.hstack([neighbors, np.full((2, pad), -1, np.int32)])
        deltas = np.vstack([deltas, np.full((pad, 3), np.nan, deltas.dtype)])
        dist = np.hstack([dist, np.full(pad, np.nan, dist.dtype)])
    num_pairs = found if device_semantics else int(dist.shape[0])
    return neighbors, deltas, dist, num_pairs


def neighbor_pairs_backward_oracle(n_atoms, neighbors, deltas, distances, grad_deltas, grad_distances):
    """d(loss)/d(positions) given d/d(deltas) and d/d(distances)
    (reference getNeighborPairsCUDA.cu:80-101: +grad on neighbors[0], -grad on neighbors[1],
    slots holding -1 skipped)."""
    gp = np.zeros((n_atoms, 3), dtype=deltas.dtype)
    valid = neighbors[0] >= 0
    g = grad_deltas[valid] + (deltas[valid] / distances[valid][:, None]) * grad_distances[valid][:, None]
    np.add.at(gp, neighbors[0][valid], g)
    np.add.at(gp, neighbors[1][valid], -g)
    return gp
