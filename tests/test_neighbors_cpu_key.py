"""torch.ops.neighbors.getNeighborPairs on HOST tensors (the CPU dispatch key the reference also registers,
src/pytorch/neighbors/getNeighborPairsCPU.cpp:102-108) against the 96 cases produced by the reference's own CPU kernel
(tests/golden/neighbors_ref.npz).  Runs without a GPU: libNNPOpsPyTorch.so loads and registers on any host."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def get_pairs():
    import NNPOps  # noqa: F401  (loads libNNPOpsPyTorch.so)
    from NNPOps.neighbors import getNeighborPairs
    return getNeighborPairs


def test_cpu_key_reproduces_the_reference_cpu_kernel(golden_dir, get_pairs):
    g = np.load(f"{golden_dir}/neighbors_ref.npz")
    for k in range(int(g["num_cases"])):
        pos, box = torch.tensor(g[f"c{k}_positions"]), g[f"c{k}_box"]
        nb, dl, ds, npairs = get_pairs(pos, float(g[f"c{k}_cutoff"]), int(g[f"c{k}_max_num_pairs"]),
                                       torch.tensor(box) if box.size else None)
        assert nb.dtype == torch.int32 and dl.dtype == pos.dtype and ds.dtype == pos.dtype
        assert np.array_equal(nb.numpy(), g[f"c{k}_neighbors"]), k                        # indices: bit-exact, same order
        tol = 1e-6 if pos.dtype == torch.float32 else 1e-13
        np.testing.assert_allclose(dl.numpy(), g[f"c{k}_deltas"], rtol=tol, atol=tol, equal_nan=True)
        np.testing.assert_allclose(ds.numpy(), g[f"c{k}_distances"], rtol=tol, atol=tol, equal_nan=True)
        assert int(npairs) == int(g[f"c{k}_num_pairs"][0])


def test_cpu_key_gradients(get_pairs):
    pos = torch.randn(24, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0), requires_grad=True)
    box = torch.tensor([[8.0, 0, 0], [1.0, 9.0, 0], [0.5, -1.0, 10.0]], dtype=torch.float64)

    def energy(p, mode):
        nb, dl, ds, _ = get_pairs(p, 2.5, mode, box)
        keep = nb[0] >= 0
        return (ds[keep] ** 2).sum() + (dl[keep] * torch.tensor([0.3, -0.2, 0.9], dtype=p.dtype)).sum()

    for mode in (-1, 400):
        assert torch.autograd.gradcheck(lambda p: energy(p, mode), (pos,), atol=1e-8)


def test_cpu_key_errors_match_the_reference_messages(get_pairs):
    pos = torch.zeros((4, 3))
    with pytest.raises(RuntimeError, match='Expected "cutoff" to be positive'):
        get_pairs(pos, -1.0)
    with pytest.raises(RuntimeError, match=r"box_vectors\[0\]\[0\] < 2\*cutoff"):
        get_pairs(pos, 3.0, -1, torch.eye(3) * 5.0)
    with pytest.raises(RuntimeError, match=r"box_vectors\[0\]\[1\] != 0"):
        get_pairs(pos, 1.0, -1, torch.tensor([[10.0, 1, 0], [0, 10, 0], [0, 0, 10]]))
    line = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]])
    with pytest.raises(RuntimeError, match="maximum number of pairs"):
        get_pairs(line, 3.0, 1, None, True)
    nb, _, _, npairs = get_pairs(line, 3.0, 1, None, False)      # not truncated, num_pairs = length after padding (:97-98)
    assert nb.shape == (2, 3) and int(npairs) == 3
