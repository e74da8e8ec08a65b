"""The batch-sharded path on real devices: 2 ranks over RCCL must assemble exactly the forces one GPU computes.
Skipped on boxes with fewer than two HIP devices (the driver's 8-GPU node runs it)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _forces(device, sizes, lo, hi):
    from nnpops_amd import workloads
    from nnpops_amd.capi import AniSymmetryFunctions
    mols = [workloads.conformer(sizes[m], seed=500 + m) for m in range(lo, hi)]
    pos = np.concatenate([m[0] for m in mols]).astype(np.float32)
    species = np.concatenate([m[1] for m in mols]).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(sizes[lo:hi])]).astype(np.int32)
    rf, af = workloads.ani2x_functions()
    dev = torch.device("cuda", device)
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, device=device)
    sym.set_molecules(offsets)
    radial, angular = sym.compute(torch.tensor(pos, device=dev))
    # upstream gradient = a fixed function of the global atom index, so that every sharding sees the same numbers
    first = int(np.sum(sizes[:lo]))
    idx = torch.arange(first, first + pos.shape[0], device=dev, dtype=torch.float32).unsqueeze(1)
    g_r = torch.sin(idx * 0.37 + torch.arange(radial.shape[1], device=dev) * 0.11)
    g_a = torch.cos(idx * 0.23 + torch.arange(angular.shape[1], device=dev) * 0.07)
    return sym.backprop(g_r.contiguous(), g_a.contiguous())


def _worker(rank, world, port, sizes, out_dir):
    import torch.distributed as dist
    from nnpops_amd.parallel import gather_rows, shard_molecules
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    blocks = shard_molecules(sizes, world)
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    rows = [int(offsets[hi] - offsets[lo]) for lo, hi in blocks]
    lo, hi = blocks[rank]
    full = gather_rows(_forces(rank, sizes, lo, hi), rows)
    torch.save(full.cpu(), os.path.join(out_dir, f"full_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices")
def test_two_ranks_gather_exactly_the_single_gpu_forces(tmp_path):
    import torch.multiprocessing as mp
    sizes = np.random.default_rng(3).integers(20, 70, size=48).tolist()
    mp.spawn(_worker, args=(2, _free_port(), sizes, str(tmp_path)), nprocs=2, join=True)
    single = _forces(0, sizes, 0, len(sizes)).cpu()
    for rank in range(2):
        assert torch.equal(torch.load(tmp_path / f"full_{rank}.pt"), single)


def test_two_rank_bench_rehearsal_on_one_device():
    """bench.py's multi-rank headline (own frame per rank, asynchronous all_gather of the forces with two buffer sets,
    bitwise self-check of the gathered block, one JSON line from rank 0) rehearsed with TWO ranks sharing ONE device over
    gloo ($NNPOPS_BENCH_BACKEND): the code path the driver's 8-GPU run takes, minus RCCL.  The numbers mean nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NNPOPS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--atoms", "2000", "--no-side", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 12 and line["scaling"] == "weak" and line["value"] > 0
    assert "roofline" in line and line["config"]["atoms"] == 2000
