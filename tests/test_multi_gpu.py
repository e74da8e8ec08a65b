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


def test_one_rank_rccl_group_runs_the_gather(tmp_path):
    """What a one-device box can say about the RCCL path: a one-rank "nccl" group (RCCL initialises, its stream ordering against
    ours holds) carries nnpops_amd.parallel.gather_rows, and tools/rccl_world1_check.py the asynchronous double-buffered
    all_gather_into_tensor / all_reduce / barrier sequence of bench.py's N > 1 path."""
    import json
    import subprocess
    import sys
    import torch.multiprocessing as mp
    sizes = np.random.default_rng(3).integers(20, 70, size=24).tolist()
    mp.spawn(_worker, args=(1, _free_port(), sizes, str(tmp_path)), nprocs=1, join=True)
    assert torch.equal(torch.load(tmp_path / "full_0.pt"), _forces(0, sizes, 0, len(sizes)).cpu())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_world1_check.py")], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["gather_matches"] and line["rccl_version"]


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


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config4_full_batch_against_the_oracle_and_shard_invariance():
    """BASELINE config 4 in its own shape: the benchmark's 1 024 conformers (61 199 atoms) through ONE batched handle.
    (i) 64 sampled molecules against the per-molecule oracle (AEV element-wise, energy 1e-5, forces 1e-4 of the largest
    component); (ii) what the 8-GPU split relies on: blocks of shard_molecules(sizes, 8) evaluated by their OWN handle give
    BITWISE the forces the same molecules get inside the full batch (no result depends on what else is in the batch)."""
    from nnpops_amd import workloads
    from nnpops_amd.capi import AniSymmetryFunctions
    from nnpops_amd.parallel import shard_molecules
    from oracle import AniOracle
    bench = _bench_module()
    sizes = bench.conformer_sizes()
    assert len(sizes) == 1024
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    dev = torch.device("cuda:0")

    def upstream(first, rows, width_r, width_a):
        idx = torch.arange(first, first + rows, device=dev, dtype=torch.float32).unsqueeze(1)
        g_r = torch.sin(idx * 0.37 + torch.arange(width_r, device=dev) * 0.11)
        g_a = torch.cos(idx * 0.23 + torch.arange(width_a, device=dev) * 0.07)
        return g_r.contiguous(), g_a.contiguous()

    def evaluate(lo, hi):
        sh = bench.ConformerShard(sizes, lo, hi, 0)
        g_r, g_a = upstream(int(offsets[lo]), sh.n, sh.sym.radial_width, sh.sym.angular_width)
        radial, angular = sh.sym.compute(sh.tpos, None)
        grad = sh.sym.backprop(g_r, g_a)
        torch.cuda.synchronize()
        return sh, radial.clone(), angular.clone(), grad.clone(), g_r, g_a

    full, radial, angular, grad, g_r, g_a = evaluate(0, 1024)
    assert full.n == int(offsets[-1])
    assert bool(torch.isfinite(grad).all())
    r, a, g = radial.cpu().numpy(), angular.cpu().numpy(), grad.cpu().numpy()
    wr, wa = g_r.cpu().numpy(), g_a.cpu().numpy()
    rf, af = workloads.ani2x_functions()
    for m in np.random.default_rng(0).choice(1024, 64, replace=False):
        lo, hi = int(offsets[m]), int(offsets[m + 1])
        pos, species = full.mols[m]
        o = AniOracle(7, 5.1, 3.5, species, rf, af)
        r_ref, a_ref = o.forward(pos)
        np.testing.assert_allclose(r[lo:hi], r_ref, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(a[lo:hi], a_ref, rtol=2e-5, atol=2e-6)
        e_ref = float(r_ref.astype(np.float64).sum() + a_ref.astype(np.float64).sum())
        e = float(r[lo:hi].astype(np.float64).sum() + a[lo:hi].astype(np.float64).sum())
        assert abs(e - e_ref) <= 1e-5 * abs(e_ref)
        g_ref = o.backward(np.ascontiguousarray(wr[lo:hi]), np.ascontiguousarray(wa[lo:hi]))
        assert np.abs(g[lo:hi] - g_ref).max() <= 1e-4 * np.abs(g_ref).max(), m
    del full
    blocks = shard_molecules(sizes, 8)
    for rank in (0, 3, 7):
        lo, hi = blocks[rank]
        _, r_s, a_s, g_s, _, _ = evaluate(lo, hi)
        first, last = int(offsets[lo]), int(offsets[hi])
        assert torch.equal(g_s, grad[first:last]), rank
        assert torch.equal(r_s, radial[first:last]) and torch.equal(a_s, angular[first:last]), rank


def test_eight_rank_conformer_rehearsal_on_one_device():
    """`bench.py --workload conformers` as the driver's 8-GPU run launches it -- eight ranks under torch.distributed.run, each
    with its block of the 1 024 conformers and its own batched handle, the forces assembled by asynchronous padded
    all_gathers with two buffer sets -- rehearsed with the eight ranks sharing ONE device over gloo ($NNPOPS_BENCH_BACKEND).
    Checks inside the bench: every rank's own block comes back bitwise, every block finite.  The numbers mean nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NNPOPS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "conformers",
           "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["conformers"] == 1024 and 7000 < line["config"]["atoms_this_rank"] < 8400


def test_default_bench_line_carries_measured_counters():
    """`python bench.py` (shortened): ONE JSON line whose roofline carries the HBM traffic and the vector-issue floor of every
    headline kernel, both measured in the run by the rocprofv3 counter passes bench.py starts on itself, and a step that adds up."""
    import json
    import shutil
    import subprocess
    import sys
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "8", "--settle", "50", "--no-side",
                          "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["side_errors"] == {} and line["dtype"] == "f32"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["traffic_source"]
    for name, k in roof["per_kernel"].items():
        assert k["traffic"] is not None and k["traffic"] >= 0.9 * k["algorithmic_bytes_per_launch"], (name, k)
        assert 0.05 < k["valu"]["frac"] <= 1.0 and k["valu"]["valu_per_atom"] > 100, (name, k)
        # the event-bracket figure of the line against rocprofv3's own duration of the same kernel, taken in the same run
        assert abs(k["us"] - k["us_rocprofv3"]) <= 0.08 * k["us_rocprofv3"], (name, k)
    assert abs(roof["frac"] - roof["frac_rocprofv3"]) <= 0.08 * roof["frac_rocprofv3"]
    assert line["kernels_us_rocprofv3"]["cell_grid"] < 20
    four = sum(k["traffic"] for k in roof["per_kernel"].values())            # (the step also counts the two cell-grid kernels)
    assert four <= roof["step"]["traffic"] <= four + 16 * 2 ** 20


def test_torchani_side_line_carries_its_variants():
    """`python bench.py --workload torchani` (shortened): config 2's line names the AEV columns the networks multiply, and
    carries the step as an eager forward + backward, without the capacity check, replayed as a HIP graph (live and dense
    networks) and as ONE energy_and_forces call -- the roofline line about the kernel with the most bytes, the longest one
    named beside it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "torchani", "--steps", "30", "--warmup", "5",
                          "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["one_autograd_node"] and cfg["aev_columns"] == 1008 and cfg["aev_columns_multiplied"] == 128      # water: H and O of 7 species
    eager, nocheck, graph = line["ms_per_step"], line["ms_per_step_without_capacity_check"], line["ms_per_step_as_hip_graph"]
    dense, call = line["ms_per_step_as_hip_graph_with_dense_networks"], line["ms_per_energy_and_forces_call"]
    assert all(isinstance(v, float) and 0.02 < v < 2.0 for v in (eager, nocheck, graph, dense, call)), line
    assert graph < dense                                     # skipping the dead columns pays on the device
    # `achieved` / `frac` price the EXECUTED flops (live columns only); the split-fp16 path issues three products per fp32 product;
    # the reference's dense formulation (all 1008 columns) is a separate figure and the only one that may exceed the peak
    roof = line["roofline"]
    assert abs(roof["issued"]["tflops"] - 3 * roof["achieved"]) <= 0.02 * roof["issued"]["tflops"]
    assert 0 < roof["frac"] < 1 and 0 < roof["issued"]["frac"] < 1
    assert roof["vs_reference_formulation"]["tflops"] > roof["achieved"]
