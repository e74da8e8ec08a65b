"""bench.py's command line on CPU: the self-spawn of `--gpus N` and the CPU-worker mode (no HIP device needed)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_respawns_itself_under_torch_distributed_run(monkeypatch):
    """`python bench.py --gpus 4` outside torchrun must start 4 ranks (round 1's --gpus was parsed and ignored)."""
    import argparse
    import torch
    bench = _bench()
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    with pytest.raises(SystemExit) as exc:
        bench.spawn_ranks_if_needed(argparse.Namespace(gpus=4))
    assert exc.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_n_is_a_no_op_under_torchrun_and_at_one_gpu(monkeypatch):
    import argparse
    bench = _bench()
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not spawn")))
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.spawn_ranks_if_needed(argparse.Namespace(gpus=4))      # already a rank
    monkeypatch.delenv("WORLD_SIZE")
    bench.spawn_ranks_if_needed(argparse.Namespace(gpus=1))      # single GPU


def test_gpus_n_refuses_when_devices_are_missing(monkeypatch):
    import argparse
    import torch
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 HIP device"):
        bench.spawn_ranks_if_needed(argparse.Namespace(gpus=8))


def test_cpu_worker_mode_times_one_frame():
    """The 'all cores' CPU figure starts one `bench.py --cpu-worker` per core; the worker must not need torch or a GPU."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "300", "5"], cwd=ROOT)
    assert float(out.decode().strip()) > 0
