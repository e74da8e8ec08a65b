#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X NNPOps hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--atoms 10000] [--no-cpu-baseline] [--no-side]

Workload (BASELINE.json metric: "AEV+forces (energy+grad evals/sec) per GPU, 10k-atom box"):
one *step* = one full ANI-2x symmetry-function evaluation of a 10 000-atom periodic box --
neighbour search + radial/angular forward (the 1008-wide AEV of every atom) + backward
(dE/dpositions for a fixed dense upstream gradient dE/dAEV) -- with positions, species, box and
the upstream gradient already resident in HBM.  Everything goes through the C ABI
(include/nnpops_hip.h) on the current HIP stream; there is no host synchronisation inside the
timed region.  Neighbour-buffer capacity is verified before and after the timed region.

Multi-GPU (`--gpus N`): when not already running under torch.distributed.run, bench.py re-executes
itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`,
one rank per GPU over RCCL.  The path shards over independent frames: every rank
evaluates its own 10k-atom frame (different seed) and the per-atom forces of all frames are assembled
with ONE all_gather per step (the only collective; "scaling": "weak");
value = total evaluations of all ranks / max-over-ranks time.  BASELINE config 4 (1024 conformers, strong
scaling over the same ranks) is run right after and reported under "side".

Rank 0 prints ONE JSON line.  At N = 1 the line also carries, under "side", short runs of the other
BASELINE configurations (each with its own roofline and CPU baseline); `--workload X` runs one of them
alone and prints its line instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP32_MATRIX_PEAK = 157.3   # TFLOP/s, v_mfma_f32_* with fp32 operands (same guide)
F16_DENSE_PEAK = 2500.0    # TFLOP/s, dense fp16/bf16 MFMA (same guide)

ROOFLINE_KERNELS = ("neighbors", "angular_forward", "angular_backward", "radial_backward")
# kernel-name fragments of the rocprofv3 trace -> the keys above (the two cell-grid kernels are reported with the step)
PMC_KERNEL_OF = {"ani_neighbors_cells": "neighbors", "ani_angular_forward_mfma": "angular_forward",
                 "ani_angular_backward_pair": "angular_backward", "ani_radial_backward_lanes": "radial_backward",
                 "bin_atoms": "cell_grid", "order_binned": "cell_grid"}
# seconds of SIMD issue time per instruction, measured on the MI355X by tools/ubench/valu_issue.hip with >= 2 waves per SIMD
# (profiles/r02_valu_issue_ubench.txt): plain VALU 1.1 ns, v_mfma_f32_4x4x1 3.5 ns; transcendentals are NOT priced extra -- their
# pipe overlaps the plain VALU of the other waves (round 3: replacing five of them by fourteen multiplies made the kernels slower)
VALU_ISSUE_NS, MFMA4X4_ISSUE_NS, SIMDS = 1.1, 3.5, 1024
ROCPROF_NAME = {"neighbors": "ani_neighbors_cells (neighbour rows + radial AEV)", "angular_forward": "ani_angular_forward_mfma",
                "angular_backward": "ani_angular_backward_pair", "radial_backward": "ani_radial_backward_lanes (+ force gather)"}


# =============================================================================================
# CPU legs (the reference's own CPU sources compiled in place under oracle/_ref, else the C restatement)
# =============================================================================================
def _cpu_classes():
    import oracle
    kind = "reference" if oracle.have_ref() else "port"
    return kind, (oracle.RefAni if kind == "reference" else oracle.AniOracle)


def _ani_eval_seconds(cls, pos, species, box, rf, af, repeats=1):
    """Seconds per forward+backward evaluation of one frame on one core."""
    import numpy as np
    from nnpops_amd import workloads
    obj = cls(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=box is not None)
    rng = np.random.default_rng(123)
    wr = wa = None
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        r, a = obj.forward(pos, box)
        if wr is None:
            wr = rng.standard_normal(r.shape).astype(np.float32)
            wa = rng.standard_normal(a.shape).astype(np.float32)
        obj.backward(wr, wa)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def usable_cores():
    """Cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota when there is one
    (os.cpu_count() reports the machine, not the container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_worker_main(argv):
    """`bench.py --cpu-worker N_ATOMS SEED`: one forward+backward of the reference CPU path on this core; prints the
    seconds.  Started nproc times in parallel by cpu_baseline() for the 'all cores' figure (no torch import here)."""
    from nnpops_amd import workloads
    n, seed = int(argv[0]), int(argv[1])
    _, cls = _cpu_classes()
    pos, species, box = workloads.random_box(n, density=0.1, seed=seed, n_species=7)
    rf, af = workloads.ani2x_functions()
    print(_ani_eval_seconds(cls, pos, species, box, rf, af), flush=True)


def cpu_baseline(pos, species, box, rf, af, budget_s=12.0, all_cores=True):
    """The reference CPU path (single-threaded by construction: no OpenMP in the reference) on ONE host core, on a bounded
    sample: as many fwd+bwd evaluations of the SAME frame as fit the budget (at least one); plus the only way the
    reference can use more cores -- one independent frame per core, all cores at once."""
    kind, cls = _cpu_classes()
    n = pos.shape[0]
    evals, t_total = 0, 0.0
    while True:
        dt = _ani_eval_seconds(cls, pos, species, box, rf, af)
        evals += 1
        t_total += dt
        if t_total + dt > budget_s:
            break
    out = {"value": evals / t_total, "unit": "evals/s", "cores": 1, "kind": kind,
           "sample": f"{evals} fwd+bwd evaluation(s) of the same {n}-atom ANI-2x periodic frame, single thread "
                     f"({t_total:.1f} s; {usable_cores()} usable cores of {os.cpu_count()} on the host, the reference CPU path is serial)"}
    if all_cores:
        # One process per core, each evaluating its own frame OF THE SAME SIZE, all started together: frames per second with
        # every core busy, measured, not scaled (frames above 20 000 atoms would take minutes per core: those fall back to
        # 4 000-atom frames and the figure is labelled extrapolated).
        ncpu = usable_cores()
        m = n if n <= 20000 else 4000
        from nnpops_amd import workloads
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(m), str(1000 + k)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT) for k in range(ncpu)]
        done, busy = 0, 0.0
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
                if p.returncode == 0 and o.strip():
                    done += 1
                    busy = max(busy, float(o.decode().strip()))
            except subprocess.TimeoutExpired:
                p.kill()
        wall = time.perf_counter() - t0
        if done:
            rate = done / busy                                # frames per second with every core busy
            if m == n:
                out["all_cores"] = {"value": rate, "unit": "evals/s", "cores": ncpu, "parallel_speedup": round(rate / out["value"], 1),
                                    "extrapolated": False,
                                    "sample": f"{done} independent {m}-atom frames (the benchmark's size), one process per core, started "
                                              f"together: slowest frame {busy:.2f} s ({wall:.1f} s wall with process start-up)"}
            else:
                p1, s1, b1 = workloads.random_box(m, density=0.1, seed=999, n_species=7)
                t_single = _ani_eval_seconds(cls, p1, s1, b1, rf, af)
                speedup = rate * t_single
                out["all_cores"] = {"value": out["value"] * speedup, "unit": "evals/s", "cores": ncpu, "parallel_speedup": round(speedup, 1),
                                    "extrapolated": True,
                                    "sample": f"{done} independent {m}-atom frames, one process per core: slowest {busy:.2f} s against "
                                              f"{t_single:.2f} s alone; that speed-up applied to the single-core {n}-atom figure"}
    return out


# =============================================================================================
# hardware counters of the headline kernels, measured IN THIS RUN: bench.py profiles three steps of itself under rocprofv3
# (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes) and reads the rocpd databases
# =============================================================================================
def _pmc_pass(counters, atoms, workdir, tag):
    """One rocprofv3 pass over `bench.py --steps 3` -> {roofline key: {counter: chip total per dispatch}} (None when rocprofv3
    is missing or the pass fails)."""
    import glob
    import shutil
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = os.path.join(workdir, tag)
    cmd = [exe, "--kernel-trace", "--pmc", *counters, "-d", out, "-o", tag, "--output-format", "rocpd", "--", sys.executable,
           os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--settle", "0", "--atoms", str(atoms), "--no-side",
           "--no-cpu-baseline", "--no-pmc"]
    env = dict(os.environ, TMPDIR=workdir)
    try:
        subprocess.run(cmd, cwd=workdir, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60, check=True)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        con = sqlite3.connect(dbs[0])
        rows = con.execute("select k.name, p.counter_name, sum(p.counter_value), count(distinct k.dispatch_id) "
                           "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except Exception:
        return None
    res = {}
    for name, counter, total, ndisp in rows:
        for frag, key in PMC_KERNEL_OF.items():
            if frag in name:
                slot = res.setdefault(key, {})
                slot[counter] = slot.get(counter, 0.0) + total / max(ndisp, 1)
    return res


def _trace_pass(atoms, workdir):
    """rocprofv3 --kernel-trace (no counters) over 300 steps of this very workload -> {roofline key: average kernel duration in us}
    (None when rocprofv3 is missing or the pass fails): the figure the committed profiles/ summaries hold, taken in the same run."""
    import glob
    import shutil
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = os.path.join(workdir, "kt")
    cmd = [exe, "--kernel-trace", "-d", out, "-o", "kt", "--output-format", "rocpd", "--", sys.executable, os.path.abspath(__file__),
           "--steps", "300", "--warmup", "0", "--settle", "50", "--atoms", str(atoms), "--no-side", "--no-cpu-baseline", "--no-pmc"]
    try:
        subprocess.run(cmd, cwd=workdir, env=dict(os.environ, TMPDIR=workdir), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=90, check=True)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        rows = sqlite3.connect(dbs[0]).execute("select name, count(*), avg(duration) from kernels group by name").fetchall()
    except Exception:
        return None
    res = {}
    for name, calls, avg_ns in rows:
        for frag, key in PMC_KERNEL_OF.items():
            if frag in name and calls >= 100:
                res[key] = res.get(key, 0.0) + avg_ns / 1e3          # (the two grid kernels add up)
    return res


def measure_counters(atoms):
    """-> ({kernel: HBM bytes per launch}, {kernel: {valu, mfma, waves}}, {kernel: rocprofv3's average duration in us}) from four
    rocprofv3 passes of this very workload, or empty dicts when rocprofv3 is not available.  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half
    the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is."""
    import shutil
    import tempfile
    workdir = tempfile.mkdtemp(prefix="nnpops_pmc_")
    try:
        fetch = _pmc_pass(["FETCH_SIZE"], atoms, workdir, "fetch")
        write = _pmc_pass(["WRITE_SIZE"], atoms, workdir, "write") if fetch else None
        sq = _pmc_pass(["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA"], atoms, workdir, "sq") if fetch else None
        traced = _trace_pass(atoms, workdir) if fetch else None
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    traffic, insts = {}, {}
    if fetch and write:
        for k in fetch:
            if k in write and "FETCH_SIZE" in fetch[k] and "WRITE_SIZE" in write[k]:
                traffic[k] = int((2.0 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024)
    if sq:
        for k, c in sq.items():
            insts[k] = {"valu": c.get("SQ_INSTS_VALU", 0.0), "mfma": c.get("SQ_INSTS_MFMA", 0.0), "salu": c.get("SQ_INSTS_SALU", 0.0),
                        "lds": c.get("SQ_INSTS_LDS", 0.0), "waves": c.get("SQ_WAVES", 0.0)}
    return traffic, insts, (traced or {})


def side_traffic(args, R, workload, scope, extra=()):
    """HBM bytes of the kernels a side line's roofline is about, measured in this run like the headline's: two rocprofv3 passes
    (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE, nothing else) over three steps of `bench.py --workload <workload>`.
    scope = [(kernel-name fragment, launches per step)] (launches None: as many as the profile shows per launch of the FIRST
    fragment's kernel, which runs once per step); -> (bytes per step over the scope, {fragment: bytes per launch}) or
    (None, None) when rocprofv3 is missing, a pass fails, --no-pmc was given or this is a multi-rank run.
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950: FETCH_SIZE reports half the bytes of wide reads,
    MI355X_MICROARCH.md)."""
    if getattr(args, "no_pmc", False) or R.world > 1:
        return None, None
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    workdir = tempfile.mkdtemp(prefix="nnpops_side_pmc_")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(workdir, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", counter, "--output-format", "rocpd", "--", sys.executable,
                   os.path.abspath(__file__), "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc",
                   "--no-shard8", *extra]
            subprocess.run(cmd, cwd=workdir, env=dict(os.environ, TMPDIR=workdir), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=240, check=True)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if not dbs:
                return None, None
            rows = sqlite3.connect(dbs[0]).execute(
                "select k.name, sum(p.counter_value), count(distinct k.dispatch_id) from pmc_events p join kernels k "
                "on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall()
            for name, total, ndisp in rows:
                for frag, _ in scope:
                    if frag in name:
                        slot = per.setdefault(frag, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                        slot[counter][0] += total
                        slot[counter][1] += ndisp
    except Exception as exc:                                  # (the line then carries "traffic": null; say why on stderr)
        print(f"bench.py: counter pass of side workload '{workload}' failed: {type(exc).__name__}: {str(exc)[:300]}", file=sys.stderr, flush=True)
        return None, None
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    by_kernel, step_bytes = {}, 0.0
    for frag, launches in scope:
        if frag not in per or per[frag]["FETCH_SIZE"][1] == 0 or per[frag]["WRITE_SIZE"][1] == 0:
            print(f"bench.py: counter pass of side workload '{workload}': no dispatch of '{frag}' in the profile", file=sys.stderr, flush=True)
            return None, None
        f, w = per[frag]["FETCH_SIZE"], per[frag]["WRITE_SIZE"]
        b = (2.0 * f[0] / f[1] + w[0] / w[1]) * 1024.0
        by_kernel[frag] = int(b)
        if launches is None:
            launches = f[1] / max(per[scope[0][0]]["FETCH_SIZE"][1], 1)
        step_bytes += launches * b
    return int(step_bytes), by_kernel


SIDE_TRAFFIC_SOURCE = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this run over three steps of the same workload: "
                       "(2 * FETCH_SIZE + WRITE_SIZE) KiB per launch, summed over the kernels of the scope")


def launch_floor_us(dev, launches=3, reps=2000):
    """What `launches` back-to-back launches of a kernel that does nothing cost on this device and runtime, eagerly and replayed as
    one HIP graph (us per group of launches): the floor under a latency-bound step."""
    import torch
    x = torch.zeros(64, device=dev)

    def group():
        for _ in range(launches):
            x.add_(0.0)
    for _ in range(50):
        group()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        group()
    torch.cuda.synchronize()
    eager = 1e6 * (time.perf_counter() - t0) / reps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        group()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        group()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return round(eager, 2), round(1e6 * (time.perf_counter() - t0) / reps, 2)


# =============================================================================================
# process group / self-spawn
# =============================================================================================
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class Ranks:
    """rank / device / process group of this process (world 1: no process group)."""

    def __init__(self):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU fallback in the product path)")
        # ($NNPOPS_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with ONE device -- the ranks share it,
        #  the collectives go through the host; numbers from such a run mean nothing)
        backend = os.environ.get("NNPOPS_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            self.local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)   # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        import torch
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        if self.dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def describe(self):
        """What the process group really is (N > 1): ranks as torch.distributed counts them, backend, RCCL version -- so that a reader
        of the line can confirm that N ranks over RCCL produced it."""
        if not self.dist:
            return None
        import torch
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                     # noqa: BLE001
            ver = None
        return {"world_size": int(self.dist.get_world_size()), "backend": str(self.dist.get_backend()),
                "rccl_version": ver, "devices_visible": int(torch.cuda.device_count())}

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


# =============================================================================================
# headline: ANI-2x AEV forward+backward, 10 000-atom periodic box
# =============================================================================================
def run_aev(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.capi import AniSymmetryFunctions, lib, OK
    rank, world, dev, dist = R.rank, R.world, R.dev, R.dist
    n = args.atoms
    pos, species, box = workloads.random_box(n, density=0.1, seed=100 + rank, n_species=7)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=True,
                               device=R.local_rank)
    if args.neighbor_algorithm:
        sym.set_neighbor_algorithm(args.neighbor_algorithm)
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev)
    radial = torch.empty((n, sym.radial_width), device=dev)
    angular = torch.empty((n, sym.angular_width), device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    g_rad = torch.randn(radial.shape, device=dev, generator=gen)
    g_ang = torch.randn(angular.shape, device=dev, generator=gen)
    # Two sets of force buffers: the all_gather of step s runs on RCCL's stream while the kernels of step s + 1 run on
    # ours, and a buffer is written again only after the gather that read it (two steps back) has been waited for -- a
    # stream-level wait, never the host.  Synchronously the ~20 us latency of a small all_gather over xGMI would be added to
    # every 90 us step.  With one rank nothing of this exists and `grad` is one buffer.
    grads = [torch.empty((n, 3), device=dev) for _ in range(2 if dist else 1)]
    gathered = [torch.empty((world * n, 3), device=dev) for _ in range(2)] if dist else None
    pending = [None, None]
    counter = [0]

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def step():
        b = counter[0] & 1 if dist else 0
        counter[0] += 1
        if dist and pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        sym.compute(tpos, tbox, radial, angular, check=False)
        sym.backprop(g_rad, g_ang, grads[b])
        if dist:                                              # the path's only exchange: per-atom forces of every frame
            pending[b] = dist.all_gather_into_tensor(gathered[b], grads[b], async_op=True)

    sym.compute(tpos, tbox, radial, angular, check=True)     # calibrates neighbour capacity (blocks)
    # About a quarter of a second of the same work before anything is counted: the W warm-up steps below are ~5 ms, too short for
    # the clocks of a device that has just been handed to this process to settle (one run in ten of this benchmark on a
    # fresh box came out 15 % low without it).  Garbage collection pauses are kept out of the loop for the same reason.
    import gc
    gc.collect()
    gc.disable()
    settle = args.settle if args.settle >= 0 else (2500 if n <= 20000 else 250)
    for _ in range(settle):                                  # (a COUNT, the same on every rank: every step holds a collective)
        step()
    drain()
    torch.cuda.synchronize()
    # Warm-up, with events around EVERY kernel: the per-kernel breakdown (diagnostic) and the choice of the
    # dominant kernel.  An event costs ~4.5 us of stream time (profiles/r02f_timeline.txt: the two gaps of a step sit
    # exactly around the bracketed kernel), so inside the timed region only the dominant kernel -- the one the roofline
    # line is about -- is bracketed, and only on every 8th step.
    # (the two bracketed passes run at least 300 steps each -- 30 ms: the bracket overhead below is a difference of three means and
    #  50 steps leave it +-1 us of noise; they are untimed like the W warm-up steps they contain)
    cal_steps = max(args.warmup, 300) if args.warmup else 0
    sym.enable_timing(True)
    # ... single brackets (one per kernel) and merged ones -- ONE bracket around neighbour build + angular forward and ONE around the two
    # backward kernels -- in alternating blocks of 100 steps: the sum of two single brackets minus the merged one is what a bracket adds
    # to the stream, measured in place (same launches, same cache state; alternating, so that a clock that drifts during these 60 ms
    # drifts under both).  It is neither the same on every box nor what an EMPTY bracket reports (3.5 us): 0.2 ... 1.7 us were seen.
    # (Two earlier round-4 attempts -- one fitted amount for all brackets; each kernel launched twice inside its bracket -- agreed with
    # rocprofv3 within 2 % on one box and missed by 7-11 % on the next.)
    def add(total, part):
        for k, (ms, c) in part.items():
            t = total.get(k, (0.0, 0))
            total[k] = (t[0] + ms, t[1] + c)
    breakdown, merged = {}, {}
    calibrate = bool(args.warmup) and not dist
    blocks = 6 if calibrate else 1
    per_block = -(-cal_steps // (blocks // 2 if calibrate else 1))
    for b in range(blocks if args.warmup else 0):
        sym.set_timing_merge(bool(b & 1))
        for _ in range(per_block):
            step()
        add(merged if b & 1 else breakdown, sym.get_timing())
    sym.set_timing_merge(False)
    sym.enable_timing(False)
    if not args.warmup:
        step()
    kern_all = {k: (1e-3 * ms / max(c, 1)) for k, (ms, c) in breakdown.items()}
    # ALGORITHMIC bytes per launch = SURVEY.md s8(d) bytes only (what an ideal implementation must move): inputs read
    # once, outputs written once, nothing of this implementation's intermediate arrays.
    na_w, nr_w = sym.angular_width, sym.radial_width
    alg = {"angular_forward": n * 16 + n * na_w * 4,          # positions+species in, the 896-float row out
           "angular_backward": n * na_w * 4 + n * 12,         # the upstream row in, forces out
           "neighbors": n * 16 + n * nr_w * 4,                # positions+species in, the radial AEV out (it is fused here)
           "radial_backward": n * nr_w * 4 + n * 12}          # the radial gradient row in, forces out
    # The HBM roofline line is about the kernel that has the most bytes to move (the angular forward pass: 44 % of the step's
    # algorithmic bytes; the kernel the round-1 and round-2 lines were about).  The LONGEST kernel of the step is named beside it
    # (`longest_kernel`): since round 3 that is the neighbour build, which moves an eighth of those bytes.
    dominant = max(ROOFLINE_KERNELS, key=lambda k: alg[k])
    longest = max(ROOFLINE_KERNELS, key=lambda k: kern_all.get(k, 0.0))
    R.barrier()
    event_overhead = sym.timing_overhead()                   # seconds reported for an EMPTY event bracket on this stream
    sym.enable_timing(True, only=[dominant], every=8)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                   # every gather of the timed steps has completed on our stream
    R.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    timing = sym.get_timing()
    sym.enable_timing(False)
    max_row, max_ang = sym.neighbor_stats()
    assert lib().nnpops_ani_check(sym._h, None, None) == OK, "neighbour buffers overflowed inside the timed region"
    last = (counter[0] - 1) & 1 if dist else 0
    grad = grads[last]
    assert bool(torch.isfinite(grad).all())
    if dist:
        assert bool(torch.equal(gathered[last][rank * n:(rank + 1) * n], grad))
    elapsed = R.max_over_ranks(elapsed)
    if rank != 0:
        return None

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    # Kernel durations from event brackets.  A bracket reports the kernel PLUS what the two events cost on the stream; that cost is
    # measured in place (warm-up above): o_fwd = bracket(build) + bracket(forward) - bracket(build + forward), o_bwd likewise for the two
    # backward kernels, and taken off the single brackets -- the dominant kernel's bracket of the TIMED region included.  The two
    # cell-grid kernels (one bracket around both) get what is left of the step: the kernels run back to back, the rocprofv3 timeline
    # of this loop has no idle time.  Without the merged brackets (N > 1 ranks, --warmup 0): one fitted amount for all brackets.
    raw = {k: v for k, v in kern_all.items() if v > 0}        # s per bracket, warm-up pass (every kernel bracketed)
    ms_dom, c_dom = timing[dominant]
    raw_dom_timed = 1e-3 * ms_dom / max(c_dom, 1)             # ... the dominant one again, from the timed region (every 8th step)
    step_s = elapsed / args.steps
    mrg = {k: (1e-3 * ms / max(c, 1)) for k, (ms, c) in merged.items() if c > 0 and ms > 0}
    calibrated = all(k in raw for k in ROOFLINE_KERNELS) and "neighbors" in mrg and "angular_backward" in mrg
    overhead = {}
    if calibrated:
        o_fwd = min(max(raw["neighbors"] + raw["angular_forward"] - mrg["neighbors"], 0.0), event_overhead)
        o_bwd = min(max(raw["angular_backward"] + raw["radial_backward"] - mrg["angular_backward"], 0.0), event_overhead)
        # (what a bracket adds is a property of the event mechanism, not of the pair of kernels it was measured on: the two estimates
        #  are pooled -- the larger one, the forward pair's comes out low whenever its merged bracket caught a slow step -- and used for
        #  all four brackets)
        o_all = max(o_fwd, o_bwd)
        overhead = {"neighbors": o_all, "angular_forward": o_all, "angular_backward": o_all, "radial_backward": o_all}
        kern = {k: 0.0 for k in kern_all}
        for k in ROOFLINE_KERNELS:
            kern[k] = max(raw[k] - overhead[k], 1e-9)
        kern[dominant] = max(raw_dom_timed - overhead[dominant], 1e-9)
        kern["cell_grid"] = max(step_s - sum(kern[k] for k in ROOFLINE_KERNELS), 0.0)
        correction = overhead[dominant]
    else:
        correction = min(max((sum(raw.values()) - step_s) / max(len(raw), 1), 0.0), event_overhead) if raw else 0.0
        kern = {k: (max(v - correction, 1e-9) if v > 0 else 0.0) for k, v in kern_all.items()}
        kern[dominant] = max(raw_dom_timed - correction, 1e-9)
    step_bytes = n * (16 + 2 * (na_w + nr_w) * 4 + 12)        # N * (16 + 2 * 4032 + 12)
    # HBM traffic and instruction counts of these kernels, measured now (rocprofv3 on three steps of this same workload; the
    # counters come from their own passes, the TIMES above from the un-profiled run)
    traffic_all, insts, traced = ({}, {}, {}) if (args.no_pmc or world > 1) else measure_counters(n)
    traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run: (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch" if traffic_all else None

    def valu_floor(k):
        """SIMD issue time the kernel's vector instructions need at the measured issue prices, against its measured duration: a
        kernel at frac ~0.7 is bound by vector-instruction issue, not by HBM."""
        c = insts.get(k)
        if not c or kern.get(k, 0) <= 0:
            return None
        plain = c["valu"] - c["mfma"]
        floor_s = (plain * VALU_ISSUE_NS + c["mfma"] * MFMA4X4_ISSUE_NS) * 1e-9 / SIMDS
        return {"valu_per_atom": round(c["valu"] / n, 1), "mfma_per_atom": round(c["mfma"] / n, 1), "salu_per_atom": round(c["salu"] / n, 1),
                "lds_per_atom": round(c["lds"] / n, 1), "issue_floor_us": round(1e6 * floor_s, 2),
                "frac": round(floor_s / kern[k], 4)}

    def roof(k):
        ach = alg[k] / kern[k] / 1e9 if kern.get(k, 0) > 0 else 0.0
        res = {"kernel": ROCPROF_NAME[k], "us": round(1e6 * kern.get(k, 0.0), 2), "algorithmic_bytes_per_launch": alg[k],
               "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic_all.get(k), "valu": valu_floor(k)}
        if traced.get(k):                                     # the same kernel as rocprofv3 --kernel-trace times it, in this run
            res["us_rocprofv3"] = round(traced[k], 2)
            res["frac_rocprofv3"] = round(alg[k] / (traced[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        return res

    dom = roof(dominant)
    out = {
        "metric": "AEV+forces evaluations/sec (ANI-2x symmetry functions, energy+gradient), 10k-atom periodic box",
        "value": round(value, 3), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ANI-2x AEV forward+backward, {n}-atom periodic cubic box at 0.1 atoms/A^3, "
                               "7 species uniform, Rcr 5.1 / Rca 3.5, 16 radial + 32 angular functions (AEV width 1008); "
                               "one independent frame per GPU" + (", forces of all frames all_gathered every step (RCCL)" if dist else ""),
                   "atoms": n, "frames_per_gpu": 1, "max_neighbors_rcr": max_row, "max_neighbors_rca": max_ang},
        "process_group": R.describe(),
        "kernels_us": {k: round(1e6 * v, 2) for k, v in kern.items()},
        "kernels_us_sum": round(1e6 * sum(kern.values()), 2),
        "kernels_us_rocprofv3": ({k: round(v, 2) for k, v in traced.items()} or None),      # rocprofv3 --kernel-trace of 300 steps, run from here
        "event_pair_overhead_us": round(1e6 * event_overhead, 2),
        "bracket_correction_us": round(1e6 * correction, 2),
        "bracket_overhead_us": {k: round(1e6 * v, 2) for k, v in overhead.items()} or None,
        "kernels_us_note": ("single event brackets minus what a bracket adds to the stream, measured in place: bracket(build) + bracket(forward) "
                            "- bracket(build + forward), and the same for the two backward kernels (`bracket_overhead_us`); the dominant kernel "
                            "from its brackets in the timed region; cell_grid = what is left of the step (the kernels run back to back): "
                            "comparable with rocprofv3 --kernel-trace averages"
                            if calibrated else
                            "event brackets minus `bracket_correction_us`, the one amount that makes the brackets of a step add up to "
                            "ms_per_step (no double brackets in this run)"),
        "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac"], "us_rocprofv3": dom.get("us_rocprofv3"), "frac_rocprofv3": dom.get("frac_rocprofv3"),
                     "traffic": dom["traffic"], "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "valu": dom["valu"],
                     "longest_kernel": roof(longest),
                     "limiter": "not HBM: the per-atom kernels are bound by vector-instruction issue while the chip is full and by "
                                "latency in the last occupancy round (DESIGN.md s3/s7: 390-1160 VALU instructions per atom per "
                                "kernel, 1.2-2.8 rounds of resident waves at 10 000 atoms)",
                     "angular": {"forward": roof("angular_forward"), "backward": roof("angular_backward")},
                     "per_kernel": {k: roof(k) for k in ROOFLINE_KERNELS},
                     "step": {"algorithmic_bytes": step_bytes, "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                              "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "traffic": (sum(traffic_all.values()) if traffic_all else None)}},
    }
    if not args.no_cpu_baseline and world == 1:              # the CPU leg is timed on rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(pos, species, box, rf, af)
    return out


# =============================================================================================
# BASELINE config 1: 50-atom molecule in vacuum, latency per evaluation
# =============================================================================================
def run_latency(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.capi import AniSymmetryFunctions
    dev = R.dev
    pos, species = workloads.conformer(50, seed=0)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, device=R.local_rank)
    tpos = torch.tensor(pos, device=dev)
    radial, angular = sym.compute(tpos, None, check=True)
    g_rad, g_ang = torch.randn_like(radial), torch.randn_like(angular)
    grad = torch.empty((50, 3), device=dev)

    def step():
        sym.compute(tpos, None, radial, angular, check=False)
        sym.backprop(g_rad, g_ang, grad)

    for _ in range(50):
        step()
    torch.cuda.synchronize()
    k = 2000 if args.steps >= 20 else 20               # (the counter passes of this very workload run three steps: side_traffic)
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    eager_us = 1e6 * (time.perf_counter() - t0) / k
    # the same five launches replayed from one HIP graph (what an MD loop would do)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        graph.replay()
    torch.cuda.synchronize()
    graph_us = 1e6 * (time.perf_counter() - t0) / k
    # ... and the workload of the reference's own benchmark (src/pytorch/BenchmarkTorchANISymmetryFunctions.py:20-58: ONE ligand
    # evaluated in a loop): the 1hvj ligand (115 atoms) the reference's tests hold, geometry and the reference CPU core's AEV from
    # the committed fixture (tests/golden/molecules_ref.npz: numbers only)
    ligand = None
    fixture = os.path.join(ROOT, "tests", "golden", "molecules_ref.npz")
    if os.path.exists(fixture):
        mol = np.load(fixture)
        lpos, lspecies = mol["1hvj_positions"].astype(np.float32), mol["1hvj_species"].astype(np.int32)
        lsym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], lspecies, rf, af, device=R.local_rank)
        ltpos = torch.tensor(lpos, device=dev)
        lrad, lang = lsym.compute(ltpos, None, check=True)

        def fixture_weights(shape, kk):      # the upstream gradients the fixture's forces belong to (tests/golden/make_golden_molecules.py)
            ii, jj = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
            return (np.round(np.cos(0.37 * ii + 1.3 * jj + 0.5 + kk) * 64) / 64).astype(np.float32)
        lg_rad = torch.tensor(fixture_weights(tuple(lrad.shape), 0), device=dev)
        lg_ang = torch.tensor(fixture_weights(tuple(lang.shape), 100), device=dev)
        lgrad = torch.empty((len(lspecies), 3), device=dev)

        def lstep():
            lsym.compute(ltpos, None, lrad, lang, check=False)
            lsym.backprop(lg_rad, lg_ang, lgrad)

        for _ in range(50):
            lstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            lstep()
        torch.cuda.synchronize()
        ligand_us = 1e6 * (time.perf_counter() - t0) / k
        err_aev = float(max(np.abs(lrad.cpu().numpy() - mol["1hvj_radial"]).max(), np.abs(lang.cpu().numpy() - mol["1hvj_angular"]).max()))
        ref_grad = mol["1hvj_grad"]
        err_grad = float(np.abs(lgrad.cpu().numpy() - ref_grad).max() / np.abs(ref_grad).max())
        assert err_aev < 2e-5 and err_grad < 1e-4, (err_aev, err_grad)
        ligand = {"molecule": "1hvj ligand, 115 atoms (reference src/pytorch/molecules/1hvj_ligand.mol2, geometry from tests/golden/molecules_ref.npz)",
                  "eager_us": round(ligand_us, 2), "max_abs_aev_error_vs_reference_cpu": err_aev,
                  "max_force_error_over_largest_force_vs_reference_cpu": err_grad}
    floor_eager, floor_graph = launch_floor_us(dev, launches=3, reps=2000 if args.steps >= 20 else 20)
    step_traffic, traffic_by_kernel = side_traffic(args, R, "latency", [("ani_build_forward", 1), ("ani_angular_backward_pair", 1),
                                                                        ("ani_radial_backward_lanes", 1)])
    out = {"metric": "ANI-2x AEV forward+backward latency, 50-atom molecule in vacuum", "value": round(min(eager_us, graph_us), 2),
           "unit": "us/eval", "higher_is_better": False, "eager_us": round(eager_us, 2), "hip_graph_us": round(graph_us, 2),
           "algorithmic_bytes": 50 * (16 + 2 * 1008 * 4 + 12), "ligand_1hvj": ligand,
           # a latency-bound line: no bandwidth or matrix peak applies to 0.4 MB in three dependent launches -- what bounds it is the
           # launch floor of this device and runtime, measured here with three launches of a kernel that does nothing
           "roofline": {"bound": "latency", "kernel": "ani_build_forward + ani_angular_backward_pair + ani_radial_backward_lanes",
                        "launches": 3, "floor_us": floor_eager, "floor_us_as_hip_graph": floor_graph,
                        "achieved": round(min(eager_us, graph_us), 2), "peak": floor_eager, "unit": "us",
                        "frac": round(floor_eager / max(min(eager_us, graph_us), 1e-9), 5),
                        "traffic": step_traffic, "traffic_by_kernel": traffic_by_kernel,
                        "traffic_source": SIDE_TRAFFIC_SOURCE if step_traffic is not None else None,
                        "algorithmic_bytes_per_launch": 50 * (16 + 2 * 1008 * 4 + 12)},
           "hip_graph_note": "three EMPTY launches replayed as one HIP graph cost floor_us_as_hip_graph against floor_us eagerly (roofline "
                             "object, measured in this run): the replay's own fixed cost is what the graph figure of this line carries over the "
                             "eager one -- the host stays ahead of a device that needs ~25 us per evaluation, so eager launches cost the step nothing",
           "config": {"workload": "BASELINE config 1: 50-atom conformer (seed 0), non-periodic, all-pairs neighbour search, "
                                  "3 launches per evaluation (fused build + forward, two backward kernels); latency-bound (0.4 MB of algorithmic traffic)"}}
    if not args.no_cpu_baseline:
        kind, cls = _cpu_classes()
        dt = _ani_eval_seconds(cls, pos, species, None, rf, af, repeats=20)
        out["cpu_baseline"] = {"value": round(1e6 * dt, 1), "unit": "us/eval", "cores": 1, "kind": kind,
                               "sample": "best of 20 fwd+bwd evaluations of the same molecule, single thread"}
    return out


# =============================================================================================
# BASELINE config 5: getNeighborPairs + AEV at 100 000 atoms
# =============================================================================================
def run_neighbors(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.capi import AniSymmetryFunctions, neighbor_pairs_forward
    dev = R.dev
    n = args.atoms if args.atoms != 10000 else 100000
    cutoff, max_pairs = 5.2, int(32 * n)
    pos, species, box = workloads.random_box(n, density=0.1, seed=6, n_species=7)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=True, device=R.local_rank)
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    radial = torch.empty((n, sym.radial_width), device=dev)
    angular = torch.empty((n, sym.angular_width), device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    g_rad = torch.randn(radial.shape, device=dev, generator=gen)
    g_ang = torch.randn(angular.shape, device=dev, generator=gen)
    grad = torch.empty((n, 3), device=dev)
    sym.compute(tpos, tbox, radial, angular, check=True)
    nb, dl, ds, npairs = neighbor_pairs_forward(tpos, cutoff, max_pairs, tbox)
    found = int(npairs.item())
    assert 0 < found < max_pairs

    def step():
        neighbor_pairs_forward(tpos, cutoff, max_pairs, tbox)
        sym.compute(tpos, tbox, radial, angular, check=False)
        sym.backprop(g_rad, g_ang, grad)

    steps, warm = min(args.steps, 50), min(args.warmup, 10)
    elapsed = _time_steps(step, steps, warm) * steps         # (SIDE_PROTOCOL)
    sym.enable_timing(True)
    step()                                             # one extra step with the handle's own kernel brackets
    torch.cuda.synchronize()
    kt = {k: 1e3 * ms / max(c, 1) for k, (ms, c) in sym.get_timing().items()}        # us per launch (event brackets included)
    sym.enable_timing(False)

    # The phase split: every phase in a loop of its own (20 calls between two events).  A single bracketed step, as rounds 1-4 took it,
    # charges getNeighborPairs the host's time to issue its ten launches -- the device idles between 5 us kernels while the host is
    # still queueing -- which a step inside the loop above does not pay (the host runs ahead there).
    def phase_ms(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_nb = phase_ms(lambda: neighbor_pairs_forward(tpos, cutoff, max_pairs, tbox))
    t_fwd = phase_ms(lambda: sym.compute(tpos, tbox, radial, angular, check=False))
    t_bwd = phase_ms(lambda: sym.backprop(g_rad, g_ang, grad))
    # the op's own backward (getNeighborPairsCUDA.cu:80-101): dE/dpositions from gradients of deltas and distances, no float atomics
    # (round 6: an owner-computes gather over the list's transposed index, which the torch op builds in forward() when the positions
    #  require a gradient; the fixed-point integer atomics of rounds 4-5 -- what a list of unknown origin still takes -- beside it)
    from nnpops_amd.capi import neighbor_pairs_backward, neighbor_pairs_backward_indexed, neighbor_pairs_build_index
    g_dl, g_ds = torch.randn(dl.shape, device=dev, generator=gen), torch.randn(ds.shape, device=dev, generator=gen)
    pair_index = neighbor_pairs_build_index(n, nb)
    t_nb_bwd = phase_ms(lambda: neighbor_pairs_backward_indexed(n, nb, dl, ds, g_dl, g_ds, pair_index))
    t_nb_index = phase_ms(lambda: neighbor_pairs_build_index(n, nb))
    t_nb_bwd_fixed = phase_ms(lambda: neighbor_pairs_backward(n, nb, dl, ds, g_dl, g_ds))
    nb_bwd_bytes = found * 40 + n * 12                    # (2 ints + 3 + 1 floats of the list, 3 + 1 floats of gradient) per pair in, 12 bytes per atom out
    # the list's immediate consumer in the reference: direct-space PME (src/pytorch/pme/pme.py:163-165)
    from nnpops_amd.capi import pme_direct
    charges = torch.randn(n, device=dev, generator=gen) * 0.3
    no_excl = torch.full((n, 1), -1, dtype=torch.int32, device=dev)
    for _ in range(3):
        pme_direct(tpos, charges, nb, dl, ds, no_excl, 0.6, 332.063713)
    pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pe[0].record()
    for _ in range(20):
        pme_direct(tpos, charges, nb, dl, ds, no_excl, 0.6, 332.063713)
    pe[1].record()
    torch.cuda.synchronize()
    t_pme = pe[0].elapsed_time(pe[1]) / 20
    # (round 6) ... and over the list's transposed index, as the torch op runs it on the list of a differentiable getNeighborPairs call
    t_pme_indexed = phase_ms(lambda: pme_direct(tpos, charges, nb, dl, ds, no_excl, 0.6, 332.063713, index=pair_index))
    pme_bytes = max_pairs * 12 + found * 16 + n * 16 + n * 16      # list read (neighbor ids of every slot, delta + r of live ones), q + derivatives
    nb_bytes = n * 12 + found * 24                      # positions in, (2 ints + 3 floats + 1 float) per pair out
    ang_bytes = n * 16 + n * sym.angular_width * 4
    aev_bytes = n * (16 + 2 * (sym.radial_width + sym.angular_width) * 4 + 12)
    # HBM bytes of the getNeighborPairs launches (this line's roofline), counters of this run
    # (the grid of a system of this size is the five-launch build with a tiled scan -- the AEV handle builds its own with the same
    #  kernels: one launch of each is charged to getNeighborPairs)
    nb_traffic, nb_by_kernel = side_traffic(args, R, "neighbors", [("pairs_cells_stage", 1), ("pairs_cells_emit", 1), ("scan_rows", 1),
                                                                  ("grid_setup", 1), ("assign_cells", 1), ("scan_cells", 1),
                                                                  ("add_tile_offsets", 1), ("fill_cells", 1), ("order_cells", 1)])
    out = {
        "metric": "getNeighborPairs + ANI-2x AEV forward+backward evaluations/sec, 100k-atom periodic box, cutoff 5.2 A",
        "value": round(steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": steps, "warmup": warm, "timing_protocol": SIDE_PROTOCOL,
        "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"getNeighborPairs(cutoff {cutoff}, max_num_pairs {max_pairs}) + ANI-2x AEV, {n} atoms periodic, "
                               "0.1 atoms/A^3, 7 species", "atoms": n, "pairs_found": found},
        "phases_ms": {"neighbor_pairs": round(t_nb, 4), "neighbor_pairs_backward": round(t_nb_bwd, 4),
                      "neighbor_pairs_transposed_index": round(t_nb_index, 4), "neighbor_pairs_backward_fixed_point": round(t_nb_bwd_fixed, 4),
                      "aev_forward": round(t_fwd, 4),
                      "aev_backward": round(t_bwd, 4), "pme_direct": round(t_pme, 4), "pme_direct_indexed": round(t_pme_indexed, 4)},
        "kernels_us": {k: round(v, 1) for k, v in kt.items()},
        "roofline": {"bound": "hbm", "kernel": "getNeighborPairs (stage, scan, emit + cell grid)", "achieved": round(nb_bytes / (t_nb * 1e-3) / 1e9, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb_bytes / (t_nb * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "traffic": nb_traffic, "traffic_by_kernel": nb_by_kernel,
                     "traffic_source": SIDE_TRAFFIC_SOURCE if nb_traffic is not None else None, "algorithmic_bytes_per_launch": nb_bytes,
                     "angular_forward": {"algorithmic_bytes": ang_bytes, "us": round(kt.get("angular_forward", 0.0), 1),
                                         "achieved": round(ang_bytes / max(kt.get("angular_forward", 0.0), 1e-3) / 1e3, 2),
                                         "frac": round(ang_bytes / max(kt.get("angular_forward", 0.0), 1e-3) / 1e3 / HBM_PEAK_GBS, 5)},
                     "neighbor_pairs_backward": {"algorithmic_bytes": nb_bwd_bytes, "achieved": round(nb_bwd_bytes / (t_nb_bwd * 1e-3) / 1e9, 2),
                                                 "frac": round(nb_bwd_bytes / (t_nb_bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                 "note": "pairs_backward_terms + pairs_backward_gather (no atomics); the index it walks is built once per "
                                                         "forward call that can be differentiated: phases_ms.neighbor_pairs_transposed_index"},
                     "pme_direct": {"algorithmic_bytes": pme_bytes, "achieved": round(pme_bytes / (t_pme * 1e-3) / 1e9, 2),
                                    "frac": round(pme_bytes / (t_pme * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "indexed": {"achieved": round(pme_bytes / (t_pme_indexed * 1e-3) / 1e9, 2), "frac": round(pme_bytes / (t_pme_indexed * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                "note": "nnpops_pme_direct_indexed: pme_direct_terms + pme_direct_gather_indexed over the list's transposed index (no atomics); what torch.ops.pme.pme_direct runs on the list of a getNeighborPairs call that can be differentiated"},
                                    "note": "energy + dE/dpositions + dE/dcharges on the pair list above (owner computes: pme_direct_pairs parks every contribution in the second atom's row, pme_direct_gather sums them in a fixed order; no float atomics)"},
                     "aev_step": {"algorithmic_bytes": aev_bytes,
                                  "achieved": round(aev_bytes / ((t_fwd + t_bwd) * 1e-3) / 1e9, 2)}},
    }
    if not args.no_cpu_baseline:
        # the reference cannot run getNeighborPairs at this size (int32 pair index); its AEV path is O(N^2): time it at
        # three sizes and extrapolate t = a N^2 + b N (least squares) to N
        kind, cls = _cpu_classes()
        sizes, times = [2000, 4000, 8000], []
        for m in sizes:
            p, s, b = workloads.random_box(m, density=0.1, seed=6, n_species=7)
            times.append(_ani_eval_seconds(cls, p, s, b, rf, af))
        A = np.array([[m * m, m] for m in sizes], dtype=np.float64)
        coef, *_ = np.linalg.lstsq(A, np.array(times), rcond=None)
        t_n = float(coef[0] * n * n + coef[1] * n)
        out["cpu_baseline"] = {"value": round(1.0 / t_n, 6), "unit": "evals/s (AEV fwd+bwd only)", "cores": 1, "kind": kind,
                               "sample": f"extrapolated: reference AEV fwd+bwd timed at N = {sizes} ({', '.join(f'{t:.2f}' for t in times)} s), "
                                         f"fit t = {coef[0]:.3e} N^2 + {coef[1]:.3e} N -> {t_n:.0f} s at N = {n}; the reference cannot "
                                         "run getNeighborPairs at this size at all"}
    return out


# =============================================================================================
# BASELINE config 2: OptimizedTorchANI, 2 001-atom periodic water box
# =============================================================================================
def run_torchani(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from NNPOps import OptimizedTorchANI
    from NNPOps.BatchedNN import TorchANIBatchedNN
    dev = R.dev
    model = workloads.torchani_like_model(n_models=8, seed=2)
    pos, species, box = workloads.water_box(667, seed=1)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=dev)
    opt = OptimizedTorchANI(model, numbers.cpu(), nn_layout=args.nn_layout, fused_step=not args.no_fused_step).to(dev)
    one_node = type(opt).__name__ == "FusedOptimizedTorchANI"
    # (pbc stays on the host: the wrapper reads it with .tolist(), reference SymmetryFunctions.py:113, which on a
    # device tensor is a synchronising copy and cannot be captured)
    cell, pbc = torch.tensor(box, device=dev), torch.tensor([True, True, True])
    tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
    n = len(species)

    def step():
        tpos.grad = None
        energy = opt((numbers, tpos), cell, pbc).energies
        energy.backward()               # (one molecule: energies is [1]; the reference's own benchmark calls it the same way, BenchmarkBatchedNN.py:75,92)
        return energy.detach()          # (not the autograd graph: one kept alive from an eager step breaks a later stream capture)

    steps, warm = min(args.steps, 200), min(args.warmup, 20)
    for _ in range(max(warm, 3)):
        step()
    torch.cuda.synchronize()
    def replay_as_graph():
        """the whole energy+forces step as one HIP graph (the AEV holder skips its capacity check while capturing; capacities were
        calibrated by the warm-up steps above): removes the host's ~10 launches and its interpreter / autograd time per step"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        tpos.grad = None
        with torch.cuda.graph(graph):
            g_energy = opt((numbers, tpos), cell, pbc).energies
            g_forces = torch.autograd.grad(g_energy.sum(), tpos)[0]
        torch.cuda.synchronize()
        eager_e = step().clone()
        eager_g = tpos.grad.detach().clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.allclose(g_energy, eager_e, rtol=1e-6, atol=1e-4) and torch.allclose(g_forces, eager_g, rtol=1e-4, atol=1e-5)
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, g_energy, g_forces

    graph_ms = dense_graph_ms = call_ms = None
    if args.graph:
        elapsed, energy, forces = replay_as_graph()
        tpos.grad = forces
    else:
        elapsed = _time_steps(step, steps, warm) * steps     # (SIDE_PROTOCOL)
        energy = step()
        torch.cuda.synchronize()
    assert bool(torch.isfinite(energy).all()) and bool(torch.isfinite(tpos.grad).all())
    # the same eager loop without the per-call capacity check of the AEV holder (one host round trip per step): what a production
    # loop at known density runs (set_check_interval, cf. check_errors of getNeighborPairs); one checked step follows
    no_check_ms = None
    if not args.graph:
        (opt if one_node else opt.aev_computer).set_check_interval(0)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        no_check_ms = 1e3 * (time.perf_counter() - t1) / steps
        (opt if one_node else opt.aev_computer).set_check_interval(1)
        step()                                               # (raises if a neighbour buffer had overflowed)
        torch.cuda.synchronize()
        try:                                                 # ... and the same step replayed as a HIP graph (device time, no host in the loop)
            graph_ms = 1e3 * replay_as_graph()[0] / steps
        except Exception as exc:                             # noqa: BLE001 -- a figure beside the line, not the line
            print(f"bench: graph replay of the torchani step failed: {exc!r}", file=sys.stderr)
        # ... and as ONE call outside autograd (FusedOptimizedTorchANI.energy_and_forces: what an MD driver that takes the forces as
        # a model output runs -- no .sum() / backward(), no autograd engine)
        if one_node:
            tp = tpos.detach()
            for _ in range(3):
                opt.energy_and_forces((numbers, tp), cell, pbc)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                e_call, f_call = opt.energy_and_forces((numbers, tp), cell, pbc)
            torch.cuda.synchronize()
            call_ms = 1e3 * (time.perf_counter() - t1) / steps
            assert bool(torch.isfinite(e_call).all()) and bool(torch.isfinite(f_call).all())
        # ... and what the step costs when the networks multiply ALL 1008 AEV columns, the identically-zero blocks of absent species
        # included, as the reference's dense BatchedLinear does (OptimizedTorchANI(live_columns=False))
        nets_live = opt.neural_networks[0]
        if one_node and graph_ms is not None and hasattr(nets_live, "x_blocks") and nets_live.x_blocks.numel():
            live_opt = opt
            try:
                opt = OptimizedTorchANI(model, numbers.cpu(), nn_layout=args.nn_layout, live_columns=False).to(dev)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                dense_graph_ms = 1e3 * replay_as_graph()[0] / steps
            except Exception as exc:                         # noqa: BLE001
                print(f"bench: dense-networks variant of the torchani step failed: {exc!r}", file=sys.stderr)
            finally:
                opt = live_opt
    # ... and with weight sets whose crude activation bound (BatchedNN.py: the products of the layers' L1 norms) is beyond what the
    # 1/16 scale of the fp16 planes holds.  The published ANI-2x weights cannot be checked offline (no torchani, no network).  Up to
    # round 4 such a set sent OptimizedTorchANI to the four-module composition with the networks on the library GEMMs (~1.0 ms, 9x);
    # the kernels now take the scale as an argument (act_scale_log2, 4..12) and the module picks it from the bound: first-layer
    # weights x 100 (bound 3.4e7) stay on the fused step, x 1e4 (bound 3.4e9, beyond 2^-12) still fall back.  Same shapes, same launches.
    unfused_ms = unfused_layout = rescaled_ms = rescaled_log2 = None
    if not args.graph and one_node:
        import copy
        live_opt = opt
        for factor in (1.0e2, 1.0e4):
            try:
                big = copy.deepcopy(model)
                for ens in big.neural_networks:
                    for net in ens.values():
                        net[0].weight.data *= factor
                opt = OptimizedTorchANI(big, numbers.cpu(), nn_layout=args.nn_layout).to(dev)
                nets_big = opt.neural_networks[0]
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t1) / steps
                if getattr(nets_big, "fused_ok", False):
                    assert factor == 1.0e2 and type(opt).__name__ == "FusedOptimizedTorchANI"
                    rescaled_ms, rescaled_log2 = ms, int(nets_big.act_scale_log2)
                else:
                    assert type(opt).__name__ != "FusedOptimizedTorchANI"
                    unfused_ms = ms
                    unfused_layout = type(opt).__name__ + " / " + type(nets_big).__name__ + " (fused_ok = False: grouped library GEMMs)"
            except Exception as exc:                             # noqa: BLE001 -- figures beside the line, not the line
                print(f"bench: the large-weights variant (x{factor:g}) of the torchani step failed: {exc!r}", file=sys.stderr)
            finally:
                opt = live_opt
    # NN flops: 2 * models * sum over atoms of the MACs of its network; backward to the
    # inputs costs the same again
    macs = {s: 1008 * a + a * b + b * c + c for s, (a, b, c) in enumerate(workloads.ANI2X_WIDTHS.values())}
    flops_fwd = 2.0 * 8 * sum(macs[int(s)] for s in species)
    nn_weight_bytes = sum(b.numel() * 4 for name, b in opt.neural_networks.named_buffers() if "layer" in name)
    tflops_dense = 2 * flops_fwd / elapsed * steps / 1e12    # the reference's formulation: every one of the 1008 columns multiplied
    # what is actually multiplied: inside the one-node step the networks run over the AEV column blocks this molecule's species
    # can fill (water: H and O of the 7 species -> 128 of the 1008 columns; the others are identically zero), DESIGN.md s3.9
    nets0 = opt.neural_networks[0]
    live_cols = 16 * int(nets0.x_blocks.numel()) if one_node and hasattr(nets0, "x_blocks") and nets0.x_blocks.numel() else 1008
    macs_live = {s: live_cols * a + a * b + b * c + c for s, (a, b, c) in enumerate(workloads.ANI2X_WIDTHS.values())}
    flops_executed = 2 * 2.0 * 8 * sum(macs_live[int(s)] for s in species)      # forward + input-gradient backward, fp32-equivalent
    tflops = flops_executed / elapsed * steps / 1e12         # EXECUTED flops: what `roofline.frac` prices
    tflops_issued = 3 * tflops                               # three fp16 matrix products per fp32 product
    split = args.nn_layout in ("fused", "gemm")
    kernel_name = {"fused": "mlp_forward + mlp_input_grad (mlp_fused.hip: a 64-atom tile of one species and one member through all "
                            "four layers in one workgroup, activations in LDS / registers)",
                   "gemm": "gemm_h2 (batched_nn.hip: one split-fp16 GEMM per layer and species, fused activations)",
                   "grouped": "BatchedNN GEMMs (hipBLASLt via torch.matmul)", "reference": "BatchedLinear on per-atom replicated weights"}[args.nn_layout]
    # HBM bytes of the network launches (this line's roofline is about them), counters of this run
    nn_traffic, nn_by_kernel = (side_traffic(args, R, "torchani", [("mlp_forward", 1), ("mlp_sum_members", 1)])
                                if args.nn_layout == "fused" else (None, None))
    out = {
        "metric": "OptimizedTorchANI energy+forces evaluations/sec, 2001-atom periodic water box, 8 models, fp32",
        "value": round(steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": steps, "warmup": warm, "timing_protocol": SIDE_PROTOCOL,
        "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + (" (network products: operands split into two fp16 planes, products exact, fp32 accumulation)" if split else ""),
        "data": "synthetic",
        "config": {"workload": f"OptimizedTorchANI, {n}-atom periodic water box (667 H2O), ANI-2x AEV + 8 x ANI-2x-shaped networks, "
                               f"random weights, BatchedNN layout = {args.nn_layout}, "
                               + ("AEV + networks as one autograd node (7 launches per energy+forces step)" if one_node else "four-module composition")
                               + (", replayed as one HIP graph" if args.graph else ""), "atoms": n,
                   "nn_weight_bytes": nn_weight_bytes, "nn_layout": args.nn_layout, "one_autograd_node": one_node,
                   "aev_columns": 1008, "aev_columns_multiplied": live_cols},
        "ms_per_step_without_capacity_check": (round(no_check_ms, 4) if no_check_ms is not None else None),
        "ms_per_step_as_hip_graph": (round(graph_ms, 4) if graph_ms is not None else None),
        "ms_per_energy_and_forces_call": (round(call_ms, 4) if call_ms is not None else None),
        "ms_per_step_with_first_layer_weights_x100": (round(rescaled_ms, 4) if rescaled_ms is not None else None),
        "act_scale_log2_with_first_layer_weights_x100": rescaled_log2,
        "ms_per_step_when_weights_fail_fused_ok": (round(unfused_ms, 4) if unfused_ms is not None else None),
        "layout_when_weights_fail_fused_ok": unfused_layout,
        "ms_per_step_as_hip_graph_with_dense_networks": (round(dense_graph_ms, 4) if dense_graph_ms is not None else None),
        "roofline": {"bound": "mfma", "kernel": kernel_name + ", forward + input-gradient backward",
                     "achieved": round(tflops, 3), "peak": FP32_MATRIX_PEAK, "unit": "TFLOP/s",
                     "frac": round(tflops / FP32_MATRIX_PEAK, 5), "traffic": nn_traffic, "traffic_by_kernel": nn_by_kernel,
                     "traffic_source": SIDE_TRAFFIC_SOURCE if nn_traffic is not None else None,
                     "issued": ({"instruction": "v_mfma_f32_16x16x32_f16, 3 products per fp32 product, over the live AEV columns only",
                                 "tflops": round(tflops_issued, 2), "peak": F16_DENSE_PEAK,
                                 "frac": round(tflops_issued / F16_DENSE_PEAK, 5)} if split else None),
                     "executed_gflop_per_step": round(flops_executed / 1e9, 3),
                     "vs_reference_formulation": {"tflops": round(tflops_dense, 3), "gflop_per_step": round(2 * flops_fwd / 1e9, 3),
                                                  "ratio_to_fp32_matrix_peak": round(tflops_dense / FP32_MATRIX_PEAK, 5),
                                                  "note": "the reference's dense BatchedLinear multiplies all 1008 AEV columns (9.8 GFLOP "
                                                          "forward, SURVEY s8(d)); this figure prices THOSE flops at this step's time -- it "
                                                          "is a comparison of formulations, not a roofline fraction, and may exceed 1"},
                     "note": "whole step time (neighbour search + AEV + networks, forward and backward) against the flops the networks "
                             "EXECUTE (layer 0 over the live AEV columns only: config.aev_columns_multiplied; the columns of absent species "
                             "are structurally zero and skipped), fp32-equivalent, priced at the fp32 matrix peak the reference's arithmetic "
                             "would run at; `issued` = the same flops x 3 (split-fp16: three v_mfma_f32_16x16x32_f16 products per fp32 "
                             "product) against the dense fp16 peak of the instruction actually issued"},
    }
    if not args.no_cpu_baseline:
        # the reference CPU AEV op (single thread) + BatchedLinear on the CPU.  The AEV leg is the
        # reference's own core compiled in place; the network leg is the reference's per-atom layout (BatchedNN.py:55-111:
        # matmul on [1, atoms, members, out, in] weights -- 10.8 MB per atom and member set) on a SAMPLE of atoms, scaled
        # by the atom count (the layout is per atom: the cost is linear in atoms by construction).
        kind, cls = _cpu_classes()
        rf, af = workloads.ani2x_functions()
        dt = _ani_eval_seconds(cls, pos, species, box, rf, af)
        sample = 150                                           # 50 waters: 1.6 GB of replicated weights on the host
        sp_s = species[:sample]
        numbers_s = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in sp_s]])
        nn_cpu = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers_s, layout="reference")
        aev_s = torch.rand((1, sample, 1008)).requires_grad_(True)
        sp_t = torch.tensor(sp_s).unsqueeze(0)
        threads = torch.get_num_threads()

        def nn_step():
            aev_s.grad = None
            nn_cpu((sp_t, aev_s)).energies.sum().backward()
        nn_step()
        t1 = time.perf_counter()
        for _ in range(3):
            nn_step()
        dt_nn = (time.perf_counter() - t1) / 3 * n / sample
        torch.set_num_threads(1)
        nn_step()
        t1 = time.perf_counter()
        nn_step()
        dt_nn1 = (time.perf_counter() - t1) * n / sample
        torch.set_num_threads(threads)
        # (`kind` of the whole figure is "port": only the AEV leg is the reference's own code; the networks leg times THIS repository's
        #  restatement of BatchedNN.cpp:30-47 on the host -- the reference's Python wrapper cannot be imported without torchani)
        out["cpu_baseline"] = {"value": round(1.0 / (dt + dt_nn1), 4), "unit": "evals/s (AEV + networks, fwd+bwd)", "cores": 1, "kind": "port",
                               "legs": {"aev": {"kind": kind, "seconds": round(dt, 3)},
                                        "networks": {"kind": "restatement", "seconds_1_thread": round(dt_nn1, 3)}},
                               "aev_seconds": round(dt, 3), "networks_seconds_1_thread": round(dt_nn1, 3),
                               "networks_seconds_all_threads": round(dt_nn, 3), "networks_threads": threads,
                               "sample": f"AEV: one fwd+bwd of the same {n}-atom water box by the reference's CPU core ({dt:.2f} s, serial). "
                                         f"Networks: the reference's per-atom BatchedLinear layout (matmul over [1, atoms, 8, out, in] weights, "
                                         f"fwd + input-gradient bwd) on the first {sample} atoms with ATen on the host, scaled by {n}/{sample}: "
                                         f"{dt_nn1:.2f} s with 1 thread, {dt_nn:.2f} s with {threads} threads (ATen's own threading; the op is this "
                                         "repository's restatement of BatchedNN.cpp:30-47 -- the same two ATen calls)"}
    return out


# =============================================================================================
# BASELINE config 4: 1 024 conformers, strong scaling over the ranks
# =============================================================================================
CONFORMER_BATCH = 1024


def conformer_sizes():
    import numpy as np
    return np.random.default_rng(5).integers(50, 71, size=CONFORMER_BATCH).tolist()


class ConformerShard:
    """Molecules [lo, hi) of the 1 024-conformer batch on one device: one batched handle, inputs resident, and the force
    buffer the kernels write into (`out`, a [rows, 3] view the caller provides or an own tensor)."""

    def __init__(self, sizes, lo, hi, device_index, seed_offset=0):
        import numpy as np
        import torch
        from nnpops_amd import workloads
        from nnpops_amd.capi import AniSymmetryFunctions
        dev = torch.device("cuda", device_index)
        self.mols = [workloads.conformer(sizes[m], seed=1000 + m) for m in range(lo, hi)]
        pos = np.concatenate([m[0] for m in self.mols]).astype(np.float32)
        species = np.concatenate([m[1] for m in self.mols]).astype(np.int32)
        offsets = np.concatenate([[0], np.cumsum(sizes[lo:hi])]).astype(np.int32)
        rf, af = workloads.ani2x_functions()
        self.sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, device=device_index)
        self.sym.set_molecules(offsets)
        self.n = n = pos.shape[0]
        self.tpos = torch.tensor(pos, device=dev)
        self.radial = torch.empty((n, self.sym.radial_width), device=dev)
        self.angular = torch.empty((n, self.sym.angular_width), device=dev)
        gen = torch.Generator(device=dev).manual_seed(7 + seed_offset)
        self.g_rad = torch.randn(self.radial.shape, device=dev, generator=gen)
        self.g_ang = torch.randn(self.angular.shape, device=dev, generator=gen)
        self.sym.compute(self.tpos, None, self.radial, self.angular, check=True)     # calibrates the capacities (blocks)

    def step(self, out):
        self.sym.compute(self.tpos, None, self.radial, self.angular, check=False)
        self.sym.backprop(self.g_rad, self.g_ang, out)


SIDE_PROTOCOL = ("settle (>= 0.2 s of untimed steps), then the best of 3 loops of `steps` steps, each behind `warmup` untimed steps and closed by "
                 "torch.cuda.synchronize()")


def _settle(fn, seconds=0.2, chunk=25):
    """Untimed steps until `seconds` of wall clock have passed: a device that has just been handed to this process, or has idled through a
    second of host-side set-up, runs its first loops up to 2.5x slow (the headline has its own --settle phase for the same reason)."""
    import torch
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize()


def _time_steps(fn, steps, warm, repeats=3, settle=True):
    """Seconds per call, ONE protocol for every side line and for both operands of every ratio between side lines:
    SIDE_PROTOCOL above -- so that the driver's `--steps 20 --warmup 5` reproduces what a 100-step run reports."""
    import torch
    if settle:
        _settle(fn)
    best = None
    for _ in range(repeats):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None else min(best, dt)
    return best


def run_conformers(args, R):
    """BASELINE config 4.  The batch is split into contiguous blocks balanced by atom count, one batched handle per rank;
    the forces of all ranks are assembled with ONE all_gather_into_tensor per step -- issued asynchronously on RCCL's stream
    into one of two preallocated padded buffer sets (the scheme of the headline: the gather of step s overlaps the kernels
    of step s + 1, a buffer is rewritten only after the gather that read it has been waited for on the stream, nothing is
    allocated or concatenated inside the loop).  value = batches / max-over-ranks time: strong scaling."""
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.parallel import molecule_work, shard_molecules
    rank, world, dev, dist = R.rank, R.world, R.dev, R.dist
    B = CONFORMER_BATCH
    sizes = conformer_sizes()
    # blocks balanced by WORK (neighbour triples + a share per atom, parallel.molecule_work: host numpy at set-up), not by atoms:
    # compact conformers hold up to 7 % more triples than loose ones of the same size and the slowest rank decides the step
    work = [molecule_work(workloads.conformer(sizes[m], seed=1000 + m)[0], workloads.ANI2X["Rca"]) for m in range(B)]
    blocks = shard_molecules(sizes, world, weights=work)
    offsets_all = np.concatenate([[0], np.cumsum(sizes)])
    rows = [int(offsets_all[hi] - offsets_all[lo]) for lo, hi in blocks]
    width = max(rows)
    lo, hi = blocks[rank]
    shard = ConformerShard(sizes, lo, hi, R.local_rank, seed_offset=rank)
    n = shard.n
    padded = [torch.zeros((width, 3), device=dev) for _ in range(2 if dist else 1)]       # the kernels write rows [0, n)
    gathered = [torch.empty((world * width, 3), device=dev) for _ in range(2)] if dist else None
    pending = [None, None]
    counter = [0]

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def step():
        b = counter[0] & 1 if dist else 0
        counter[0] += 1
        if dist and pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        shard.step(padded[b][:n])
        if dist:
            pending[b] = dist.all_gather_into_tensor(gathered[b], padded[b], async_op=True)

    steps, warm = min(args.steps, 100), min(args.warmup, 10)
    if not dist:
        elapsed = _time_steps(step, steps, warm) * steps     # (SIDE_PROTOCOL: the same as the eight blocks below)
    else:
        for _ in range(max(warm, 200)):                      # (a count, the same on every rank: every step holds a collective)
            step()
        drain()
        R.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        R.barrier()
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
    last = (counter[0] - 1) & 1 if dist else 0
    own = padded[last][:n]
    assert bool(torch.isfinite(own).all())
    if dist:        # every rank holds every block: its own bitwise, all of them finite
        g = gathered[last].view(world, width, 3)
        assert bool(torch.equal(g[rank, :n], own))
        assert all(bool(torch.isfinite(g[r, :rows[r]]).all()) for r in range(world))
    if rank != 0:
        return None
    atoms_total = int(offsets_all[-1])
    step_bytes = atoms_total * (16 + 2 * 1008 * 4 + 12)
    # HBM bytes of one batch step (every launch of the step), counters of this run
    conf_traffic, conf_by_kernel = side_traffic(args, R, "conformers", [("ani_radial_backward_lanes", 1), ("ani_neighbors_allpairs", 1),
                                                                       ("ani_angular_forward_mfma", 1), ("ani_angular_backward_pair", None)])
    out = {
        "metric": "AEV+forces evaluations/sec of a 1024-conformer batch (ANI-2x, ~60 atoms each)",
        "value": round(steps / elapsed, 3), "unit": "batch evals/s", "n_gpus": world, "steps": steps,
        "warmup": warm, "timing_protocol": SIDE_PROTOCOL if world == 1 else "200 untimed steps, then one loop of `steps` steps between barriers, max over ranks", "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ANI-2x AEV forward+backward, 1024 independent conformers of 50-70 atoms, contiguous batch "
                               "blocks per GPU, one batched handle per GPU, one asynchronous all_gather of the forces per step "
                               "(two buffer sets)", "conformers": B,
                   "atoms_total": atoms_total, "atoms_this_rank": n},
        "roofline": {"bound": "hbm", "kernel": "whole step (5 launches per GPU)", "achieved": round(step_bytes / elapsed * steps / 1e9, 2),
                     "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(step_bytes / elapsed * steps / 1e9 / (HBM_PEAK_GBS * world), 5),
                     "traffic": conf_traffic, "traffic_by_kernel": conf_by_kernel,
                     "traffic_source": SIDE_TRAFFIC_SOURCE if conf_traffic is not None else None, "algorithmic_bytes_per_launch": step_bytes},
    }
    if world == 1 and not args.no_shard8:
        # What one GPU of an 8-GPU node would run: every block of shard_molecules(sizes, 8) timed alone on THIS device
        # (same handle type, same step).  projected_scaling = t(1024 conformers, 1 GPU) / max over the 8 blocks -- the
        # all_gather (0.74 MB in all) is off the critical path when it overlaps the next step; the second figure adds a
        # synchronous 25 us per step for it (an assumption about a small RCCL all_gather over xGMI, not a measurement).
        t_full = elapsed / steps
        del shard
        t_blocks = []
        for r8, (l8, h8) in enumerate(shard_molecules(sizes, 8, weights=work)):
            sh = ConformerShard(sizes, l8, h8, R.local_rank, seed_offset=r8)
            buf = torch.empty((sh.n, 3), device=dev)
            t_blocks.append(_time_steps(lambda: sh.step(buf), steps, warm))
            del sh
        t_max = max(t_blocks)
        out["shard8"] = {"ms_per_block_step": [round(1e3 * t, 4) for t in t_blocks], "slowest_block_ms": round(1e3 * t_max, 4),
                         "projected_scaling": round(t_full / t_max, 2),
                         "projected_scaling_with_synchronous_gather": round(t_full / (t_max + 25e-6), 2),
                         "protocol": "both operands of the projections -- the whole batch (ms_per_step of this line) and every block -- are timed by the "
                                     "same function: " + SIDE_PROTOCOL,
                         "note": "each of the 8 blocks of shard_molecules(sizes, 8, weights = neighbour triples + 130 per atom) (~128 conformers, ~7.7 k atoms) timed alone on "
                                 "this one device; projected = t_1024 / slowest block (gather overlapped) and / (slowest block + "
                                 "25 us assumed for a synchronous all_gather); no 8-GPU node was available to measure it"}
    if not args.no_cpu_baseline and world == 1:
        # the reference has no batch dimension (SymmetryFunctions.py:110): a loop over per-molecule objects on one core
        kind, cls = _cpu_classes()
        sample = 64
        rf, af = workloads.ani2x_functions()
        mols = [workloads.conformer(sizes[m], seed=1000 + m) for m in range(sample)]
        t_cpu = sum(_ani_eval_seconds(cls, mols[m][0], mols[m][1], None, rf, af) for m in range(sample))
        out["cpu_baseline"] = {"value": round(1.0 / (t_cpu / sample * B), 4), "unit": "batch evals/s", "cores": 1, "kind": kind,
                               "sample": f"{sample} of the 1024 molecules, one object each, fwd+bwd: {1e6 * t_cpu / sample:.0f} us per molecule, "
                                         f"x 1024 = {t_cpu / sample * B:.2f} s per batch"}
    return out


# =============================================================================================
# BASELINE config 3: SchNet CFConv, 10 000-atom periodic box
# =============================================================================================
def run_cfconv(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    dev = R.dev
    n, W, G, cutoff, sigma = args.atoms, 128, 50, 5.0, 0.1
    pos, _, box = workloads.random_box(n, density=0.1, seed=3)
    rng = np.random.default_rng(4)
    w1 = (0.1 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.1 * rng.standard_normal((W, W))).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    nb = CFConvNeighbors(n, cutoff, periodic=True, device=R.local_rank)
    cf = CFConv(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True, device=R.local_rank)
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    out = torch.empty_like(tx)
    nb.build(tpos, tbox, check=True)
    pairs = nb.num_pairs()

    def step():
        nb.build(tpos, tbox, check=False)
        cf.compute(nb, tpos, tx, tbox, out)
        return cf.backprop(nb, tpos, tx, tg, tbox)

    steps, warm = min(args.steps, 200), min(args.warmup, 20)
    for _ in range(max(warm, 2)):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    # per-phase times of one eager step
    ev[0].record(); nb.build(tpos, tbox, check=False)
    ev[1].record(); cf.compute(nb, tpos, tx, tbox, out)
    ev[2].record(); eager_xg, eager_pg = cf.backprop(nb, tpos, tx, tg, tbox)
    ev[3].record()
    torch.cuda.synchronize()
    tb, tf, tbw = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
    if args.graph:
        # the nine launches of a step as one HIP graph (buffers were sized by the warm-up steps above)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_xg, g_pg = step()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g_xg, eager_xg) and torch.equal(g_pg, eager_pg)
        elapsed = _time_steps(graph.replay, steps, warm) * steps
    else:
        elapsed = _time_steps(step, steps, warm) * steps     # (SIDE_PROTOCOL)
    flops_fwd = 2.0 * (G * W + W * W) * pairs            # per half pair
    split = os.environ.get("NNPOPS_CFCONV_SPLIT", "1") != "0" and os.environ.get("NNPOPS_CFCONV_HALF", "1") != "0"
    tflops = flops_fwd / (tf * 1e-3) / 1e12
    # HBM bytes of the FORWARD kernels (this line's roofline is the forward of one layer) and of the backward ones, counters of this run;
    # the kernels of the two directions differ in a template argument (the mangled name of cfconv_filters_h2 carries ...Lb0E / ...Lb1E
    # for BWD behind the two integers, cfconv_gather prints <false / <true)
    fw_names = ("cfconv_filters_h2ILi0ELi8ELb0", "cfconv_gather<false") if split else ("cfconv_filters_mfmaILi0ELi8ELb0", "cfconv_gather<false")
    bw_names = ("cfconv_filters_h2ILi0ELi8ELb1", "cfconv_gather<true") if split else ("cfconv_filters_mfmaILi0ELi8ELb1", "cfconv_gather<true")
    cf_traffic, cf_by_kernel = side_traffic(args, R, "cfconv", [(fw_names[0], 1), (fw_names[1], 1)])
    cf_bwd_traffic, cf_bwd_by_kernel = side_traffic(args, R, "cfconv", [(bw_names[0], 1), (bw_names[1], 1)]) if cf_traffic is not None else (None, None)
    out_json = {
        "metric": "CFConv build+forward+backward evaluations/sec, W=128 G=50 cutoff 5 A, 10k-atom periodic box",
        "value": round(steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": steps, "warmup": warm, "timing_protocol": SIDE_PROTOCOL,
        "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + (" (dense layers: operands split into two fp16 planes, products exact, fp32 accumulation)" if split else ""),
        "data": "synthetic",
        "config": {"workload": f"SchNet CFConv + neighbour list, {n} atoms periodic, W={W}, G={G}, cutoff {cutoff} A, ssp"
                               + (", replayed as one HIP graph" if args.graph else ""), "half_pairs": pairs},
        "phases_ms": {"build": round(tb, 4), "forward": round(tf, 4), "backward": round(tbw, 4)},
        "roofline": {"bound": "mfma", "kernel": ("cfconv_filters_h2" if split else "cfconv_filters_mfma") + " + cfconv_gather (forward)",
                     "achieved": round(tflops, 3), "peak": FP32_MATRIX_PEAK, "unit": "TFLOP/s", "frac": round(tflops / FP32_MATRIX_PEAK, 5),
                     "traffic": cf_traffic, "traffic_by_kernel": cf_by_kernel,
                     "traffic_source": SIDE_TRAFFIC_SOURCE if cf_traffic is not None else None,
                     "backward": {"traffic": cf_bwd_traffic, "traffic_by_kernel": cf_bwd_by_kernel,
                                  "algorithmic_bytes": int(n * W * 4 * 3 + n * 24 + pairs * 24)},
                     "algorithmic_bytes_per_launch": int(n * W * 4 * 2 + pairs * 24),
                     "issued": ({"instruction": "v_mfma_f32_16x16x32_f16, 3 products per fp32 product", "tflops": round(3 * tflops, 2),
                                 "peak": F16_DENSE_PEAK, "frac": round(3 * tflops / F16_DENSE_PEAK, 5)} if split else None),
                     "note": "algorithmic flops (half-pair count) / measured forward time (filters kernel + gather kernel); `frac` is "
                             "against the fp32 matrix peak the reference's arithmetic would be priced at, `issued` against the dense "
                             "fp16 peak of the instruction actually issued"},
    }
    if not args.no_cpu_baseline:
        import oracle
        kind = "reference" if oracle.have_ref() else "port"
        NB, CF = (oracle.RefCFConvNeighbors, oracle.RefCFConv) if kind == "reference" else (oracle.CFConvNeighborsOracle, oracle.CFConvOracle)
        m = 2000                                        # bounded sample: the work is linear in N at fixed density
        pos_s, _, box_s = workloads.random_box(m, density=0.1, seed=3)
        onb = NB(m, cutoff, True)
        ocf = CF(m, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True)
        t1 = time.perf_counter()
        onb.build(pos_s, box_s)
        ocf.forward(onb, pos_s, x[:m], box_s)
        ocf.backward(onb, pos_s, x[:m], gy[:m], box_s)
        dt = time.perf_counter() - t1
        out_json["cpu_baseline"] = {"value": round(1.0 / dt * m / n, 5), "unit": "evals/s", "cores": 1, "kind": kind,
                                    "sample": f"one build+fwd+bwd of a {m}-atom box of the same density ({dt:.1f} s), scaled by "
                                              f"{m}/{n} atoms (pair count is linear in N at fixed density)"}
    return out_json


# =============================================================================================
# The reference's OWN CFConv benchmark (src/schnet/BenchmarkCudaCFConv.cu:62-110): width 128, 50 Gaussians, cutoff 10 A,
# Gaussian width 0.2, shifted softplus, N(0, 1) weights; one iteration = ONE neighbour build + SIX x (compute + backprop)
# -- the regime a SchNet runs in (3-6 interaction layers per list), where the build is amortised.
# =============================================================================================
def run_cfconv_reference(args, R):
    import numpy as np
    import torch
    from nnpops_amd import workloads
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    dev = R.dev
    n, W, G, cutoff, sigma, layers = args.atoms, 128, 50, 10.0, 0.2, 6
    pos, _, box = workloads.random_box(n, density=0.1, seed=3)
    rng = np.random.default_rng(0)
    w1, w2 = rng.standard_normal((W, G)).astype(np.float32), rng.standard_normal((W, W)).astype(np.float32)
    b1, b2 = rng.standard_normal(W).astype(np.float32), rng.standard_normal(W).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    nb = CFConvNeighbors(n, cutoff, periodic=True, device=R.local_rank)
    cf = CFConv(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True, device=R.local_rank)
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    out = torch.empty_like(tx)
    nb.build(tpos, tbox, check=True)
    pairs = nb.num_pairs()

    def iteration():
        nb.build(tpos, tbox, check=False)
        for _ in range(layers):
            cf.compute(nb, tpos, tx, tbox, out)
            xg, pg = cf.backprop(nb, tpos, tx, tg, tbox)
        return xg, pg

    xg, pg = iteration()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(xg).all()) and bool(torch.isfinite(pg).all())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); nb.build(tpos, tbox, check=False)
    ev[1].record(); cf.compute(nb, tpos, tx, tbox, out)
    ev[2].record(); cf.backprop(nb, tpos, tx, tg, tbox)
    ev[3].record()
    torch.cuda.synchronize()
    tb, tf, tbw = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
    iters = max(3, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / iters
    flops_fwd = 2.0 * (G * W + W * W) * pairs
    tflops = flops_fwd / (tf * 1e-3) / 1e12
    return {
        "metric": "iterations/sec of the reference's CFConv benchmark (1 neighbour build + 6 x (compute + backprop)), W=128 G=50 cutoff 10 A",
        "value": round(1.0 / elapsed, 3), "unit": "iterations/s", "n_gpus": 1, "steps": iters, "warmup": 1,
        "ms_per_step": round(1e3 * elapsed, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (dense layers: operands split into two fp16 planes when the weights keep them inside the fp16 range)", "data": "synthetic",
        "config": {"workload": f"BenchmarkCudaCFConv.cu:62-110 on a {n}-atom periodic box at 0.1 atoms/A^3: W={W}, G={G}, cutoff {cutoff} A, "
                               f"sigma {sigma}, ssp, N(0,1) weights, {layers} convolution layers (forward + backward) per neighbour build",
                   "half_pairs": pairs, "layers_per_build": layers},
        "phases_ms": {"build": round(tb, 4), "forward_one_layer": round(tf, 4), "backward_one_layer": round(tbw, 4)},
        "roofline": {"bound": "mfma", "kernel": "cfconv filters + gather (forward of one layer)", "achieved": round(tflops, 3),
                     "peak": FP32_MATRIX_PEAK, "unit": "TFLOP/s", "frac": round(tflops / FP32_MATRIX_PEAK, 5), "traffic": None,
                     "note": "algorithmic flops (half-pair count) / measured forward time of one layer, against the fp32 matrix peak"},
    }


WORKLOADS = {"aev": run_aev, "cfconv": run_cfconv, "cfconv_reference": run_cfconv_reference, "conformers": run_conformers, "neighbors": run_neighbors,
             "torchani": run_torchani, "latency": run_latency}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)     # 0.06 s of GPU time: long enough for clocks and caches to settle
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--atoms", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (HBM traffic, instruction counts) of the headline kernels")
    ap.add_argument("--settle", type=int, default=-1, help="untimed settling steps before the warm-up (-1: 2500 for frames up to 20 000 atoms)")
    ap.add_argument("--strict-side", action="store_true", help="exit with status 3 when a side workload failed (the line is still printed)")
    ap.add_argument("--graph", action="store_true", help="torchani / cfconv workloads: replay the step as one captured HIP graph")
    ap.add_argument("--no-shard8", action="store_true", help="conformers workload: skip the eight per-shard timings (profiles of the full batch alone)")
    ap.add_argument("--nn-layout", default="fused", choices=["fused", "gemm", "grouped", "reference"],
                    help="torchani workload: the fused network kernels (default), per-layer split-fp16 GEMMs, library GEMMs, or the "
                         "reference's per-atom replicated weights")
    ap.add_argument("--no-fused-step", action="store_true", help="torchani workload: the four-module composition instead of one autograd node")
    ap.add_argument("--neighbor-algorithm", type=int, default=0)
    ap.add_argument("--workload", default="aev", choices=sorted(WORKLOADS),
                    help="aev: the headline metric (default, with the others attached under 'side'); the rest run one BASELINE "
                         "configuration alone: latency = config 1, torchani = 2, cfconv = 3, conformers = 4, neighbors = 5")
    args = ap.parse_args()
    spawn_ranks_if_needed(args)
    R = Ranks()
    if args.workload != "aev":
        out = WORKLOADS[args.workload](args, R)
        if R.rank == 0:
            print(json.dumps(out), flush=True)
        R.close()
        return
    out = run_aev(args, R)
    side, side_errors = {}, {}
    if not args.no_side:
        sargs = argparse.Namespace(**vars(args))
        sargs.steps, sargs.warmup, sargs.atoms = min(args.steps, 100), min(args.warmup, 10), 10000
        names = ["conformers"] if R.world > 1 else ["latency", "torchani", "cfconv", "cfconv_reference", "conformers", "neighbors"]
        for name in names:
            try:
                res = WORKLOADS[name](sargs, R)
            except Exception as exc:                          # a side measurement must not cost the headline line -- but it must SHOW
                import traceback
                res = {"error": f"{type(exc).__name__}: {exc}"}
                side_errors[name] = res["error"]
                print(f"bench.py: side workload '{name}' FAILED on rank {R.rank}:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            if R.rank == 0:
                side[name] = res
    if R.rank == 0:
        if side:
            out["side"] = side
        out["side_errors"] = side_errors                      # {} when every side workload ran: the driver can key on it
        conf = side.get("conformers")
        if R.world > 1 and conf and "value" in conf:          # north_star's batch-parallel target, next to the weak-scaling headline
            out["conformers_strong_scaling"] = {"metric": conf["metric"], "value": conf["value"], "unit": conf["unit"],
                                                "ms_per_step": conf["ms_per_step"], "n_gpus": conf["n_gpus"], "scaling": "strong"}
        print(json.dumps(out), flush=True)
    R.close()
    if side_errors and args.strict_side:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
