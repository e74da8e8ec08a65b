#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X NNPOps hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--atoms 10000] [--no-cpu-baseline]

Workload (BASELINE.json metric: "AEV+forces (energy+grad evals/sec) per GPU, 10k-atom box"):
one *step* = one full ANI-2x symmetry-function evaluation of a 10 000-atom periodic box --
neighbour search + radial/angular forward (the 1008-wide AEV of every atom) + backward
(dE/dpositions for a fixed dense upstream gradient dE/dAEV) -- with positions, species, box and
the upstream gradient already resident in HBM.  Everything goes through the C ABI
(include/nnpops_hip.h) on the current HIP stream; there is no host synchronisation inside the
timed region.  Neighbour-buffer capacity is verified before and after the timed region.

Multi-GPU (--gpus N under torch.distributed.run): the path shards over independent frames, so
every rank evaluates its own 10k-atom frame (different seed) with no data-path collective
("scaling": "weak"); value = total evaluations of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nnpops_amd import workloads  # noqa: E402
from nnpops_amd.capi import AniSymmetryFunctions  # noqa: E402

ROOFLINE_KERNELS = ("neighbors", "angular_forward", "angular_backward", "radial_backward")   # candidates for "dominant"
ROCPROF_NAME = {"neighbors": "ani_neighbors_cells (neighbour rows + radial AEV)", "angular_forward": "ani_angular_forward",
                "angular_backward": "ani_angular_backward", "radial_backward": "ani_radial_backward (+ force gather)"}

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(pos, species, box, rf, af, budget_s=25.0):
    """Time the reference CPU path (oracle/_ref, the reference's own sources compiled in place) --
    or, where that library is absent, this repository's C restatement -- on ONE host core, on a
    bounded sample: as many fwd+bwd evaluations of the SAME 10k-atom frame as fit the budget
    (at least one)."""
    import oracle
    kind = "reference" if oracle.have_ref() else "port"
    cls = oracle.RefAni if kind == "reference" else oracle.AniOracle
    n = pos.shape[0]
    obj = cls(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=True)
    rng = np.random.default_rng(123)
    wr = wa = None
    evals, t_total = 0, 0.0
    while True:
        t0 = time.perf_counter()
        r, a = obj.forward(pos, box)
        if wr is None:
            wr = rng.standard_normal(r.shape).astype(np.float32)
            wa = rng.standard_normal(a.shape).astype(np.float32)
        obj.backward(wr, wa)
        dt = time.perf_counter() - t0
        evals += 1
        t_total += dt
        if t_total + dt > budget_s:
            break
    return {"value": evals / t_total, "unit": "evals/s", "cores": 1, "kind": kind,
            "sample": f"{evals} fwd+bwd evaluation(s) of the same {n}-atom ANI-2x periodic frame, single thread "
                      f"({t_total:.1f} s; host has {os.cpu_count()} cores, the reference CPU path is serial)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)     # 0.07 s of GPU time: long enough for clocks and caches to settle
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--atoms", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="torchani / cfconv workloads: replay the step as one captured HIP graph")
    ap.add_argument("--nn-layout", default="fused", choices=["fused", "grouped", "reference"],
                    help="torchani workload: species-grouped GEMMs (default) or the reference's per-atom replicated weights")
    ap.add_argument("--neighbor-algorithm", type=int, default=0)
    ap.add_argument("--workload", default="aev", choices=["aev", "cfconv", "conformers", "neighbors", "torchani"],
                    help="aev: the headline metric (default); cfconv: BASELINE config 3; conformers: BASELINE config 4; "
                         "neighbors: BASELINE config 5; torchani: BASELINE config 2 (side measurements, same JSON shape)")
    args = ap.parse_args()
    if args.workload == "cfconv":
        return main_cfconv(args)
    if args.workload == "conformers":
        return main_conformers(args)
    if args.workload == "neighbors":
        return main_neighbors(args)
    if args.workload == "torchani":
        return main_torchani(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- synthetic frame of this rank ----
    n = args.atoms
    pos, species, box = workloads.random_box(n, density=0.1, seed=100 + rank, n_species=7)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=True,
                               device=local_rank)
    if args.neighbor_algorithm:
        sym.set_neighbor_algorithm(args.neighbor_algorithm)
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev)
    radial = torch.empty((n, sym.radial_width), device=dev)
    angular = torch.empty((n, sym.angular_width), device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    g_rad = torch.randn(radial.shape, device=dev, generator=gen)
    g_ang = torch.randn(angular.shape, device=dev, generator=gen)
    grad = torch.empty((n, 3), device=dev)

    def step():
        sym.compute(tpos, tbox, radial, angular, check=False)
        sym.backprop(g_rad, g_ang, grad)

    sym.compute(tpos, tbox, radial, angular, check=True)     # calibrates neighbour capacity (blocks)
    # Warm-up, with events around EVERY kernel: the per-kernel breakdown (diagnostic) and the choice of the
    # dominant kernel.  An event pair costs ~3 us of stream time, so inside the timed region only the dominant
    # kernel -- the one the roofline line is about -- is bracketed.
    sym.enable_timing(True)
    for _ in range(args.warmup):
        step()
    breakdown = sym.get_timing() if args.warmup else {}
    sym.enable_timing(False)
    if not args.warmup:
        step()
    kern_all = {k: (1e-3 * ms / max(c, 1)) for k, (ms, c) in breakdown.items()}
    dominant = max(ROOFLINE_KERNELS, key=lambda k: kern_all.get(k, 0.0))
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # what an event pair reports for an EMPTY bracket on this stream (marker processing, not kernel time): the
    # per-kernel figures below are net of it, which is what makes them agree with rocprofv3's kernel durations
    event_overhead = sym.timing_overhead()                                 # seconds
    sym.enable_timing(True, only=[dominant])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = sym.get_timing()
    sym.enable_timing(False)
    max_row, max_ang = sym.neighbor_stats()                   # raises nothing; verify no overflow happened
    from nnpops_amd.capi import lib, OK
    assert lib().nnpops_ani_check(sym._h, None, None) == OK, "neighbour buffers overflowed inside the timed region"
    assert bool(torch.isfinite(grad).all())

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        # roofline of the dominant kernel family: the angular kernels move one 896-float row per atom
        # (forward: written once; backward: read once) + positions/species.  SURVEY.md s8(d):
        # forward N*16 + N*896*4 bytes, backward N*896*4 + N*12 bytes.
        kern = {k: max(v - event_overhead, 0.0) if v > 0 else 0.0 for k, v in kern_all.items()}   # s per launch (warm-up pass)
        ms_dom, c_dom = timing[dominant]
        kern[dominant] = max(1e-3 * ms_dom / max(c_dom, 1) - event_overhead, 1e-9)  # ... the dominant one from the timed region
        # Algorithmic bytes per launch: unique bytes in + bytes out, no re-reads.
        #   angular forward   N*16 (records) + N*896*4 (row written once)
        #   angular backward  N*896*4 (upstream row read once) + N*12
        #   neighbours        N*16 in (cell-ordered positions) + per atom: row <n_Rcr>*16, records <n_Rca>*36,
        #                     triple list <triples>*4, radial AEV S*nR*4, counts 8   (liquid-density means of s8)
        #   radial backward   N*S*nR*4 (gradient row) + row <n_Rcr>*16 + legs <n_Rca>*20 + N*12
        nb_na, nb_nr = sym.angular_width, sym.radial_width
        n_rcr, n_rca, n_tri = 55.6, 18.0, 153.0
        alg_bytes = {"angular_forward": n * 16 + n * nb_na * 4, "angular_backward": n * nb_na * 4 + n * 12,
                     "neighbors": int(n * (16 + n_rcr * 16 + n_rca * 36 + n_tri * 4 + nb_nr * 4 + 8)),
                     "radial_backward": int(n * (nb_nr * 4 + n_rcr * 16 + n_rca * 20 + 12))}
        achieved = alg_bytes[dominant] / kern[dominant] / 1e9 if kern[dominant] > 0 else 0.0
        traffic = None                                           # HBM bytes/launch from the committed PMC passes
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("atoms") == n:
                traffic = tj.get(dominant)
        except (OSError, ValueError):
            pass
        out = {
            "metric": "AEV+forces evaluations/sec (ANI-2x symmetry functions, energy+gradient), 10k-atom periodic box",
            "value": round(value, 3), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ANI-2x AEV forward+backward, {n}-atom periodic cubic box at 0.1 atoms/A^3, "
                                   "7 species uniform, Rcr 5.1 / Rca 3.5, 16 radial + 32 angular functions (AEV width 1008); "
                                   "one independent frame per GPU",
                       "atoms": n, "frames_per_gpu": 1, "max_neighbors_rcr": max_row, "max_neighbors_rca": max_ang},
            "kernels_us": {k: round(1e6 * v, 2) for k, v in kern.items()},
            "event_pair_overhead_us": round(1e6 * event_overhead, 2),
            "roofline": {"bound": "hbm", "kernel": ROCPROF_NAME.get(dominant, dominant), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes[dominant]},
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(pos, species, box, rf, af)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main_neighbors(args):
    """BASELINE config 5: getNeighborPairs (cutoff 5.2 A, compact mode) + ANI-2x AEV on the same 100 000-atom
    periodic box (uniform 0.1 atoms/A^3, seed 6).  The reference cannot run this size at all (O(N^2) pair index
    overflows int32, getNeighborPairsCUDA.cu:129).  One step = one neighbour-pair list + one AEV forward+backward.
    Side measurement (not the headline metric)."""
    from nnpops_amd.capi import neighbor_pairs_forward
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n = args.atoms if args.atoms != 10000 else 100000
    cutoff, max_pairs = 5.2, int(32 * n)
    pos, species, box = workloads.random_box(n, density=0.1, seed=6, n_species=7)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, periodic=True, device=local_rank)
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

    def step(ev=None):
        if ev: ev[0].record()
        neighbor_pairs_forward(tpos, cutoff, max_pairs, tbox)
        if ev: ev[1].record()
        sym.compute(tpos, tbox, radial, angular, check=False)
        if ev: ev[2].record()
        sym.backprop(g_rad, g_ang, grad)
        if ev: ev[3].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    step(ev)                                           # one extra, event-bracketed step for the phase split
    torch.cuda.synchronize()
    t_nb, t_fwd, t_bwd = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
    nb_bytes = n * 12 + found * 24                      # positions in, (2 ints + 3 floats + 1 float) per pair out
    aev_bytes = n * (16 + 2 * (sym.radial_width + sym.angular_width) * 4 + 12)
    print(json.dumps({
        "metric": "getNeighborPairs + ANI-2x AEV forward+backward evaluations/sec, 100k-atom periodic box, cutoff 5.2 A",
        "value": round(args.steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"getNeighborPairs(cutoff {cutoff}, max_num_pairs {max_pairs}) + ANI-2x AEV, {n} atoms periodic, "
                               "0.1 atoms/A^3, 7 species", "atoms": n, "pairs_found": found},
        "phases_ms": {"neighbor_pairs": round(t_nb, 4), "aev_forward": round(t_fwd, 4), "aev_backward": round(t_bwd, 4)},
        "roofline": {"bound": "hbm", "kernel": "getNeighborPairs (3 launches + cell grid)", "achieved": round(nb_bytes / (t_nb * 1e-3) / 1e9, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb_bytes / (t_nb * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "traffic": None, "algorithmic_bytes_per_launch": nb_bytes,
                     "aev_step": {"algorithmic_bytes": aev_bytes,
                                  "achieved": round(aev_bytes / ((t_fwd + t_bwd) * 1e-3) / 1e9, 2)}},
    }), flush=True)


def main_torchani(args):
    """BASELINE config 2: OptimizedTorchANI (species converter + HIP AEV + BatchedNN + energy shifter) on a
    2 001-atom periodic water box, fp32, 8 models with the ANI-2x layer widths and random weights (torchani and
    its parameters are not available offline).  One step = energy forward + backward to the forces, through the
    torch.ops / autograd surface exactly as a user calls it.  Side measurement (not the headline metric)."""
    sys.path.insert(0, ROOT)
    from NNPOps import OptimizedTorchANI
    from NNPOps.BatchedNN import TorchANIBatchedNN
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    model = workloads.torchani_like_model(n_models=8, seed=2)
    pos, species, box = workloads.water_box(667, seed=1)
    numbers = torch.tensor([[workloads.Z_OF_SPECIES[s] for s in species]], device=dev)
    opt = OptimizedTorchANI(model, numbers.cpu())
    if args.nn_layout != "fused":
        opt.neural_networks = TorchANIBatchedNN(model.species_converter, model.neural_networks, numbers.cpu(), layout=args.nn_layout)
    opt = opt.to(dev)
    # (pbc stays on the host: the wrapper reads it with .tolist(), reference SymmetryFunctions.py:113, which on a
    # device tensor is a synchronising copy and cannot be captured)
    cell, pbc = torch.tensor(box, device=dev), torch.tensor([True, True, True])
    tpos = torch.tensor(pos, device=dev).unsqueeze(0).requires_grad_(True)
    n = len(species)

    def step():
        tpos.grad = None
        energy = opt((numbers, tpos), cell, pbc).energies
        energy.sum().backward()
        return energy

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.graph:
        # the whole energy+forces step as one HIP graph (the AEV holder skips its capacity check while capturing;
        # capacities were calibrated by the warm-up steps above): removes the ~70 host launches per step
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
        eager_e = step().detach().clone()
        eager_g = tpos.grad.detach().clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.allclose(g_energy, eager_e, rtol=1e-6, atol=1e-4) and torch.allclose(g_forces, eager_g, rtol=1e-4, atol=1e-5)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        energy = g_energy
        tpos.grad = g_forces
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            energy = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(energy).all()) and bool(torch.isfinite(tpos.grad).all())
    # NN flops: 2 * models * sum over atoms of the MACs of its network; backward to the
    # inputs costs the same again
    macs = {s: 1008 * a + a * b + b * c + c for s, (a, b, c) in enumerate(workloads.ANI2X_WIDTHS.values())}
    flops_fwd = 2.0 * 8 * sum(macs[int(s)] for s in species)
    nn_weight_bytes = sum(b.numel() * 4 for name, b in opt.neural_networks.named_buffers() if "layer" in name)
    print(json.dumps({
        "metric": "OptimizedTorchANI energy+forces evaluations/sec, 2001-atom periodic water box, 8 models, fp32",
        "value": round(args.steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + (" (network GEMMs: operands split into two fp16 planes, products exact, fp32 accumulation)" if args.nn_layout == "fused" else ""),
        "data": "synthetic",
        "config": {"workload": f"OptimizedTorchANI, {n}-atom periodic water box (667 H2O), ANI-2x AEV + 8 x ANI-2x-shaped networks, "
                               f"random weights, BatchedNN layout = {args.nn_layout}" + (", replayed as one HIP graph" if args.graph else ""), "atoms": n,
                   "nn_weight_bytes": nn_weight_bytes},
        "roofline": {"bound": "mfma", "kernel": ("gemm_h2 (batched_nn.hip: split-fp16 GEMMs with fused activations)" if args.nn_layout == "fused"
                                                else "BatchedNN GEMMs (hipBLASLt via torch.matmul)") + ", forward + input-gradient backward",
                     "achieved": round(2 * flops_fwd / elapsed * args.steps / 1e12, 3), "peak": 157.3, "unit": "TFLOP/s",
                     "frac": round(2 * flops_fwd / elapsed * args.steps / 1e12 / 157.3, 5), "traffic": None,
                     "note": "whole step time (AEV + NN + autograd overhead) against the NN's algorithmic flops; fp32 matrix peak"},
    }), flush=True)


def main_conformers(args):
    """BASELINE config 4: ANI-2x AEV forward+backward on 1 024 independent ~60-atom conformers, sharded over
    the ranks by contiguous blocks of the batch (strong scaling: total work is fixed).  Each rank evaluates its
    block with ONE batched handle (nnpops_ani_set_molecules); the only collective is the final all_gather of the
    per-atom forces (RCCL over xGMI; ~0.74 MB in total)."""
    from nnpops_amd.parallel import gather_rows, shard_molecules
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    B = 1024
    rng = np.random.default_rng(5)
    sizes = rng.integers(50, 71, size=B).tolist()
    blocks = shard_molecules(sizes, world)
    offsets_all = np.concatenate([[0], np.cumsum(sizes)])
    rows = [int(offsets_all[hi] - offsets_all[lo]) for lo, hi in blocks]
    lo, hi = blocks[rank]
    mols = [workloads.conformer(sizes[m], seed=1000 + m) for m in range(lo, hi)]
    pos = np.concatenate([m[0] for m in mols]).astype(np.float32)
    species = np.concatenate([m[1] for m in mols]).astype(np.int32)
    offsets = (offsets_all[lo:hi + 1] - offsets_all[lo]).astype(np.int32)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, workloads.ANI2X["Rcr"], workloads.ANI2X["Rca"], species, rf, af, device=local_rank)
    sym.set_molecules(offsets)
    n = pos.shape[0]
    tpos = torch.tensor(pos, device=dev)
    radial = torch.empty((n, sym.radial_width), device=dev)
    angular = torch.empty((n, sym.angular_width), device=dev)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    g_rad = torch.randn(radial.shape, device=dev, generator=gen)
    g_ang = torch.randn(angular.shape, device=dev, generator=gen)
    grad = torch.empty((n, 3), device=dev)

    def step():
        sym.compute(tpos, None, radial, angular, check=False)
        sym.backprop(g_rad, g_ang, grad)
        return gather_rows(grad, rows) if dist else grad

    sym.compute(tpos, None, radial, angular, check=True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        forces = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert forces.shape[0] == int(offsets_all[-1]) and bool(torch.isfinite(forces).all())
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "AEV+forces evaluations/sec of a 1024-conformer batch (ANI-2x, ~60 atoms each)",
            "value": round(args.steps / elapsed, 3), "unit": "batch evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ANI-2x AEV forward+backward, 1024 independent conformers of 50-70 atoms, contiguous batch "
                                   "blocks per GPU, one all_gather of the forces per step", "conformers": B,
                       "atoms_total": int(offsets_all[-1]), "atoms_this_rank": n}}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main_cfconv(args):
    """BASELINE config 3: CFConv + CFConvNeighbors, W=128, G=50, 5 A cutoff, 10 000-atom periodic box.
    One step = neighbour build + forward + backward.  Side measurement (not the headline metric)."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n, W, G, cutoff, sigma = args.atoms, 128, 50, 5.0, 0.1
    pos, _, box = workloads.random_box(n, density=0.1, seed=3)
    rng = np.random.default_rng(4)
    w1 = (0.1 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (0.1 * rng.standard_normal((W, W))).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    gy = rng.standard_normal((n, W)).astype(np.float32)
    nb = CFConvNeighbors(n, cutoff, periodic=True, device=local_rank)
    cf = CFConv(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True, device=local_rank)
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
    out = torch.empty_like(tx)
    nb.build(tpos, tbox, check=True)
    pairs = nb.num_pairs()

    def step():
        nb.build(tpos, tbox, check=False)
        cf.compute(nb, tpos, tx, tbox, out)
        return cf.backprop(nb, tpos, tx, tg, tbox)

    for _ in range(args.warmup):
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
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    flops_fwd = 2.0 * (G * W + W * W) * pairs            # per half pair
    split = os.environ.get("NNPOPS_CFCONV_SPLIT", "1") != "0" and os.environ.get("NNPOPS_CFCONV_HALF", "1") != "0"
    out_json = {
        "metric": "CFConv build+forward+backward evaluations/sec, W=128 G=50 cutoff 5 A, 10k-atom periodic box",
        "value": round(args.steps / elapsed, 3), "unit": "evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + (" (dense layers: operands split into two fp16 planes, products exact, fp32 accumulation)" if split else ""),
        "data": "synthetic",
        "config": {"workload": f"SchNet CFConv + neighbour list, {n} atoms periodic, W={W}, G={G}, cutoff {cutoff} A, ssp"
                               + (", replayed as one HIP graph" if args.graph else ""), "half_pairs": pairs},
        "phases_ms": {"build": round(tb, 4), "forward": round(tf, 4), "backward": round(tbw, 4)},
        "roofline": {"bound": "mfma", "kernel": ("cfconv_filters_h2" if split else "cfconv_filters_mfma") + " + cfconv_gather (forward)", "achieved": round(flops_fwd / (tf * 1e-3) / 1e12, 3),
                     "peak": 157.3, "unit": "TFLOP/s", "frac": round(flops_fwd / (tf * 1e-3) / 1e12 / 157.3, 5), "traffic": None,
                     "note": "algorithmic flops (half-pair count) / measured forward time (filters kernel + gather kernel); fp32 matrix peak"},
    }
    if not args.no_cpu_baseline:
        import oracle
        kind = "reference" if oracle.have_ref() else "port"
        NB, CF = (oracle.RefCFConvNeighbors, oracle.RefCFConv) if kind == "reference" else (oracle.CFConvNeighborsOracle, oracle.CFConvOracle)
        m = 2000                                        # bounded sample: the first 2000 atoms' worth of work scales ~linearly
        pos_s, _, box_s = workloads.random_box(m, density=0.1, seed=3)
        onb = NB(m, cutoff, True)
        ocf = CF(m, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True)
        t1 = time.perf_counter()
        onb.build(pos_s, box_s)
        y = ocf.forward(onb, pos_s, x[:m], box_s)
        ocf.backward(onb, pos_s, x[:m], gy[:m], box_s)
        dt = time.perf_counter() - t1
        out_json["cpu_baseline"] = {"value": round(1.0 / dt * m / n, 5), "unit": "evals/s", "cores": 1, "kind": kind,
                                    "sample": f"one build+fwd+bwd of a {m}-atom box of the same density ({dt:.1f} s), scaled by "
                                              f"{m}/{n} atoms (pair count is linear in N at fixed density)"}
    print(json.dumps(out_json), flush=True)


if __name__ == "__main__":
    main()
