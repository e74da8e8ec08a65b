#!/usr/bin/env python3
"""Interleaved A/B timing of ANI kernel variants in ONE process (box-to-box and DVFS noise is ~1 us per kernel, so variants
must be measured side by side).  Every variant is a set of environment variables read at handle creation.

    python tools/ab.py "NNPOPS_ANI_FWD_CHUNK=128" "NNPOPS_ANI_FWD_CHUNK=192" [--atoms 10000] [--rounds 9] [--steps 40] [--species 7]
    python tools/ab.py "LIB=tools/_ref/libnnpops_hip.so" ""      # another BUILD of the library against the current one
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnpops_amd import workloads  # noqa: E402
from nnpops_amd.capi import AniSymmetryFunctions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--atoms", type=int, default=10000)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--water", action="store_true", help="a water box (2 species present) instead of 7 uniform species")
    ap.add_argument("--workload", default="box", choices=["box", "latency", "block", "batch"],
                    help="box: periodic liquid of --atoms atoms; latency: the 50-atom conformer of BASELINE config 1; block / batch: 128 / "
                         "1 024 conformers of config 4 in one batched handle")
    ap.add_argument("--block", type=int, default=-1, help="block workload: the k-th of the eight work-balanced blocks of config 4 (default: the first 128 conformers)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    offsets = None
    if args.workload == "latency":
        pos, species = workloads.conformer(50, seed=0)
        box = None
    elif args.workload in ("block", "batch"):
        import bench
        sizes = bench.conformer_sizes()
        first = 0
        if args.workload == "block" and args.block >= 0:
            from nnpops_amd.parallel import molecule_work, shard_molecules
            work = [molecule_work(workloads.conformer(sizes[m], seed=1000 + m)[0], 3.5) for m in range(len(sizes))]
            first, last = shard_molecules(sizes, 8, weights=work)[args.block]
            sizes = sizes[first:last]
        else:
            sizes = sizes[:128 if args.workload == "block" else 1024]
        mols = [workloads.conformer(sizes[m], seed=1000 + first + m) for m in range(len(sizes))]
        pos = np.concatenate([m[0] for m in mols]).astype(np.float32)
        species = np.concatenate([m[1] for m in mols]).astype(np.int32)
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        box = None
    elif args.water:
        pos, species, box = workloads.water_box(args.atoms // 3, seed=1)
    else:
        pos, species, box = workloads.random_box(args.atoms, density=0.1, seed=100, n_species=7)
    n = len(species)
    rf, af = workloads.ani2x_functions()
    tpos = torch.tensor(pos, device=dev)
    tbox = torch.tensor(box, device=dev) if box is not None else None
    handles = []
    from nnpops_amd import capi
    product = capi.lib()
    for v in args.variants:
        saved = dict(os.environ)
        capi._lib = product
        for kv in v.split():
            if "=" in kv:
                k, val = kv.split("=", 1)
                if k == "LIB":                                 # another build of the library (an older one: no source-hash check)
                    import ctypes as C
                    other = C.CDLL(os.path.abspath(val))
                    for name, (restype, argtypes) in capi.SIGNATURES.items():
                        if hasattr(other, name):
                            fn = getattr(other, name)
                            fn.restype, fn.argtypes = restype, argtypes
                    capi._lib = other
                else:
                    os.environ[k] = val
        sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=box is not None)
        if offsets is not None:
            sym.set_molecules(offsets)
        capi._lib = product
        os.environ.clear(); os.environ.update(saved)
        radial = torch.empty((n, sym.radial_width), device=dev)
        angular = torch.empty((n, sym.angular_width), device=dev)
        g_r, g_a = torch.randn_like(radial), torch.randn_like(angular)
        grad = torch.empty((n, 3), device=dev)
        sym.compute(tpos, tbox, radial, angular, check=True)
        sym.compute(tpos, tbox, radial, angular, check=True)   # (a schedule the first check prepared is in place from the next compute on)
        handles.append((v, sym, radial, angular, g_r, g_a, grad))
    results = {v: [] for v in args.variants}
    walls = {v: [] for v in args.variants}
    for r in range(args.rounds):
        for v, sym, radial, angular, g_r, g_a, grad in handles:
            sym.enable_timing(True)
            for _ in range(args.steps):
                sym.compute(tpos, tbox, radial, angular, check=False)
                sym.backprop(g_r, g_a, grad)
            t = sym.get_timing()
            sym.enable_timing(False)
            results[v].append({k: 1e3 * ms / max(c, 1) for k, (ms, c) in t.items()})
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                sym.compute(tpos, tbox, radial, angular, check=False)
                sym.backprop(g_r, g_a, grad)
            e1.record(); torch.cuda.synchronize()
            walls[v].append(1e3 * e0.elapsed_time(e1) / args.steps)
    ovh = 1e6 * handles[0][1].timing_overhead()
    for v in args.variants:
        med = {k: float(np.median([x[k] for x in results[v]])) - ovh for k in results[v][0] if results[v][0][k] > 0}
        print(f"{v or '(default)':60s} step {np.median(walls[v]):7.2f} us | " + "  ".join(f"{k} {val:6.2f}" for k, val in med.items()))


if __name__ == "__main__":
    main()
