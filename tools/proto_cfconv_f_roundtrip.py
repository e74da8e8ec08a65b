#!/usr/bin/env python3
"""Upper bound on what "not round-tripping F" could gain the SchNet CFConv at BASELINE config 3 ("one real
attempt at not round-tripping F ... or a committed prototype measurement showing the bound, as you did for f1").

The filters kernel writes one filter row F[pair] (512 B at W = 128) per half pair and the gather reads it from both ends of the pair
(DESIGN.md s3.6): 133 MB written, 2 x 135 MB read at 10 000 atoms.  A variant that keeps F on chip -- spatial tiles with both
owners' accumulators on one CU, boundary pairs computed twice -- would remove that traffic and pay for it with duplicated matrix
work.  This tool measures the CEILING of the gain before any of the cost: a second copy of the library, built from a PATCHED copy of
the kernel sources (never the product sources), in which every filter row is written to and read from the same 16 rows -- 8 KB that
never leave the L2: every instruction of the product kernels is still executed (the stores, the loads, the address arithmetic), only
the fabric traffic of F is gone.  The results of that build are wrong by construction; the numbers are durations only.

    python tools/proto_cfconv_f_roundtrip.py --build-only tools/_proto_cf      # here (hipcc, no GPU)
    python tools/proto_cfconv_f_roundtrip.py --lib tools/_proto_cf/libnnpops_hip.so      # on the GPU box; prints one JSON line
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATCHES = {
    "cfconv.hip": [
        ("                float* frow = filt + (size_t)p * W + col;\n",
         "                float* frow = filt + (size_t)(p & 15) * W + col;\n", 2),
        ("            const size_t fo = (size_t)__builtin_amdgcn_readlane(my_p, q) * W, vo = (size_t)__builtin_amdgcn_readlane(my_j, q) * W;\n",
         "            const size_t fo = (size_t)(__builtin_amdgcn_readlane(my_p, q) & 15) * W, vo = (size_t)__builtin_amdgcn_readlane(my_j, q) * W;\n", 1),
    ],
}


def build_variant(outdir):
    from nnpops_amd import build as hb
    os.makedirs(outdir, exist_ok=True)
    src_dir = os.path.join(outdir, "src", "nnpops_amd", "csrc")          # (host_common.h includes ../../include/nnpops_hip.h)
    shutil.rmtree(os.path.join(outdir, "src"), ignore_errors=True)
    shutil.copytree(hb.CSRC, src_dir, ignore=shutil.ignore_patterns("_obj"))
    os.makedirs(os.path.join(outdir, "src", "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "nnpops_hip.h"), os.path.join(outdir, "src", "include", "nnpops_hip.h"))
    for name, edits in PATCHES.items():
        path = os.path.join(src_dir, name)
        text = open(path).read()
        for anchor, replacement, count in edits:
            assert text.count(anchor) == count, f"prototype patch: anchor found {text.count(anchor)}x (want {count}) in {name}"
            text = text.replace(anchor, replacement)
        open(path, "w").write(text)
    objs, procs = [], []
    for unit in hb.UNITS:
        obj = os.path.join(outdir, unit + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                       f'-DNNPOPS_SOURCE_HASH="{hb.source_hash()}"', "-c", os.path.join(src_dir, unit), "-o", obj]))
    for p in procs:
        assert p.wait() == 0
    lib = os.path.join(outdir, "libnnpops_hip.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    shutil.rmtree(os.path.join(outdir, "src"), ignore_errors=True)
    return lib


def measure(lib_path):
    """build / forward / backward of BASELINE config 3 (10 000 atoms, W = 128, G = 50, 5 A) with the given library."""
    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from nnpops_amd import capi, workloads
capi.LIB_PATH = %r
from nnpops_amd.capi import CFConv, CFConvNeighbors
dev = torch.device("cuda:0")
n, W, G, cutoff, sigma = 10000, 128, 50, 5.0, 0.1
pos, _, box = workloads.random_box(n, density=0.1, seed=3)
rng = np.random.default_rng(4)
w1 = (0.1 * rng.standard_normal((W, G))).astype(np.float32); w2 = (0.1 * rng.standard_normal((W, W))).astype(np.float32)
b1 = (0.1 * rng.standard_normal(W)).astype(np.float32); b2 = (0.1 * rng.standard_normal(W)).astype(np.float32)
x = rng.standard_normal((n, W)).astype(np.float32); gy = rng.standard_normal((n, W)).astype(np.float32)
nb = CFConvNeighbors(n, cutoff, periodic=True)
cf = CFConv(n, W, G, cutoff, sigma, "ssp", w1, b1, w2, b2, periodic=True)
tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
tx, tg = torch.tensor(x, device=dev), torch.tensor(gy, device=dev)
out = torch.empty_like(tx)
nb.build(tpos, tbox, check=True)
def t(fn, reps=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
fwd = lambda: cf.compute(nb, tpos, tx, tbox, out)
bwd = lambda: cf.backprop(nb, tpos, tx, tg, tbox)
fwd(); bwd()
print(json.dumps({"forward_us": t(fwd), "backward_us": t(bwd)}))
''' % (ROOT, os.path.abspath(lib_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", default=None)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    if args.build_only:
        print(build_variant(args.build_only))
        return
    from nnpops_amd import capi
    with tempfile.TemporaryDirectory(prefix="nnpops_proto_cf_") as wd:
        proto_lib = os.path.abspath(args.lib) if args.lib else build_variant(wd)
        rounds = [(measure(capi.LIB_PATH), measure(proto_lib)) for _ in range(3)]
    prod = {k: round(sorted(r[0][k] for r in rounds)[1], 2) for k in rounds[0][0]}
    proto = {k: round(sorted(r[1][k] for r in rounds)[1], 2) for k in rounds[0][1]}
    saved = {k: round(prod[k] - proto[k], 2) for k in prod}
    print(json.dumps({"workload": "BASELINE config 3: CFConv W = 128, G = 50, 5 A, 10 000-atom periodic box; one convolution forward / backward "
                                  "on a built neighbour list (filters kernel + gather kernel each), median of 3 runs of 100",
                      "product_us": prod, "filter_rows_never_leave_the_l2_us": proto, "upper_bound_of_the_gain_us": saved,
                      "note": "the prototype executes every instruction of the product kernels; only the fabric traffic of the filter rows "
                              "(133 MB written, 2 x 135 MB read per convolution) is gone.  What a spatial-tile variant would have to pay for "
                              "it -- boundary pairs computed twice on the matrix cores, both owners' accumulators in LDS -- is not charged."}))


if __name__ == "__main__":
    main()
