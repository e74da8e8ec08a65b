#!/usr/bin/env python3
"""Where does the time of the SchNet CFConv kernels go (BASELINE config 3: W = 128, G = 50, 5 A, 10 000 atoms)?  A PROBE build of
libnnpops_hip.so -- a patched copy of the kernel sources under a scratch directory, never the product sources -- in which parts of
cfconv_filters_h2 and cfconv_gather can be switched off at run time (`nnpops_debug_set_cf_probe(mask)`), timed with HIP events
around one forward / one backward convolution on a built list.  A probed run computes garbage; the numbers are durations only.

    python tools/probe_cfconv.py --build-only tools/_probe_cf        # here (hipcc, no GPU)
    python tools/probe_cfconv.py --lib tools/_probe_cf/libnnpops_hip.so   # on the GPU box; prints one JSON line
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PM = "cf_probe_mask"

PATCHES = [
    ("constexpr int kPairTile = 8;\n", "static __device__ int cf_probe_mask;\nconstexpr int kPairTile = 8;\n", 1),
    # 1: backward epilogue without the x / gout gathers and the contraction (pair_s = 0)
    ("                float sc = 0.f;\n#pragma unroll\n                for (int cb = 0; cb < NCB; cb++) {\n                    const size_t c = (size_t)cb * 16 + col;\n",
     f"                float sc = 0.f;\n                if (!({PM} & 1))\n#pragma unroll\n                for (int cb = 0; cb < NCB; cb++) {{\n                    const size_t c = (size_t)cb * 16 + col;\n", 1),
    # 2: filter rows not stored
    ("            if (p < pairs) {                                // uniform over the 16 lanes of a row\n                float* frow = filt + (size_t)p * W + col;\n",
     f"            if (p < pairs && !({PM} & 2)) {{\n                float* frow = filt + (size_t)p * W + col;\n", 1),
    ("            if (p < pairs && !(BWD && P.skip_filter_store)) {      // uniform over the 16 lanes of a row\n                float* frow = filt + (size_t)p * W + col;\n",
     f"            if (p < pairs && !(BWD && P.skip_filter_store) && !({PM} & 2)) {{\n                float* frow = filt + (size_t)p * W + col;\n", 1),
    # 4: no derivative pass of layer 1
    ("                l1_pass(std::true_type{}, dacc);\n", f"                if (!({PM} & 4)) l1_pass(std::true_type{{}}, dacc);\n", 1),
    # 8: no second pass of layer 2 (dY1)
    ("            h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, dacc, acc2);\n",
     f"            if (!({PM} & 8)) h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, dacc, acc2);\n", 1),
    # 16: no first pass of layer 2 (both directions)
    ("            h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);\n",
     f"            if (!({PM} & 16)) h2_layer<NCB, W, true, false, false>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);\n", 1),
    ("            h2_layer<NCB, W, false, true, true>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);\n",
     f"            if (!({PM} & 16)) h2_layer<NCB, W, false, true, true>(a_h, a_l, s_w2h, s_w2l, col, grp, col, acc, acc2);\n"
     f"            else for (int cb = 0; cb < NCB; cb++) {{ acc[cb] = zero4; acc2[cb] = zero4; }}\n", 1),
    # 32: no value pass of layer 1
    ("            l1_pass(std::false_type{}, acc);\n", f"            if (!({PM} & 32)) l1_pass(std::false_type{{}}, acc);\n            else for (int cb = 0; cb < NCB; cb++) acc[cb] = zero4;\n", 1),
    # 64: gather kernels return at once
    ("    const int k = __builtin_amdgcn_readfirstlane(xcd_contiguous_wave_id());      // (wave-uniform: the atom's id and count through the scalar cache)\n    if (k >= N) return;\n",
     f"    const int k = __builtin_amdgcn_readfirstlane(xcd_contiguous_wave_id());\n    if (k >= N || ({PM} & 64)) return;\n", 1),
    # 128: filter kernels return after their weights are in LDS (launch + prologue only)
    ("    const int col = lane & 15, grp = lane >> 4;\n    float b2v[NCB];                                         // (backward: re-read per tile, the registers are needed)\n",
     f"    if ({PM} & 128) return;\n    const int col = lane & 15, grp = lane >> 4;\n    float b2v[NCB];\n", 1),
    ("int nnpops_cfconv_set_stream(", "int nnpops_debug_set_cf_probe(int mask) {\n    return hipMemcpyToSymbol(HIP_SYMBOL(cf_probe_mask), &mask, sizeof(int)) == hipSuccess ? 0 : 1;\n}\n\nint nnpops_cfconv_set_stream(", 1),
]

PROBES = [
    (0, "nothing switched off"),
    (64, "gather kernel off (filters kernel alone)"),
    (64 | 128, "gather off, filters kernel: launch + weights into LDS only"),
    (64 | 1, "filters backward: no x / gout gathers, no contraction"),
    (64 | 2, "filters: filter rows not stored"),
    (64 | 4, "filters backward: no derivative pass of layer 1"),
    (64 | 8, "filters backward: no second pass of layer 2 (dY1)"),
    (64 | 16, "filters: no (first) pass of layer 2"),
    (64 | 32, "filters: no value pass of layer 1"),
    (64 | 4 | 8 | 16 | 32, "filters: no matrix products at all (Gaussians not formed either)"),
    (64 | 1 | 2 | 4 | 8 | 16 | 32, "filters: activation, splits, tile bookkeeping only"),
]


def build_variant(outdir):
    from nnpops_amd import build as hb
    os.makedirs(outdir, exist_ok=True)
    src_dir = os.path.join(outdir, "src", "nnpops_amd", "csrc")
    shutil.rmtree(os.path.join(outdir, "src"), ignore_errors=True)
    shutil.copytree(hb.CSRC, src_dir, ignore=shutil.ignore_patterns("_obj"))
    os.makedirs(os.path.join(outdir, "src", "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "nnpops_hip.h"), os.path.join(outdir, "src", "include", "nnpops_hip.h"))
    path = os.path.join(src_dir, "cfconv.hip")
    text = open(path).read()
    for anchor, replacement, count in PATCHES:
        assert text.count(anchor) == count, f"probe patch: anchor found {text.count(anchor)}x (want {count}): {anchor[:70]!r}"
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", default=None)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    if args.build_only:
        print(build_variant(args.build_only))
        return
    import ctypes as C
    import numpy as np
    import torch
    from nnpops_amd import capi, workloads
    capi.LIB_PATH = os.path.abspath(args.lib)
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    lib = capi.lib()
    setp = C.CDLL(capi.LIB_PATH).nnpops_debug_set_cf_probe
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

    def t(fn, reps=60):
        for _ in range(6):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return round(1e3 * a.elapsed_time(b) / reps, 2)

    fwd = lambda: cf.compute(nb, tpos, tx, tbox, out)
    bwd = lambda: cf.backprop(nb, tpos, tx, tg, tbox)
    fwd(); bwd()
    res = []
    for mask, what in PROBES:
        assert setp(mask) == 0
        torch.cuda.synchronize()
        res.append({"mask": mask, "off": what, "forward_us": t(fwd), "backward_us": t(bwd)})
    setp(0)
    print(json.dumps({"workload": "BASELINE config 3: one CFConv forward / backward on a built list, W = 128, G = 50, 5 A, 10 000 atoms "
                                  "(filters kernel + gather kernel); HIP events around 60 calls", "probes": res}))


if __name__ == "__main__":
    main()
