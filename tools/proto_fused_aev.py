#!/usr/bin/env python3
"""Upper bound on what fusion (A) of SURVEY s8f rank 1 -- AEV tiles handed to the first network layer through LDS, never
through HBM -- could gain on BASELINE config 2 ("keep the prototype's numbers in profiles/").

Builds a second copy of libnnpops_hip.so under /tmp from a PATCHED copy of the kernel sources (PATCHES below; the product
sources carry no prototype code): the angular forward assembles its
rows in LDS as always but does not store them, and the fused networks take their layer-0 operand from registers instead of
reading the [N, 1008] array.  Both kernels then do ALL of their arithmetic, LDS traffic and weight streaming and NONE of the
AEV round trip; the difference to the product library is everything fusion (A) could remove (it would still have to pay for
species-sorted tiles, eight-fold recomputation or a producer/consumer hand-over -- none of which is charged here).

    python tools/proto_fused_aev.py          # on the GPU box (needs hipcc); prints one JSON line
"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# The prototype is a PATCH applied to a temporary copy of the kernel sources (round 4: the product kernels carry no
# prototype #ifdef any more): (anchor in the product source, replacement).  An anchor that no longer matches stops the tool.
PATCHES = {
    "ani_angular_mfma.h": [(
        "                store_row16(out + 4 * q, mfma_f4{v.x, v.y, v.z, v.w}, (vec_ok >> 1) & 3);\n",
        # the AEV rows stay in LDS (one token store per atom, never taken, keeps the work alive)
        "                if (q == 0 && v.x == 12345.678f) store_row16(out, mfma_f4{v.x, v.y, v.z, v.w}, 0);\n")],
    "mlp_fused.hip": [(
        "            xraw = *reinterpret_cast<const float4*>(xsrc + 16 * xgroup[in ? 2 * s + (piece >> 2) : 0]);\n",
        # what the kernel would cost if the AEV never came from memory: the layer-0 operand from registers
        "            xraw = make_float4(0.25f, 0.5f, 0.75f, 1.0f);\n")],
}


def build_variant(workdir):
    import shutil
    from nnpops_amd import build as hb
    src_dir = os.path.join(workdir, "nnpops_amd", "csrc")      # (host_common.h includes ../../include/nnpops_hip.h)
    shutil.rmtree(os.path.join(workdir, "nnpops_amd"), ignore_errors=True)
    shutil.copytree(hb.CSRC, src_dir, ignore=shutil.ignore_patterns("_obj"))
    os.makedirs(os.path.join(workdir, "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "nnpops_hip.h"), os.path.join(workdir, "include", "nnpops_hip.h"))
    for name, edits in PATCHES.items():
        path = os.path.join(src_dir, name)
        text = open(path).read()
        for anchor, replacement in edits:
            assert text.count(anchor) == 1, f"prototype patch: anchor not found exactly once in {name}"
            text = text.replace(anchor, replacement)
        open(path, "w").write(text)
    objs = []
    procs = []
    for unit in hb.UNITS:
        src = os.path.join(src_dir, unit)
        obj = os.path.join(workdir, unit + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                       f'-DNNPOPS_SOURCE_HASH="{hb.source_hash()}"', "-c", src, "-o", obj]))
    for p in procs:
        assert p.wait() == 0
    lib = os.path.join(workdir, "libnnpops_hip.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    shutil.rmtree(os.path.join(workdir, "nnpops_amd"), ignore_errors=True)       # (the patched copy of the sources is not kept)
    shutil.rmtree(os.path.join(workdir, "include"), ignore_errors=True)
    return lib


def measure(lib_path, frame="water"):
    """AEV forward (kernel event times) and the fused networks (HIP events) of config 2 with the given library."""
    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from nnpops_amd import capi, workloads
capi.LIB_PATH = %r
from nnpops_amd.capi import AniSymmetryFunctions, FusedMLP
sys.path.insert(0, %r)
frame = %r
if frame == "water":
    pos, species, box = workloads.water_box(667, seed=1)          # BASELINE config 2: 2 of the 7 species, 128 of 1008 columns live
elif frame == "seven100k":
    pos, species, box = workloads.random_box(100000, density=0.1, seed=1, n_species=7)  # the [N, 1008] array is 403 MB: more than the 256 MiB Infinity Cache
else:
    pos, species, box = workloads.random_box(2000, density=0.1, seed=1, n_species=7)    # all 7 species: every AEV column live
rf, af = workloads.ani2x_functions()
dev = torch.device("cuda:0")
sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
aev = torch.zeros((len(species), 1008), device=dev)
radial, angular = aev[:, :112], aev[:, 112:]
import ctypes as C
def aev_forward():
    capi._check(capi.lib().nnpops_ani_set_stream(sym._h, capi._stream_ptr(dev)))
    capi._check(capi.lib().nnpops_ani_compute_strided(sym._h, capi._ptr(tpos), capi._ptr(tbox), capi._ptr(aev), 1008, C.c_void_p(aev.data_ptr() + 448), 1008))
sym.compute(tpos, tbox)          # calibrates capacities
def t(fn, reps=300 if len(species) <= 20000 else 30):
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
g = torch.Generator().manual_seed(0)
def nets(w, s):
    r = lambda *sh, fan: (torch.randn(sh, generator=g) / np.sqrt(fan)).float().cuda()
    h1, h2, h3 = w
    return dict(w0=r(8, h1, 1008, fan=1008), b0=r(8, h1, fan=100), w2=r(8, h2, h1, fan=h1), b2=r(8, h2, fan=100), w4=r(8, h3, h2, fan=h2),
                b4=r(8, h3, fan=100), w6=r(8, h3, fan=h3), b6=r(8, fan=100))
sp = torch.tensor(species)
kinds = []
widths = {0: (256, 192, 160), 1: (224, 192, 160), 2: (192, 160, 128), 3: (192, 160, 128), 4: (160, 128, 96), 5: (160, 128, 96), 6: (160, 128, 96)}
for s in sorted(set(int(x) for x in species)):
    w = widths[s]
    kd = nets(w, s); kd["atoms"] = torch.nonzero(sp == s).flatten().to(torch.int32).cuda(); kinds.append(kd)
mlp = FusedMLP(kinds, 1008)
aev_forward()
out = {"aev_forward_us": t(aev_forward), "networks_forward_us": t(lambda: mlp.forward(aev, with_gradient=True))}
# the rest of the step, so that the bound can be stated as a share of it: the networks' input gradient and the AEV backward
gaev = torch.zeros_like(aev)
mlp.forward(aev, with_gradient=True)
out["networks_input_grad_us"] = t(lambda: mlp.input_grad(aev, out=gaev))
g_r, g_a = torch.randn((len(species), 112), device=dev), torch.randn((len(species), 896), device=dev)
grad = torch.empty((len(species), 3), device=dev)
sym.compute(tpos, tbox)
out["aev_backward_us"] = t(lambda: sym.backprop(g_r, g_a, grad))
print(json.dumps(out))
''' % (ROOT, lib_path, ROOT, frame)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", default=None, help="build the prototype library into this directory (hipcc, no GPU needed) and stop")
    ap.add_argument("--lib", default=None, help="a prototype library built earlier with --build-only")
    ap.add_argument("--frames", default="water,seven", help="comma-separated: water, seven (2 000 atoms), seven100k (100 000 atoms)")
    args = ap.parse_args()
    if args.build_only:
        os.makedirs(args.build_only, exist_ok=True)
        print(build_variant(args.build_only))
        return
    from nnpops_amd import capi
    out = {}
    with tempfile.TemporaryDirectory(prefix="nnpops_proto_") as wd:
        proto_lib = os.path.abspath(args.lib) if args.lib else build_variant(wd)
        labels = {"water": "BASELINE config 2: 2001-atom water box (2 species, 128 live AEV columns)",
                  "seven": "2000 atoms, 7 species uniform (all 1008 AEV columns live: the networks read 8x what they read for water)",
                  "seven100k": "100 000 atoms, 7 species uniform (all 1008 columns live; the AEV array is 403 MB, beyond the 256 MiB Infinity Cache: "
                               "written once by the angular forward, read by the networks from HBM)"}
        for frame in args.frames.split(","):
            label = labels[frame]
            product = measure(capi.LIB_PATH, frame)
            proto = measure(proto_lib, frame)
            fused = ("aev_forward_us", "networks_forward_us")          # the two launches fusion (A) would merge
            saved = {k: round(product[k] - proto[k], 2) for k in fused}
            step = sum(product.values())
            out[frame] = {"workload": label + "; AEV forward (fused build + forward) and the fused networks' forward launch; the networks' input "
                                              "gradient and the AEV backward complete the step",
                          "product_us": {k: round(v, 2) for k, v in product.items()},
                          "aev_never_in_memory_us": {k: round(proto[k], 2) for k in fused},
                          "upper_bound_of_fusion_A_us": saved, "total_upper_bound_us": round(sum(saved.values()), 2),
                          "step_us": round(step, 2), "upper_bound_share_of_step": round(sum(saved.values()) / step, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
