#!/usr/bin/env python3
"""Where does the time of the four per-atom ANI kernels go?  A PROBE build of libnnpops_hip.so -- a patched copy of the
kernel sources under a scratch directory, never the product sources -- in which parts of a kernel can be switched off at run
time (a device word, `nnpops_debug_set_probe(mask)`), timed kernel by kernel with the handle's own event brackets.

A switched-off part leaves the arrays of the last VALID evaluation in place (same positions every step), so the kernels
downstream keep running on valid data; the numbers are durations only, the results of a probed step mean nothing.

    python tools/probe_ani.py --build-only tools/_probe      # here (hipcc, no GPU): builds tools/_probe/libnnpops_hip.so
    python tools/probe_ani.py --lib tools/_probe/libnnpops_hip.so [--atoms 10000]     # on the GPU box; prints one JSON line
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PM = "nnpops_probe_mask"
# file -> [(anchor, replacement, expected count)]
PATCHES = {
    "ani_kernels.h": [
        ("constexpr int kMaxRadialFns = 64;\n", "static __device__ int nnpops_probe_mask;\nconstexpr int kMaxRadialFns = 64;\n", 1),
        ("    for (int t = lane; t < E; t += 64) {\n        int p, q;\n",
         f"    if (!({PM} & 1))\n    for (int t = lane; t < E; t += 64) {{\n        int p, q;\n", 1),
        ("    flush_row(row, stage, cap, na, nro);\n    radial_forward_from_lds(P, stage, cap, n, nro_c, rscratch, radial + (size_t)i * ld_radial);\n    finalize_angular(",
         f"    if (!({PM} & 8)) flush_row(row, stage, cap, na, nro);\n    if (!({PM} & 2)) radial_forward_from_lds(P, stage, cap, n, nro_c, rscratch, radial + (size_t)i * ld_radial);\n"
         f"    if (!({PM} & 4)) finalize_angular(", 2),
        ("        store_wt(recA + rank, a);\n        store_wt(recB + rank, b2);\n",
         f"        if (!({PM} & 32)) {{ store_wt(recA + rank, a);\n        store_wt(recB + rank, b2); }}\n", 1),
        # bit 4 (16): no candidate scan at all -- the counts of the last valid build are put back; the LDS row is then garbage, so
        # this bit is only ever set together with 2 | 4 | 8 (nothing that reads the row runs)
        ("    for (int base = 0; base < st.total; base += 64 * GROUP) {\n",
         f"    if ({PM} & 16) {{ na = cnt_a[i]; nro = cnt_ro[i]; }} else\n    for (int base = 0; base < st.total; base += 64 * GROUP) {{\n", 1),
    ],
    "ani_angular_mfma.h": [
        ("            const int steps = wave_max_nonneg(cmax);             // wave-uniform trip count\n",
         f"            const int steps = ({PM} & 64) ? 0 : wave_max_nonneg(cmax);\n", 1),
        ("                if (t < c1) {\n                    const int p = word & 0xff, q = (word >> 8) & 0xff;\n                    const float4 A = recA[p], B = recA[q];\n",
         f"                if (t < c1 && !({PM} & 128)) {{\n                    const int p = word & 0xff, q = (word >> 8) & 0xff;\n                    const float4 A = recA[p], B = recA[q];\n", 1),
        ("                store_row16(out + 4 * q, mfma_f4{v.x, v.y, v.z, v.w}, (vec_ok >> 1) & 3);\n",
         f"                if (!({PM} & 256)) store_row16(out + 4 * q, mfma_f4{{v.x, v.y, v.z, v.w}}, (vec_ok >> 1) & 3);\n", 1),
    ],
    "ani_angular_mfma.h#2": [
        ("        F.atom(i, n, [&](int t) { return tri[t]; },", f"        if (!({PM} & 8192)) F.atom(i, n, [&](int t) {{ return tri[t]; }},", 1),
        ("        }\n\n        for (int c0 = 0; c0 < T; c0 += CH) {\n", f"        }}\n        if ({PM} & 16384) return;\n\n        for (int c0 = 0; c0 < T; c0 += CH) {{\n", 1),
        ("        if constexpr (DYN) {\n            // the quads of a bucket are consecutive and inside one wave",
         f"        if ({PM} & 32768) return;\n        if constexpr (DYN) {{\n            // the quads of a bucket are consecutive and inside one wave", 1),
    ],
    "ani_angular_bwd.h": [
        ("            if (t < T) {\n                const int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;",
         f"            if (t < T && !({PM} & 512)) {{\n                const int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;", 1),
        ("        if (role == 0) {\n            float cx = 0.f, cy = 0.f, cz = 0.f;\n",
         f"        if (role == 0 && !({PM} & 1024)) {{\n            float cx = 0.f, cy = 0.f, cz = 0.f;\n", 1),
    ],
    "ani_radial_bwd.h": [
        ("        const bool look = !RECV && base == 0 && na > 0;\n", f"        const bool look = !RECV && base == 0 && na > 0 && !({PM} & 2048);\n", 1),
        ("        for (int c = 0; c < NR4; c++) gj[c] = grow[c];\n",
         f"        for (int c = 0; c < NR4; c++) gj[c] = ({PM} & 4096) ? make_float4(0.f, 0.f, 0.f, 0.f) : grow[c];\n", 1),
    ],
    "ani.hip": [
        ("int nnpops_ani_set_stream(nnpops_ani_t h, void* stream) {\n",
         "int nnpops_debug_set_probe(int mask) {\n    return hipMemcpyToSymbol(HIP_SYMBOL(nnpops::nnpops_probe_mask), &mask, sizeof(int)) == hipSuccess ? 0 : 1;\n}\n\n"
         "int nnpops_ani_set_stream(nnpops_ani_t h, void* stream) {\n", 1),
    ],
    # bits 65536 / 131072 (round 5, VERDICT r04 "next" #1): what the two angular kernels would have to do per triple WITHOUT the
    # builder's triple list -- the pair from the folded enumeration, its species pair from the records, its place in the bucket-major
    # staging order from per-species tables in LDS (forward), the bucket alone (backward) -- executed IN ADDITION to the product path
    # and folded into an index through a factor that is zero at run time (bit 30 of the mask, never set): results unchanged, the
    # instructions and LDS look-ups are all there.  With bit 1 (builder without its triple loop) this prices the whole idea.
    "ani_angular_mfma.h#3": [
        ("                    const int p = word & 0xff, q = (word >> 8) & 0xff;\n                    const float4 A = recA[p], B = recA[q];\n",
         f"                    int p = word & 0xff, q = (word >> 8) & 0xff;\n"
         f"                    if ({PM} & 65536) {{\n"
         f"                        int p2, q2;\n"
         f"                        const bool ok2 = decode_pair_folded(t, n, __builtin_amdgcn_rcpf((float)max(n - 1, 1)), p2, q2);\n"
         f"                        const int A2 = __float_as_int(recB[min(p2, capA - 1)].w) >> kTagShift, B2 = __float_as_int(recB[min(q2, capA - 1)].w) >> kTagShift;\n"
         f"                        const int S2 = 7, bucket2 = __mul24(A2, S2) - __mul24(A2, A2 - 1) / 2 + (B2 - A2);\n"
         f"                        const int ga = qtab[A2 & 31], gb2 = qtab[32 + (B2 & 31)];                  // {{first slot | count << 8}} of the two species: two LDS look-ups\n"
         f"                        const int ia = p2 - (ga & 0xff), ib = q2 - (gb2 & 0xff), cnt = gb2 >> 8;\n"
         f"                        const int local = A2 == B2 ? __mul24(ia, 2 * cnt - ia - 1) / 2 + (ib - ia - 1) : __mul24(ia, cnt) + ib;\n"
         f"                        const int pos2 = qtab[bucket2 & 63] + local + (ok2 ? 0 : 1);           // first triple of the bucket: a third look-up\n"
         f"                        p += pos2 * (({PM} >> 30) & 1);\n"
         f"                    }}\n"
         f"                    const float4 A = recA[p], B = recA[q];\n", 1),
    ],
    "ani_angular_bwd.h#2": [
        ("                const int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;\n",
         f"                int p = word & 0xff, q = (word >> 8) & 0xff, bucket = word >> 16;\n"
         f"                if ({PM} & 131072) {{\n"
         f"                    int p2, q2;\n"
         f"                    const bool ok2 = decode_pair_folded(t, n, __builtin_amdgcn_rcpf((float)max(n - 1, 1)), p2, q2);\n"
         f"                    const int A2 = __float_as_int(recB[min(p2, tile - 1)].w) >> kTagShift, B2 = __float_as_int(recB[min(q2, tile - 1)].w) >> kTagShift;\n"
         f"                    const int bucket2 = __mul24(A2, 7) - __mul24(A2, A2 - 1) / 2 + (B2 - A2) + (ok2 ? 0 : 1);\n"
         f"                    bucket += (bucket2 + p2 + q2) * (({PM} >> 30) & 1);\n"
         f"                }}\n", 1),
    ],
}

PROBES = [
    (0, "nothing switched off"),
    (1, "builder: no triple list (decode + place + 153 scattered 4-byte stores per atom)"),
    (32, "builder: records computed, not stored (recA / recB)"),
    (2, "builder: no radial AEV"),
    (4, "builder: no finalize_angular at all (sort, records, ids, bucket offsets, triple list)"),
    (8, "builder: neighbour row not written"),
    (2 | 4 | 8 | 16, "builder: prologue + stencil ranges only"),
    (64, "angular forward: no phase 2 (MFMA step loop)"),
    (128, "angular forward: no phase 1 (triple arithmetic + staging)"),
    (64 | 128, "angular forward: neither phase (loads, barriers, row assembly, stores)"),
    (64 | 128 | 256, "angular forward: neither phase, rows not stored"),
    (8192, "angular forward: workgroup prologue only (constants, zero record), no atom"),
    (16384, "angular forward: prologue + the atom's loads + first barrier"),
    (64 | 128 | 32768, "angular forward: neither phase, no epilogue (no row assembly, no stores)"),
    (32768, "angular forward: both phases, no epilogue"),
    (512, "angular backward: no triple arithmetic"),
    (1024, "angular backward: no row sums / leg-force stores"),
    (512 | 1024, "angular backward: loads and barriers only"),
    (2048, "radial backward: no reverse lookup / leg-force gather"),
    (4096, "radial backward: neighbours' gradient rows not gathered"),
    (2048 | 4096, "radial backward: own row only"),
    (65536, "angular forward: + per triple what phase 1 would do WITHOUT a triple list (folded decode, species pair, place from LDS tables)"),
    (131072, "angular backward: + per triple what it would do without a triple list (folded decode, bucket from the species pair)"),
    (1 | 65536 | 131072, "NO TRIPLE LIST, priced: builder without its triple loop, both angular kernels with the decode on top of their work"),
]


def build(outdir):
    from nnpops_amd import build as hb
    os.makedirs(outdir, exist_ok=True)
    src_dir = os.path.join(outdir, "src", "nnpops_amd", "csrc")          # (host_common.h includes ../../include/nnpops_hip.h)
    shutil.rmtree(os.path.join(outdir, "src"), ignore_errors=True)
    shutil.copytree(hb.CSRC, src_dir, ignore=shutil.ignore_patterns("_obj"))
    os.makedirs(os.path.join(outdir, "src", "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "nnpops_hip.h"), os.path.join(outdir, "src", "include", "nnpops_hip.h"))
    for name, edits in PATCHES.items():
        path = os.path.join(src_dir, name.split("#")[0])
        text = open(path).read()
        for anchor, replacement, count in edits:
            assert text.count(anchor) == count, f"probe patch: anchor found {text.count(anchor)}x (want {count}) in {name}: {anchor[:60]!r}"
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


def measure(lib_path, atoms, steps, rounds):
    import ctypes as C
    import numpy as np
    import torch
    from nnpops_amd import capi, workloads
    capi.LIB_PATH = os.path.abspath(lib_path)
    from nnpops_amd.capi import AniSymmetryFunctions
    dev = torch.device("cuda:0")
    pos, species, box = workloads.random_box(atoms, density=0.1, seed=100, n_species=7)
    rf, af = workloads.ani2x_functions()
    sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
    set_probe = capi.lib().nnpops_debug_set_probe             # (only the probe build exports it)
    set_probe.argtypes, set_probe.restype = [C.c_int], C.c_int
    tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
    n = len(species)
    radial = torch.empty((n, sym.radial_width), device=dev)
    angular = torch.empty((n, sym.angular_width), device=dev)
    g_r, g_a = torch.randn_like(radial), torch.randn_like(angular)
    grad = torch.empty((n, 3), device=dev)
    sym.compute(tpos, tbox, radial, angular, check=True)
    sym.backprop(g_r, g_a, grad)
    torch.cuda.synchronize()
    samples = {m: [] for m, _ in PROBES}
    for _ in range(rounds):                                   # interleaved: box and clock drift hit every probe alike
        for mask, _ in PROBES:
            assert set_probe(mask) == 0
            sym.enable_timing(True)
            for _ in range(steps):
                sym.compute(tpos, tbox, radial, angular, check=False)
                sym.backprop(g_r, g_a, grad)
            t = sym.get_timing()
            sym.enable_timing(False)
            samples[mask].append({k: 1e3 * ms / max(c, 1) for k, (ms, c) in t.items()})
    set_probe(0)
    ovh = 1e6 * sym.timing_overhead()
    out = {"atoms": n, "event_pair_overhead_us": round(ovh, 2), "note": "event-bracket medians in us, raw (bracket overhead not subtracted)",
           "probes": []}
    for mask, what in PROBES:
        med = {k: round(float(np.median([s[k] for s in samples[mask]])), 2) for k in samples[mask][0] if samples[mask][0][k] > 0}
        out["probes"].append({"mask": mask, "off": what, "us": med})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", default=None)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--atoms", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    if args.build_only:
        print(build(args.build_only))
        return
    lib = args.lib or build(os.path.join(ROOT, "tools", "_probe"))
    print(json.dumps(measure(lib, args.atoms, args.steps, args.rounds)))


if __name__ == "__main__":
    main()
