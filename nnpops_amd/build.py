"""Build libnnpops_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m nnpops_amd.build            # incremental: rebuilds only when a source is newer
    python -m nnpops_amd.build --force

hipcc cross-compiles without a GPU present; the resulting .so sits next to this file so that it
travels with the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnnpops_hip.so")
ARCH = "gfx950"

# translation units of the C-ABI library (the torch binding is built separately, see torch_binding.py)
UNITS = ["capi_common.hip", "ani.hip", "cfconv.hip", "neighbor_pairs.hip", "batched_nn.hip", "mlp_fused.hip", "pme.hip"]


def _sources():
    return [os.path.join(CSRC, u) for u in UNITS if os.path.exists(os.path.join(CSRC, u))]


def source_hash():
    """sha256 over every source of the library (file names and contents), first 16 hex digits: embedded in
    nnpops_version() at build time and checked by capi.lib() at load time, so a binary older than its sources -- the
    .so files are git-ignored but travel to the GPU box -- cannot be used silently."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))
    files.append(os.path.join(HERE, "..", "include", "nnpops_hip.h"))
    for f in files:
        if os.path.basename(f) == "torch_binding.cpp":
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def built_hash():
    """The hash recorded inside the existing binary, or None."""
    if not os.path.exists(LIB):
        return None
    import re
    m = re.search(rb"nnpops_hip [0-9.]+ gfx950 src:([0-9a-f]{16})", open(LIB, "rb").read())
    return m.group(1).decode() if m else None


def _stale():
    if not os.path.exists(LIB) or built_hash() != source_hash():
        return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + \
        [os.path.join(HERE, "..", "include", "nnpops_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f'-DNNPOPS_SOURCE_HASH="{source_hash()}"', "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
