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
UNITS = ["capi_common.hip", "ani.hip", "cfconv.hip", "neighbor_pairs.hip", "pairs_index.hip", "batched_nn.hip", "mlp_fused.hip", "pme.hip"]


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
    flags = " ".join(os.environ.get("NNPOPS_HIPCC_FLAGS", "").split())
    for src in _sources():                                    # (a development build with other flags is not the product binary)
        stamp = os.path.join(CSRC, "_obj", os.path.basename(src) + ".o.flags")
        if not os.path.exists(stamp) or open(stamp).read() != flags:
            return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + \
        [os.path.join(HERE, "..", "include", "nnpops_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps(path, seen=None):
    """The source and every header it includes with quotes, recursively (paths relative to the including file)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = os.environ.get("NNPOPS_HIPCC_FLAGS", "").split()     # (development: e.g. -DNNPOPS_ONLY_ANI2X_SHAPE, see ani.hip)
    objs = []
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # An object is reused when it is newer than its source and every header that source includes AND was compiled with the
        # same flags (recorded next to it).  capi_common.hip carries the source hash of the whole library: always recompiled.
        stamp = obj + ".flags"
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
                 and os.path.basename(src) != "capi_common.hip"
                 and all(os.path.getmtime(d) < os.path.getmtime(obj) for d in _deps(src)))
        if fresh:
            continue
        cmd = [hipcc, *flags, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f'-DNNPOPS_SOURCE_HASH="{source_hash()}"', "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, stamp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, stamp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        open(stamp, "w").write(" ".join(flags))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
