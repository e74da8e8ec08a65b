"""Build and load ``libNNPOpsPyTorch.so`` -- the TORCH_LIBRARY surface of the reference
(torch.classes.NNPOps*.Holder, torch.ops.NNPOps*.operation, torch.ops.neighbors.getNeighborPairs,
torch.ops.NNPOpsBatchedNN.BatchedLinear) implemented on top of the C ABI.

    python -m nnpops_amd.torch_binding          # build (incremental)

The library is built in-tree, next to libnnpops_hip.so (which it links with an $ORIGIN rpath), with
the same name the reference uses (src/pytorch/__init__.py:14) and is loaded with
``torch.ops.load_library`` exactly as the reference does.  It is plain C++ (no device code), so g++
is enough; hipcc is only needed for libnnpops_hip.so.
"""
import os
import subprocess
import sys

import torch
from torch.utils import cpp_extension

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "torch_binding.cpp")
LIB = os.path.join(HERE, "libNNPOpsPyTorch.so")
_loaded = False


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in (SRC, os.path.join(HERE, "..", "include", "nnpops_hip.h")):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def built_hash():
    if not os.path.exists(LIB):
        return None
    import re
    m = re.search(rb"nnpops_torch_binding src:([0-9a-f]{16})", open(LIB, "rb").read())
    return m.group(1).decode() if m else None


def _stale():
    return built_hash() != source_hash()


def build(force=False, verbose=False):
    from . import build as hip_build
    hip_build.build()
    if not force and not _stale():
        return LIB
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f'-DNNPOPS_BINDING_HASH="{source_hash()}"']
    for inc in cpp_extension.include_paths(False) + ["/opt/rocm/include"]:
        cmd += ["-isystem", inc]
    cmd += [SRC, "-o", LIB, f"-L{HERE}", "-lnnpops_hip", f"-L{torch_lib}", "-ltorch", "-ltorch_cpu", "-lc10",
            "-ltorch_hip", "-lc10_hip", "-L/opt/rocm/lib", "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{torch_lib}", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def load():
    """torch.ops.load_library(libNNPOpsPyTorch.so); raises if it has not been built."""
    global _loaded
    if _loaded:
        return
    if not os.path.exists(LIB):
        raise ImportError(f"{LIB} is missing: build it with `python -m nnpops_amd.torch_binding`")
    if built_hash() != source_hash():
        raise ImportError(f"{LIB} is stale (built from other sources than the ones next to it): rebuild with "
                          "`python -m nnpops_amd.torch_binding`")
    torch.ops.load_library(LIB)
    _loaded = True


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
