"""Structure of the repository and of the C ABI (CPU only, no compute calls)."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nnpops_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnpops_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from nnpops_amd import build, capi
    path = build.build()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(handle, s)]
    assert not missing, missing
    # the ctypes table covers exactly the header
    assert sorted(capi.SIGNATURES) == declared
    capi.lib()                                  # loads and binds without touching a GPU


def test_library_reports_errors_without_a_gpu():
    from nnpops_amd import capi
    L = capi.lib()
    assert L.nnpops_version().startswith(b"nnpops_hip")
    out = ctypes.c_void_p()
    code = L.nnpops_cfconv_neighbors_create(ctypes.byref(out), 0, 1.0, 0, 0)
    assert code < 0 and b"num_atoms" in L.nnpops_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under nnpops_amd/ (or the NNPOps facade) may import,
    load or execute it."""
    offenders = []
    for pkg in ("nnpops_amd", "NNPOps"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|oracle/_ref|liboracle|libnnpops_ref", text, flags=re.M):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    code = "import sys; import nnpops_amd, nnpops_amd.capi, nnpops_amd.workloads; assert 'oracle' not in sys.modules"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_no_reference_sources_in_tree():
    """Fixtures are data; the reference's sources stay under /root/reference."""
    bad = []
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "_obj", "_build", "_ref", "__pycache__")]
        for f in files:
            if f.endswith((".cu", ".cuh")) or f in ("CpuANISymmetryFunctions.cpp", "CpuCFConv.cpp", "getNeighborPairsCPU.cpp"):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad
