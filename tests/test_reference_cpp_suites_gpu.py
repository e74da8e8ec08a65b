"""The reference's OWN C++ test suites, run against the MI355X implementation.

`make -C oracle ref_tests` (part of __graft_entry__.build() wherever /root/reference exists) compiles
src/ani/TestANISymmetryFunctions.h and src/schnet/TestCFConv.h -- read in place, never copied -- with the
reference-side classes of integration/ (HipANISymmetryFunctions, HipCFConvNeighbors, HipCFConv: subclasses of the
reference's abstract core API that forward to libnnpops_hip.so), exactly as the reference instantiates the same
headers with its Cuda* classes (src/ani/TestCudaANISymmetryFunctions.cu).  The binaries land in oracle/_ref/ and
travel to the GPU box; the reference itself does not.

What the suites assert (their words, their tolerances): TorchANI / SchNetPack golden values of the 18-atom water
cluster, non-periodic, periodic and triclinic, torchani and paper modes, ssp and tanh; finite-difference checks of
the position (and input) gradients.  A failed assertion throws, the binary exits non-zero.
"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("binary", ["test_hip_ani", "test_hip_cfconv"])
def test_reference_suite_passes(binary):
    path = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (needs /root/reference at build time: make -C oracle ref_tests)")
    lib = os.path.join(ROOT, "nnpops_amd", "libnnpops_hip.so")
    assert os.path.exists(lib), "the HIP library is missing: the product path has no fallback"
    done = subprocess.run([path], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, f"{binary} failed (exit {done.returncode})\n{done.stdout[-2000:]}\n{done.stderr[-2000:]}"
