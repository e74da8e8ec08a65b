"""nnpops_amd -- MI355X (gfx950) native implementation of the NNPOps per-atom hot path.

The arithmetic lives in hand-written HIP kernels behind a C ABI (``include/nnpops_hip.h`` ->
``nnpops_amd/libnnpops_hip.so``); this package is the Python host side that mirrors the
reference's operator surface.  There is deliberately no CPU implementation here: importing the
package is cheap and GPU-free, but every compute entry point raises if the HIP library or a HIP
device is missing.
"""
__version__ = "0.1.0"

from . import capi  # noqa: F401  (does not load the .so until first use)
