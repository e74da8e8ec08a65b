"""Errors of the CFConv kernels against the oracle: the register-fed split-fp16 kernels (round 6), the plane kernels (rounds 3-5) and the fp32
matrix kernel, relative to the largest entry of every output.   python tools/cfconv_split_error.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_cfconv_gpu as T
from nnpops_amd import workloads
pos, _, box = workloads.random_box(1400, seed=91)
for (W, G, act, sigma) in [(96, 33, "tanh", 0.1), (96, 33, "ssp", 0.1), (128, 50, "tanh", 0.1), (64, 20, "tanh", 0.1), (128, 50, "tanh", 0.5), (128, 50, "ssp", 0.5)]:
    res = {}
    for name, env in (("new", {}), ("old", {"NNPOPS_CFCONV_FWD32": "0", "NNPOPS_CFCONV_BWD1": "0"}), ("l2only", {"NNPOPS_CFCONV_SPLIT": "1"}), ("fp32", {"NNPOPS_CFCONV_SPLIT": "0"})):
        for k in ("NNPOPS_CFCONV_FWD32", "NNPOPS_CFCONV_BWD1", "NNPOPS_CFCONV_SPLIT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        keep = {}
        T._case(pos, box, W, G, 5.0, sigma, act, seed=31, keep=keep)
        res[name] = keep
    for key in ("y", "xg", "pg"):
        ref = res["new"][key + "_ref"].astype(np.float64)
        sc = np.abs(ref).max()
        print(W, G, act, sigma, key, " ".join(f"{n}:{np.abs(res[n][key] - ref).max() / sc:.2e}" for n in res),
              f"new-old:{np.abs(res['new'][key] - res['old'][key]).max() / sc:.2e}")
