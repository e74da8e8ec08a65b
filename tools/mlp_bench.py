"""Times the fused atomic networks alone on the shapes of BASELINE config 2 (and larger frames):
    python tools/mlp_bench.py [n_waters ...]
forward(+small-layer backward) and input-gradient launches, HIP events over 200 launches each."""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nnpops_amd.capi import FusedMLP


def nets(widths, members, seed):
    g = torch.Generator().manual_seed(seed)
    h1, h2, h3 = widths
    r = lambda *s, fan: (torch.randn(s, generator=g) / np.sqrt(fan)).float().cuda()
    return dict(w0=r(members, h1, 1008, fan=1008), b0=r(members, h1, fan=100), w2=r(members, h2, h1, fan=h1), b2=r(members, h2, fan=100),
                w4=r(members, h3, h2, fan=h2), b4=r(members, h3, fan=100), w6=r(members, h3, fan=h3), b6=r(members, fan=100))


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


# --live: the networks over the AEV column blocks a water box can fill (species H = 0 and O = 3 of 7: 8 of the 63 blocks), with
# the input gradient formed inside the forward launch -- what OptimizedTorchANI runs
LIVE = None
if "--live" in sys.argv:
    sys.argv.remove("--live")
    S, nR, nA = 7, 16, 32
    cols = []
    for s_ in (0, 3):
        cols += list(range(s_ * nR, (s_ + 1) * nR))
    for a_, b_ in ((0, 0), (0, 3), (3, 3)):
        bucket = a_ * S - a_ * (a_ - 1) // 2 + (b_ - a_)
        cols += list(range(S * nR + bucket * nA, S * nR + (bucket + 1) * nA))
    LIVE = sorted({c // 16 for c in cols})

for waters in [int(a) for a in sys.argv[1:]] or [667, 3334]:
    n = 3 * waters
    species = torch.tensor([3, 0, 0] * waters)
    kinds = []
    for s, w in ((0, (256, 192, 160)), (3, (192, 160, 128))):
        kd = nets(w, 8, s)
        kd["atoms"] = torch.nonzero(species == s).flatten().to(torch.int32).cuda()
        kinds.append(kd)
    mlp = FusedMLP(kinds, 1008, live_groups=LIVE)
    x = torch.rand((n, 1008), device="cuda")
    dx = torch.empty_like(x)
    t_e = timeit(lambda: mlp.forward(x, with_gradient=False))
    t_f = timeit(lambda: mlp.forward(x, with_gradient=True))
    t_g = timeit(lambda: mlp.input_grad(x, out=dx))
    macs = 8 * (waters * (1008 * 192 + 192 * 160 + 160 * 128 + 128) + 2 * waters * (1008 * 256 + 256 * 192 + 192 * 160 + 160))
    print(f"{n} atoms: energy-only {t_e:.1f} us | forward + small-layer backward {t_f:.1f} us | input gradient {t_g:.1f} us | "
          f"fwd+bwd {t_f + t_g:.1f} us = {4 * macs / (t_f + t_g) / 1e6:.1f} TFLOP/s algorithmic")
