#!/bin/bash
# Ablation of the ANI neighbour builder (timing experiments only; results are wrong with any bit set):
#   64 skip the radial AEV, 128 stop before the species sort / records / triple list, 256 skip the candidate scan
for dbg in 0 64 128 192 448; do
  NNPOPS_ANI_DEBUG=$dbg python - <<'PY'
import os, torch
from nnpops_amd import workloads
from nnpops_amd.capi import AniSymmetryFunctions
n = 10000
pos, species, box = workloads.random_box(n, density=0.1, seed=100, n_species=7)
rf, af = workloads.ani2x_functions()
sym = AniSymmetryFunctions(7, 5.1, 3.5, species, rf, af, periodic=True)
dev = torch.device("cuda:0")
tpos, tbox = torch.tensor(pos, device=dev), torch.tensor(box, device=dev)
radial = torch.empty((n, sym.radial_width), device=dev); angular = torch.empty((n, sym.angular_width), device=dev)
for _ in range(5): sym.compute(tpos, tbox, radial, angular, check=False)
sym.enable_timing(True)
for _ in range(30): sym.compute(tpos, tbox, radial, angular, check=False)
t = sym.get_timing()
print("dbg", os.environ["NNPOPS_ANI_DEBUG"], {k: round(1e3 * ms / max(c, 1), 2) for k, (ms, c) in t.items() if c})
PY
done
