"""One block of shard_molecules(sizes, 8) of the 1 024-conformer batch alone on the device: step time eager / as a HIP graph,
per-kernel event times; `NNPOPS_ANI_FUSE=1 python tools/shard_probe.py` forces the fused build + forward."""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from nnpops_amd.parallel import shard_molecules

sizes = bench.conformer_sizes()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
from nnpops_amd import workloads
from nnpops_amd.parallel import molecule_work
work = [molecule_work(workloads.conformer(sizes[m], seed=1000 + m)[0], 3.5) for m in range(len(sizes))]
lo, hi = shard_molecules(sizes, world, weights=work)[which]
sh = bench.ConformerShard(sizes, lo, hi, 0)
buf = torch.empty((sh.n, 3), device="cuda")
step = lambda: sh.step(buf)
t_eager = bench._time_steps(step, 300, 30, repeats=1)
sh.sym.enable_timing(True)
for _ in range(50):
    step()
kt = {k: 1e3 * ms / max(c, 1) for k, (ms, c) in sh.sym.get_timing().items()}
sh.sym.enable_timing(False)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
t_graph = bench._time_steps(g.replay, 300, 30, repeats=1)
print(sh.sym.describe())
print(f"block {which}: {hi - lo} conformers, {sh.n} atoms: eager {1e3 * t_eager:.4f} ms, graph {1e3 * t_graph:.4f} ms; kernels (us, event brackets included): "
      + ", ".join(f"{k} {v:.1f}" for k, v in kt.items()), "max row / angular:", sh.sym.neighbor_stats())
