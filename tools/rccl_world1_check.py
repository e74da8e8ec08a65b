#!/usr/bin/env python3
"""RCCL sanity on a one-device box: a one-rank "nccl" process group running the collectives bench.py's N > 1 path uses
(asynchronous all_gather_into_tensor on alternating buffers, all_reduce MAX, barrier).  No scaling information -- it only shows
that the RCCL build of this image initialises and completes these calls with the arguments bench.py passes."""
import json
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 10000
grads = [torch.randn((n, 3), device=dev) for _ in range(2)]
gathered = [torch.empty((1 * n, 3), device=dev) for _ in range(2)]
works = [None, None]
for step in range(20):
    k = step & 1
    if works[k] is not None:
        works[k].wait()
    grads[k].add_(1.0)
    works[k] = dist.all_gather_into_tensor(gathered[k], grads[k], async_op=True)
for w in works:
    w.wait()
torch.cuda.synchronize()
ok = bool(torch.equal(gathered[0], grads[0]) and torch.equal(gathered[1], grads[1]))
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print(json.dumps({"rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "backend": dist.get_backend(),
                  "world_size": dist.get_world_size(), "gather_matches": ok, "all_reduce_max": float(t.item())}))
dist.destroy_process_group()
