#!/bin/bash
# Kernel ablation timings (results are WRONG under ablation; only the per-kernel times mean anything).
for d in 0 32 16 2 3 4 7; do
  NNPOPS_ANI_DEBUG=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$d', {k: d['kernels_us'][k] for k in ('angular_forward','angular_backward')})"
done
