for v in "X=1" "X=2"; do
  echo "== $v"; env $v python bench.py --workload latency --steps 3000 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], {k:v for k,v in d.items() if 'graph' in k or 'kernels' in k})"
done
