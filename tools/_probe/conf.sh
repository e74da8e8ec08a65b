for v in "X=1" "X=2"; do
  echo "== $v"; env $v python bench.py --workload conformers --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('shard8'); print(d['ms_per_step'], s['ms_per_block_step'], s['projected_scaling'], s['projected_scaling_with_synchronous_gather'])"
done
python -m pytest tests/test_ani_gpu.py tests/test_multi_gpu.py -x -q 2>&1 | tail -2
