for v in "NNPOPS_ANI_BWD_CLASS_MIN=0" "NNPOPS_ANI_BWD_CLASS_MIN=512" "NNPOPS_ANI_BWD_CLASS_MIN=3000" "NNPOPS_ANI_BWD_CLASS_MIN=1000000"; do
  echo "== $v"; env $v python bench.py --workload conformers --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('shard8'); print(d['ms_per_step'], s['ms_per_block_step'], s['projected_scaling'], s['projected_scaling_with_synchronous_gather'])"
done
python tools/ab.py "" "LIB=tools/_ref/libnnpops_hip_r03.so" --rounds 9 2>&1 | tail -2
