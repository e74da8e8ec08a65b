for v in "NNPOPS_ANI_FWD_DYN=0" "NNPOPS_ANI_FWD_DYN=1"; do
  echo "== $v"; env $v python bench.py --workload conformers --steps 50 --warmup 5 --no-cpu-baseline --no-shard8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('kernels_us'))"
  env $v python bench.py --workload latency --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ligand_1hvj']['eager_us'])"
done
