for v in "X=1" "NNPOPS_ANI_CELL_ATOMS=2200"; do
echo "== $v"; env $v python bench.py --workload torchani --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k.startswith('ms_')})"
done
