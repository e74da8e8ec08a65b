# getNeighborPairs at 100 000 atoms: parity tests, the bench phases, per-kernel durations (run on the GPU box)
timeout 600 python -m pytest tests/test_neighbor_pairs_gpu.py tests/test_full_size_gpu.py tests/test_pme_gpu.py -x -q -k "pairs or neighbor or pme" 2>&1 | tail -2
timeout 200 python bench.py --workload neighbors --no-cpu-baseline --no-pmc 2>&1 | tail -1 > gpurun_out/nb.json
python -c "
import json; d=json.load(open('gpurun_out/nb.json')); print(d['ms_per_step'], d.get('phases_ms'))"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p
rocprofv3 --kernel-trace --stats -d /tmp/p -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload neighbors --no-cpu-baseline --no-pmc --steps 10 --warmup 3 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:20]:
    if any(k in r["Name"] for k in ("pairs_cells","cells","grid_setup","fill_tail","scan","order","pme")): print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
