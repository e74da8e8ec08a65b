mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/t10.txt; cat gpurun_out/t10.txt
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['kernels_us']); print(d.get('cpu_baseline'))
for k,v in d.get('side',{}).items(): print(k, {kk: v.get(kk) for kk in ('value','unit','ms_per_step','error','eager_us','hip_graph_us')}, (v.get('cpu_baseline') or {}).get('value'))
P
