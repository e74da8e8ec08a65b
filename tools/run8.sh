mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/t8.txt; cat gpurun_out/t8.txt
NNPOPS_ANI_OCC=6 timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/t8b.txt; cat gpurun_out/t8b.txt
rm -f gpurun_out/b_*.json
for occ in 5 6; do NNPOPS_ANI_OCC=$occ python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_occ${occ}.json; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
