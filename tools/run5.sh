mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/t5.txt; cat gpurun_out/t5.txt
NNPOPS_ANI_OCC=5 timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/t5b.txt; cat gpurun_out/t5b.txt
rm -f gpurun_out/b_*.json
for occ in 5 6; do for bw in 0 1; do NNPOPS_ANI_OCC=$occ NNPOPS_ANI_BACKWARD=$bw python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_occ${occ}_bw${bw}.json; done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
