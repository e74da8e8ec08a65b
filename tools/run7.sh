mkdir -p gpurun_out
for bw in 2 3 4; do NNPOPS_ANI_BACKWARD=$bw timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/t7_$bw.txt; cat gpurun_out/t7_$bw.txt; done
NNPOPS_ANI_FWD_EARLY=1 timeout 1200 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/t7_e.txt; cat gpurun_out/t7_e.txt
rm -f gpurun_out/b_*.json
for bw in 0 1 2 3 4; do for e in 0 1; do NNPOPS_ANI_FWD_EARLY=$e NNPOPS_ANI_BACKWARD=$bw python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_bw${bw}_e$e.json; done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
