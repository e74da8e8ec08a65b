set -x
mkdir -p gpurun_out
./tools/ubench/mfma4x4_layout > gpurun_out/layout.txt 2>&1; cat gpurun_out/layout.txt
timeout 900 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/t1.txt; cat gpurun_out/t1.txt
for f in 0 1 2; do NNPOPS_ANI_FORWARD=$f python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_f$f.json; done
for ch in 64 192 256; do NNPOPS_ANI_FWD_CHUNK=$ch python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_ch$ch.json; done
for w in 16 20 24; do NNPOPS_ANI_FWD_WAVES=$w python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_w$w.json; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
