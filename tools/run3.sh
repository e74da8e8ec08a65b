mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ani_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/t3.txt; cat gpurun_out/t3.txt
NNPOPS_ANI_FWD_WPA=1 NNPOPS_ANI_FWD_CHUNK=80 NNPOPS_ANI_FWD_APG=3 timeout 900 python -m pytest tests/test_ani_gpu.py -x -q -k "forward_kernels or water18 or strided or conformer" 2>&1 | tail -5 > gpurun_out/t3b.txt; cat gpurun_out/t3b.txt
NNPOPS_ANI_FWD_WPA=2 NNPOPS_ANI_FWD_CHUNK=80 NNPOPS_ANI_FWD_APG=3 timeout 900 python -m pytest tests/test_ani_gpu.py -x -q -k "forward_kernels or water18 or strided or conformer" 2>&1 | tail -5 > gpurun_out/t3c.txt; cat gpurun_out/t3c.txt
rm -f gpurun_out/b_*.json
for wpa in 1 2; do for ch in 128 160 192 256; do for apg in 1 2; do
NNPOPS_ANI_FWD_WPA=$wpa NNPOPS_ANI_FWD_CHUNK=$ch NNPOPS_ANI_FWD_APG=$apg python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_wpa${wpa}_ch${ch}_apg${apg}.json
done; done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us']['angular_forward'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
