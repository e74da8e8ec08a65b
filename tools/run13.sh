timeout 900 python -m pytest tests/test_ani_gpu.py tests/test_torch_surface_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -3
python tools/ab.py "NNPOPS_ANI_STREAMS=1" "NNPOPS_ANI_STREAMS=2" "NNPOPS_ANI_STREAMS=3" "NNPOPS_ANI_STREAMS=4" 2>&1 | tail -4
python bench.py --no-side --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_us'])"
NNPOPS_ANI_STREAMS=1 python bench.py --no-side --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_us'])"
