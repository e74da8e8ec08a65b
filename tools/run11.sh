mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ani_gpu.py tests/test_torch_surface_gpu.py -x -q -k "arbitrary or forward_batch or water18 or conformer" 2>&1 | tail -5
python tools/ab.py "NNPOPS_ANI_FWD_CHUNK=128" "NNPOPS_ANI_FWD_CHUNK=144" "NNPOPS_ANI_FWD_CHUNK=160" "NNPOPS_ANI_FWD_CHUNK=176" "NNPOPS_ANI_FWD_CHUNK=192" "NNPOPS_ANI_FWD_CHUNK=224" "NNPOPS_ANI_FWD_CHUNK=256" "NNPOPS_ANI_FWD_WPA=1" "NNPOPS_ANI_BACKWARD=0" "NNPOPS_ANI_BACKWARD=2" "NNPOPS_ANI_BACKWARD=3" "NNPOPS_ANI_FORWARD=1" 2>&1 | tail -14
python tools/ab.py --water "NNPOPS_ANI_FWD_CHUNK=128" "NNPOPS_ANI_FWD_CHUNK=192" "NNPOPS_ANI_FWD_CHUNK=256" "NNPOPS_ANI_FORWARD=1" "NNPOPS_ANI_FORWARD=0" "NNPOPS_ANI_BACKWARD=0" 2>&1 | tail -8
