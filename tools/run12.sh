python tools/ab.py "NNPOPS_ANI_STORE=0" "NNPOPS_ANI_STORE=1" "NNPOPS_ANI_STORE=2" "NNPOPS_ANI_STORE=3" 2>&1 | tail -5
NNPOPS_ANI_STORE=1 timeout 600 python -m pytest tests/test_ani_gpu.py -x -q -k "water18 or conformer or periodic_box" 2>&1 | tail -2
