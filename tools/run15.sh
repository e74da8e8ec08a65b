timeout 900 python -m pytest tests/test_ani_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
python tools/ab.py "NNPOPS_ANI_FWD_ROWLDS=0" "NNPOPS_ANI_FWD_ROWLDS=1" "NNPOPS_ANI_FWD_ROWLDS=1 NNPOPS_ANI_STORE=0" 2>&1 | tail -3
python tools/ab.py --water "NNPOPS_ANI_FWD_ROWLDS=0" "NNPOPS_ANI_FWD_ROWLDS=1" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o write --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > /dev/null 2>&1
cd $R; python tools/pmc_report.py $(find gpurun_out/prof_write -name "*.db") --filter ani_angular_forward; rm -rf gpurun_out/prof_write
