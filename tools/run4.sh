mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ani_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/t4.txt; cat gpurun_out/t4.txt
rm -f gpurun_out/b_*.json
for bw in 0 1; do NNPOPS_ANI_BACKWARD=$bw python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_bw${bw}.json; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['kernels_us'])
    except Exception as e: print(f, 'ERR', open(f).read()[-300:])
P
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmcA -o pmcA --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcA.log 2>&1
cd $R
python tools/pmc_report.py $(find gpurun_out/pmcA -name "*.db") --filter ani_angular > gpurun_out/pmcA.txt 2>&1
cat gpurun_out/pmcA.txt
rm -rf gpurun_out/pmcA
