mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t14.txt; cat gpurun_out/t14.txt
bash tools/profile_aev.sh r02b > gpurun_out/profile_r02b.log 2>&1
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
head -12 gpurun_out/r02b_kernel_stats.txt
