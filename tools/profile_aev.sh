# Profiles of the default bench command (run on the GPU box): kernel stats + timeline, SQ counters, LDS counters, HBM traffic.
#   bash tools/profile_aev.sh <tag>      -> gpurun_out/<tag>_*.txt
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt --output-format rocpd -- python $R/bench.py --no-side --no-cpu-baseline --no-pmc > $O/${TAG}_bench_under_rocprofv3.json 2> $O/${TAG}_bench_under_rocprofv3.err
P="python $R/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-side --no-pmc"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/prof_sq -o sq --output-format rocpd -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/prof_lds -o lds --output-format rocpd -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o fetch --output-format rocpd -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o write --output-format rocpd -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/prof_tcc -o tcc --output-format rocpd -- $P > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_kt -name "*.db") gpurun_out/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-side --no-cpu-baseline" > /dev/null
python tools/rocprof_timeline.py $(find gpurun_out/prof_kt -name "*.db") 18 > gpurun_out/${TAG}_timeline.txt
python tools/pmc_report.py $(find gpurun_out/prof_sq -name "*.db") --filter ani > gpurun_out/${TAG}_sq_counters_pmc.txt
python tools/pmc_report.py $(find gpurun_out/prof_lds -name "*.db") --filter ani > gpurun_out/${TAG}_lds_counters_pmc.txt
python tools/pmc_report.py $(find gpurun_out/prof_fetch -name "*.db") $(find gpurun_out/prof_write -name "*.db") $(find gpurun_out/prof_tcc -name "*.db") --filter ani > gpurun_out/${TAG}_hbm_traffic_pmc.txt
rm -rf gpurun_out/prof_kt gpurun_out/prof_sq gpurun_out/prof_lds gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_tcc
head -12 gpurun_out/${TAG}_kernel_stats.txt; cat gpurun_out/${TAG}_timeline.txt; cat gpurun_out/${TAG}_hbm_traffic_pmc.txt; cat gpurun_out/${TAG}_lds_counters_pmc.txt | grep -v "row_stats" 
