set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
export NNPOPS_ANI_FWD_CHUNK=192
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmcA -o pmcA --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/pmcB -o pmcB --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcB.log 2>&1
cd $R
python tools/pmc_report.py $(find gpurun_out/pmcA -name "*.db") --filter ani_ > gpurun_out/pmcA.txt 2>&1
python tools/pmc_report.py $(find gpurun_out/pmcB -name "*.db") --filter ani_ > gpurun_out/pmcB.txt 2>&1
cat gpurun_out/pmcA.txt gpurun_out/pmcB.txt
rm -rf gpurun_out/pmcA gpurun_out/pmcB
