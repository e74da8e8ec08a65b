# (a pass with the TA_* counters did not finish in 15 minutes on this pool: every pass now runs under its own timeout)
# Memory-path counters of the fused networks alone (run on the GPU box):  bash tools/profile_mlp.sh <tag> [n_waters]
#   -> gpurun_out/<tag>_mlp_memory_path_pmc.txt
set -x
TAG=${1:-r03}; W=${2:-667}; EXTRA=${3:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/mlp_bench.py $W $EXTRA"
i=0
for set in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $O/prof_m$i -o m --output-format rocpd -- $P > /dev/null 2>&1
done
cd $R
python tools/pmc_report.py $(find gpurun_out/prof_m* -name "*.db") --filter mlp_forward,mlp_input_grad,mlp_sum_members > gpurun_out/${TAG}_mlp_memory_path_pmc.txt
rm -rf gpurun_out/prof_m*
cat gpurun_out/${TAG}_mlp_memory_path_pmc.txt
