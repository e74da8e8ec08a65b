mkdir -p gpurun_out
./tools/ubench/valu_issue > gpurun_out/valu_issue.txt 2>&1; cat gpurun_out/valu_issue.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export NNPOPS_ANI_OCC=5
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM -d $O/pmcC -o pmcC --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES -d $O/pmcD -o pmcD --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcD.log 2>&1
NNPOPS_ANI_BACKWARD=0 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM -d $O/pmcE -o pmcE --output-format rocpd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmcE.log 2>&1
cd $R
for x in C D E; do python tools/pmc_report.py $(find gpurun_out/pmc$x -name "*.db") --filter ani_ > gpurun_out/pmc$x.txt 2>&1; rm -rf gpurun_out/pmc$x; done
cat gpurun_out/pmcC.txt gpurun_out/pmcD.txt; grep -A9 "ani_angular_backward<" gpurun_out/pmcE.txt
