# every profile the round's DESIGN / bench figures cite, in one GPU session:  bash tools/session_profiles.sh r05
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
timeout 900 bash tools/profile_aev.sh $TAG > gpurun_out/${TAG}_profile_aev.log 2>&1
timeout 600 bash tools/profile_workload.sh ${TAG}_cfconv cfconv,half_slots,rows_cells,scan_half --workload cfconv --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_profile_cfconv.log 2>&1
timeout 600 bash tools/profile_workload.sh ${TAG}_conformers ani_ --workload conformers --no-cpu-baseline --no-shard8 --no-pmc > gpurun_out/${TAG}_profile_conformers.log 2>&1
timeout 600 bash tools/profile_workload.sh ${TAG}_neighbors pairs_,scan_rows,ani_,pme_ --workload neighbors --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_profile_neighbors.log 2>&1
timeout 600 bash tools/profile_workload.sh ${TAG}_torchani mlp_,ani_ --workload torchani --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_profile_torchani.log 2>&1
timeout 600 bash tools/profile_workload.sh ${TAG}_latency ani_ --workload latency --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_profile_latency.log 2>&1
ls -la gpurun_out | grep ${TAG}_ | wc -l
