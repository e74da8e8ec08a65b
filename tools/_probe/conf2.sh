cd /tmp && export TMPDIR=/tmp
for v in "NNPOPS_ANI_FWD_DYN=0" "NNPOPS_ANI_FWD_DYN=1"; do
  echo "== $v"
  env $v rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o kt --output-format rocpd -- python $GRAFT_REPO_ROOT/bench.py --workload conformers --steps 50 --warmup 5 --no-cpu-baseline --no-shard8 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_$v -name "*.db") /tmp/ks_$v.txt "conformers" > /dev/null; head -9 /tmp/ks_$v.txt | tail -5 | cut -c1-140
done
