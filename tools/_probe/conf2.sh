cd /tmp && export TMPDIR=/tmp
for v in "NNPOPS_ANI_BWD_CLASSES=1" "NNPOPS_ANI_BWD_CLASSES=1 NNPOPS_ANI_BACKWARD=3" "NNPOPS_ANI_BWD_CLASSES=1 NNPOPS_ANI_BACKWARD=1"; do
  echo "== $v"
  rm -rf /tmp/prof_x
  env $v rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o kt --output-format rocpd -- python $GRAFT_REPO_ROOT/bench.py --workload conformers --steps 50 --warmup 5 --no-cpu-baseline --no-shard8 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_x -name "*.db") /tmp/ks_x.txt "conformers" > /dev/null; head -10 /tmp/ks_x.txt | tail -6 | cut -c1-150
done
