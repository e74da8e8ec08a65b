cd $GRAFT_REPO_ROOT
NNPOPS_CFCONV_BWD_WAVES=4 python -m pytest tests/test_cfconv_gpu.py -x -q -m gpu 2>&1 | tail -1
for v in 0 1 2; do
  NNPOPS_CFCONV_X2=$v NNPOPS_CFCONV_BWD_WAVES=4 python -m pytest tests/test_cfconv_gpu.py -x -q -m gpu -k "golden or oracle or matches" 2>&1 | tail -1
  NNPOPS_CFCONV_X2=$v NNPOPS_CFCONV_BWD_WAVES=4 bash tools/prof_cmd.sh cfp python $GRAFT_REPO_ROOT/bench.py --workload cfconv --steps 60 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | grep "filters_h2" | awk -v m=$v '{print "x2 variant",m,$2,$3,substr($5,1,50)}'
done
