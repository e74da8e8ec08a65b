# round 4, final measurement session: default bench line (with side workloads and in-run PMC passes), rocprofv3 kernel stats /
# timeline / counters of the same command, kernel stats of the side workloads, the probe breakdown at the final code
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python bench.py > $O/r04_bench_default_with_side.json 2> $O/r04_bench_default_with_side.err; tail -3 $O/r04_bench_default_with_side.err
timeout 900 bash tools/profile_aev.sh r04 > $O/r04_profile_aev.log 2>&1
timeout 600 python tools/probe_ani.py --lib tools/_probe/libnnpops_hip.so > $O/r04_probe_10k.json 2> $O/r04_probe.err
cd /tmp && export TMPDIR=/tmp
for W in neighbors conformers latency torchani cfconv; do
  rm -rf /tmp/prof_w
  EXTRA=""; [ $W = conformers ] && EXTRA="--no-shard8"
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o kt --output-format rocpd -- python $R/bench.py --workload $W --no-cpu-baseline $EXTRA > $O/r04_${W}_bench_under_rocprofv3.json 2> /dev/null
  python $R/tools/rocprof_summary.py $(find /tmp/prof_w -name "*.db") $O/r04_${W}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --no-cpu-baseline $EXTRA" > /dev/null
done
cd $R
head -12 $O/r04_kernel_stats.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_default_with_side.json'))
print(d['value'], d['ms_per_step'], d['kernels_us'], d['kernels_us_sum'], d['bracket_correction_us'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['step'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
for k,v in d.get('side',{}).items():
    print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
print(d.get('side_errors'))
PY
