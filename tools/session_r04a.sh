# round 4, GPU session a: where the time goes (probes), fusion-A bound on two frames, bench line vs rocprofv3
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_ani.py --lib tools/_probe/libnnpops_hip.so > $O/r04a_probe_10k.json 2> $O/r04a_probe.err; tail -3 $O/r04a_probe.err
timeout 600 python tools/proto_fused_aev.py --lib tools/_proto/libnnpops_hip.so > $O/r04_fusionA_upper_bound.json 2> $O/r04a_proto.err; tail -3 $O/r04a_proto.err
timeout 600 python bench.py --no-side --no-cpu-baseline > $O/r04a_bench_head.json 2> $O/r04a_bench_head.err; tail -3 $O/r04a_bench_head.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt --output-format rocpd -- python $R/bench.py --no-side --no-cpu-baseline --no-pmc > $O/r04a_bench_under_rocprofv3.json 2> $O/r04a_bench_under_rocprofv3.err
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_kt -name "*.db") gpurun_out/r04a_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-side --no-cpu-baseline --no-pmc" > /dev/null
rm -rf gpurun_out/prof_kt
head -14 gpurun_out/r04a_kernel_stats.txt
cat $O/r04a_bench_head.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_us'], d['kernels_us_sum'], d['bracket_correction_us'], d['roofline']['frac'])"
cat $O/r04_fusionA_upper_bound.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_probe_10k.json'))
for p in d['probes']: print(p['mask'], p['us'], p['off'])
PY
