# rocprofv3 kernel stats of any command, top kernels printed:   bash tools/prof_cmd.sh <tag> <command...>
# (run on the GPU box: gpurun -- 'bash tools/prof_cmd.sh pairs python tools/pairs_bwd_time.py')
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o kt --output-format csv -- "$@" > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us  {float(r['Percentage']):6.2f}%  {r['Name'][:110]}")
PY
