cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/pp$v
  NNPOPS_PAIRS_FINE_GRID=$v rocprofv3 --kernel-trace --stats -d /tmp/pp$v -o kt --output-format csv -- python $GRAFT_REPO_ROOT/tools/pairs_ab.py "" > /dev/null 2>&1
  echo "== FINE_GRID=$v"
  f=$(find /tmp/pp$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:90]}")
PY
done
