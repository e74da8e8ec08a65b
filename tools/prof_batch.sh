# per-kernel rocprofv3 durations of the 1 024-conformer batch: round-4 library vs working tree
cd /tmp && export TMPDIR=/tmp
for v in "LIB=tools/_ref/libnnpops_hip_r04.so" ""; do
  rm -rf /tmp/pb
  ( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/pb -o kt --output-format csv -- python tools/ab.py "$v" --rounds 3 --workload batch > /dev/null 2>&1 )
  echo "== variant '$v'"
  f=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name']
    if 'ani_' not in name: continue
    key=(name[:70], r.get('Grid_Size') or r.get('Grid_Size_X'), r.get('LDS_Block_Size') )
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v)
    print(f"{len(v):5d} med {v[len(v)//2]:8.2f} us  grid {k[1]} lds {k[2]}  {k[0]}")
PY
done
