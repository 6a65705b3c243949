#!/bin/bash
# per-kernel timeline of the map build for several library builds: tools/build_trace.sh <outdir> <lib> ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export NDT_LIB_PATH=$GRAFT_REPO_ROOT/$lib
  (cd $GRAFT_REPO_ROOT && python3 tools/prof_build.py 30 C3 | tail -1 | sed "s/^/$tag C3 /"; python3 tools/prof_build.py 10 C5 | tail -1 | sed "s/^/$tag C5 /")
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/prof_build.py 10 C3 > $OUT/$tag.log 2>&1)
  python3 - $OUT/$tag/run_kernel_trace.csv <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'map_' in n or 'scan_' in n or 'fill_f2' in n:
        n=n[n.find('::')+2:]; n=n[:n.find('(')]
        d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("   " + "  ".join("%s %.1f" % (k, sorted(v)[len(v)//2]) for k,v in d.items()))
PY
done
