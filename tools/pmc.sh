#!/bin/bash
# Mean of hardware counters per kernel over a short bench run: tools/pmc.sh <tag> "<bench flags>" "<counters of pass 1>" ["<pass 2>" ...]
# (one rocprofv3 --pmc run per pass, nothing else traced; the program directly behind `--`)
TAG=$1; FLAGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1)); rm -rf $OUT/p$i
  rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -o run -- python3 bench.py --no-single-scan --no-cpu-baseline --steps 6 $FLAGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
  python3 - $OUT/p$i/run_counter_collection.csv <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(ndt_align_kernel|fitness_\w+|map_\w+)", r["Kernel_Name"])
    if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k.startswith("map_"): continue
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
done
