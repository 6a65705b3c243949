#!/bin/bash
# per-kernel times of a bench run under rocprofv3: tools/kstats.sh <tag> [bench flags]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --no-single-scan --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, json
for r in csv.DictReader(open(sys.argv[1] + "/stats/run_kernel_stats.csv")):
    print("%-50s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
d = json.loads([l for l in open(sys.argv[1] + "/stats.log") if l.startswith("{")][-1])
print("bench under the profiler: value %.0f ms/step %.4f interval_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
