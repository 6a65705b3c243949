#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the match kernel for library builds: tools/traffic.sh <lib|-> ...
export TMPDIR=/tmp
for lib in "$@"; do
  [ "$lib" = "-" ] && unset NDT_LIB_PATH || export NDT_LIB_PATH=$PWD/$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/tr_tmp; mkdir -p gpurun_out/tr_tmp
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/tr_tmp -o run -- python3 bench.py --no-single-scan --no-cpu-baseline --steps 6 > /dev/null 2>&1
    python3 - "$lib" $c <<'PY'
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("gpurun_out/tr_tmp/run_counter_collection.csv")) if "ndt_align" in r["Kernel_Name"]]
print("%-28s %-10s %.0f MB" % (sys.argv[1], sys.argv[2], sum(v)/len(v)*1024/1e6))
PY
  done
done
