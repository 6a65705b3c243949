#!/bin/bash
# A/B of library builds and bench flags in one gpurun call: tools/ab.sh <outdir> "<lib>|<bench flags>" ...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
i=0
for spec in "$@"; do
  lib="${spec%%|*}"; flags="${spec#*|}"
  i=$((i+1))
  if [ -n "$lib" ]; then export NDT_LIB_PATH=$PWD/$lib; else unset NDT_LIB_PATH; fi
  timeout -k 10 240 python3 bench.py --no-single-scan --no-cpu-baseline $flags > $OUT/ab_$i.log 2> $OUT/ab_$i.err || { echo "spec $i failed"; tail -5 $OUT/ab_$i.err; exit 1; }
  python3 - "$OUT/ab_$i.log" "$spec" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
    print("%-50s value %8.0f ms/step %.4f match_ms %.4f fit_ms %.4f frac %.4f evals %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["fitness"]["ms"], r["frac"], r["mean_evals"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
