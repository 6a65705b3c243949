#!/bin/bash
# A/B of library builds with result hashes: tools/abh.sh <outdir> <lib or -> ...   (bench line + tools/hash_results.py per build)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ "$lib" != "-" ]; then export NDT_LIB_PATH=$PWD/$lib; else unset NDT_LIB_PATH; fi
  for rep in 1 2; do
  timeout -k 10 240 python3 bench.py --no-single-scan --no-cpu-baseline --warmup 20 --steps 100 > $OUT/ab_$i.log 2> $OUT/ab_$i.err || { echo "build $i failed"; tail -5 $OUT/ab_$i.err; exit 1; }
  python3 - "$OUT/ab_$i.log" "$lib" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%-28s value %8.0f ms/step %.4f match_ms %.4f fit_ms %.4f frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["fitness"]["ms"], r["frac"]))
PY
  done
  timeout -k 10 300 python3 tools/hash_results.py ${HASH_FLAGS} > $OUT/hash_$i.log 2>&1 || { echo "hash $i failed"; tail -5 $OUT/hash_$i.log; exit 1; }
  cat $OUT/hash_$i.log
done
