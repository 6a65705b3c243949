#!/bin/bash
# per-build phase timers: tools/prof_fit.sh <lib> ...
for lib in "$@"; do
  export NDT_LIB_PATH=$PWD/$lib
  python3 tools/prof_phases.py 256 > /tmp/pp.log 2>&1
  echo "== $lib"; python3 tools/prof_analyze.py gpurun_out/prof_dump_batch.bin | grep -v "^  scan\|histogram\|attach"; grep "align ms" /tmp/pp.log | tail -2
done
