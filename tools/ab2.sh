#!/bin/bash
# A/B of library builds with per-kernel times: tools/ab2.sh <tag> <lib> ...
for lib in "$@"; do
  [ "$lib" = "-" ] && unset NDT_LIB_PATH || export NDT_LIB_PATH=$PWD/$lib
  echo "== $lib"; tools/kstats.sh ab2_tmp | grep "calls   23\|bench under"
done
