#!/bin/bash
# tools/pmc_build.sh <tag>: PMC counters of the map build's kernels in a build-only loop (tools/prof_build.py), one counter
# group per rocprofv3 pass (FETCH_SIZE and WRITE_SIZE asked for together abort the profiler); run through gpurun
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/a -o run -- python3 $GRAFT_REPO_ROOT/tools/prof_build.py 10 C3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT --output-format csv -d $OUT/b -o run -- python3 $GRAFT_REPO_ROOT/tools/prof_build.py 10 C3 > $OUT/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c -o run -- python3 $GRAFT_REPO_ROOT/tools/prof_build.py 10 C3 > $OUT/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d -o run -- python3 $GRAFT_REPO_ROOT/tools/prof_build.py 10 C3 > $OUT/d.log 2>&1
python3 - $OUT <<'PY'
import csv,sys,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if not any(k in n for k in ("map_","scan_","fill_f2")): continue
        n=n[n.find("::")+2:]; n=n[:n.find("(")]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in agg.items():
    print(n, " ".join("%s=%.3g" % (k, sum(v)/len(v)) for k,v in sorted(c.items())))
PY
