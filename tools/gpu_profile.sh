#!/bin/bash
# Evidence set of one round on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile.sh <tag> [bench flags...]
# writes gpurun_out/<tag>/{bench.log,stats/,pmc_*/,build_*/,alone_*/,c5_*/}; tools/save_profiles.py <tag> <round> copies the
# summaries to profiles/.  One counter group per rocprofv3 pass, never together with a trace (gpurun refuses the combination).
# Parts (round 5): the pipelined bench (as before) -- the map build ALONE (build-only loop: trace + FETCH /
# WRITE) -- the match + fitness kernels ALONE (synchronous loop, no build beside them: trace) -- `--config C5` (trace + the SQ
# counters of its fitness kernels) -- VMEM / FLAT instruction counts of the match kernel (scratch traffic would show there).
set -e
TAG=${1:-prof}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --no-single-scan --no-cpu-baseline $*"
python3 bench.py $* > $OUT/bench.log 2> $OUT/bench.err
echo "bench done"; tail -c 300 $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $B > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $B --steps 8 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $B --steps 8 > $OUT/pmc_write.log 2>&1
echo "traffic done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_valu -o run -- python3 $B --steps 8 > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_lds -o run -- python3 $B --steps 8 > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_FLAT_LDS_ONLY --output-format csv -d $OUT/pmc_vmem -o run -- python3 $B --steps 8 > $OUT/pmc_vmem.log 2>&1 || echo "pmc_vmem pass failed (counter names?)"
echo "counters done"
# (i) the map build alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build_stats -o run -- python3 tools/prof_build.py 20 C3 > $OUT/build_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/build_fetch -o run -- python3 tools/prof_build.py 10 C3 > $OUT/build_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/build_write -o run -- python3 tools/prof_build.py 10 C3 > $OUT/build_write.log 2>&1
echo "build done"
# (ii) match + fitness kernels alone (synchronous launches: nothing else on the chip)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone_stats -o run -- python3 tools/prof_single.py 256 12 > $OUT/alone_stats.log 2>&1
echo "alone done"
# (iii) configs[4]
python3 bench.py --config C5 --no-cpu-baseline > $OUT/c5_bench.log 2> $OUT/c5_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o run -- python3 bench.py --config C5 --no-cpu-baseline --no-single-scan > $OUT/c5_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/c5_valu -o run -- python3 bench.py --config C5 --no-cpu-baseline --no-single-scan --steps 6 > $OUT/c5_valu.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o run -- python3 bench.py --config C5 --no-cpu-baseline --no-single-scan --steps 6 > $OUT/c5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -o run -- python3 bench.py --config C5 --no-cpu-baseline --no-single-scan --steps 6 > $OUT/c5_write.log 2>&1
echo "c5 done"
ls $OUT | head -60
