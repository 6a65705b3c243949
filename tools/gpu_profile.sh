#!/bin/bash
# Evidence set of one round on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile.sh <tag> [bench flags...]
# writes gpurun_out/<tag>/{bench.log,stats/,pmc_*/}; tools/save_profiles.py <tag> <round> copies the summaries to profiles/.
set -e
TAG=${1:-prof}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --no-single-scan --no-cpu-baseline $*"
python3 bench.py $* > $OUT/bench.log 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $B > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $B --steps 8 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $B --steps 8 > $OUT/pmc_write.log 2>&1
echo "traffic done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_valu -o run -- python3 $B --steps 8 > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_lds -o run -- python3 $B --steps 8 > $OUT/pmc_lds.log 2>&1
echo "counters done"
ls $OUT $OUT/*/ | head -60
