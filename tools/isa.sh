#!/bin/bash
# ISA of ndt_align_kernel<true,false> for the current sources: tools/isa.sh [out.s] [extra hipcc flags]
OUT=${1:-/root/repo/build_tmp/align.s}; shift
mkdir -p /root/repo/build_tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -I/root/repo/include "$@" \
  -S --cuda-device-only -o /root/repo/build_tmp/ndt_all.s /root/repo/ndt_slam_amd/csrc/ndt_mi355x.hip 2>/dev/null
a=$(grep -n "^_ZN12_GLOBAL__N_116ndt_align_kernelILb1ELb0ELb0E.*:" /root/repo/build_tmp/ndt_all.s | head -1 | cut -d: -f1)
b=$(grep -n "\.amdhsa_kernel _ZN12_GLOBAL__N_116ndt_align_kernelILb1ELb0ELb0E" /root/repo/build_tmp/ndt_all.s | cut -d: -f1)
sed -n "${a},${b}p" /root/repo/build_tmp/ndt_all.s > $OUT
wc -l $OUT
