#!/bin/bash
# builds scratch/_dbg/libhmmc_<name>.so from the working-tree gemm_f16.hip with extra -D flags (same-box A/B runs):
#   bash scratch/build_variant.sh sched1 -DHMMC_GEMM_SCHED=1
set -e
cd /root/repo
name=$1; shift
mkdir -p scratch/_dbg /tmp/hmmc_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I include -I hmmc_amd/csrc "$@" -c hmmc_amd/csrc/gemm_f16.hip -o /tmp/hmmc_var/gemm_f16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_dbg/libhmmc_$name.so /tmp/hmmc_var/gemm_f16_$name.o $(ls hmmc_amd/csrc/_obj/*.o | grep -v /gemm_f16.o)
echo built scratch/_dbg/libhmmc_$name.so
