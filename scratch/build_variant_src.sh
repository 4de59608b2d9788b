#!/bin/bash
# builds scratch/_dbg/libhmmc_<name>.so with ONE source of hmmc_amd/csrc rebuilt from the working tree with extra -D flags:
#   bash scratch/build_variant_src.sh attention_f16 noP0 -DHMMC_SCRATCH -DHMMC_ATTN_ABL=1
set -e
cd /root/repo
src=$1; name=$2; shift 2
mkdir -p scratch/_dbg /tmp/hmmc_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I include -I hmmc_amd/csrc "$@" -c hmmc_amd/csrc/$src.hip -o /tmp/hmmc_var/${src}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_dbg/libhmmc_$name.so /tmp/hmmc_var/${src}_$name.o $(ls hmmc_amd/csrc/_obj/*.o | grep -v /$src.o)
echo built scratch/_dbg/libhmmc_$name.so
