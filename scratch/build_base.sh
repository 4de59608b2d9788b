#!/bin/bash
# builds scratch/_dbg/libhmmc_base.so from the committed (HEAD) gemm_f16.hip + the current other objects, for same-box A/B runs
set -e
cd /root/repo
mkdir -p scratch/_dbg
git show HEAD:hmmc_amd/csrc/gemm_f16.hip > /tmp/gemm_f16_base.hip
cp hmmc_amd/csrc/common.h /tmp/common.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I include -I hmmc_amd/csrc -c /tmp/gemm_f16_base.hip -o /tmp/gemm_base.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_dbg/libhmmc_base.so /tmp/gemm_base.o $(ls hmmc_amd/csrc/_obj/*.o | grep -v gemm_f16.o)
echo built scratch/_dbg/libhmmc_base.so
