#!/bin/bash
# builds scratch/_dbg/libhmmc_base.so with the committed (HEAD) version of the named kernel files (default gemm_f16.hip)
# and the current objects of all others, for same-box A/B runs (HMMC_LIB=scratch/_dbg/libhmmc_base.so in the scratch benches)
set -e
cd /root/repo
mkdir -p scratch/_dbg /tmp/hmmc_base
files=${@:-gemm_f16.hip}
cp hmmc_amd/csrc/*.h /tmp/hmmc_base/
objs=""; skip=""
for f in $files; do
  git show HEAD:hmmc_amd/csrc/$f > /tmp/hmmc_base/$f
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I include -I /tmp/hmmc_base -c /tmp/hmmc_base/$f -o /tmp/hmmc_base/${f%.hip}.o
  objs="$objs /tmp/hmmc_base/${f%.hip}.o"; skip="$skip -e /${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_dbg/libhmmc_base.so $objs $(ls hmmc_amd/csrc/_obj/*.o | grep -v $skip)
echo built scratch/_dbg/libhmmc_base.so
