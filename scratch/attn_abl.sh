#!/bin/bash
# waves per workgroup of the short attention kernels (scratch/build_variant_src.sh attention_f16 fw<N> -DHMMC_SCRATCH -DHMMC_ATTN_FWD_WPB=<N>)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in fw4 fw2 fw1 fw2 fw4; do
  echo "== $v"; HMMC_LIB=$R/scratch/_dbg/libhmmc_$v.so timeout -k 10 120 python scratch/attn_bench.py 2>/dev/null | grep attention
done
