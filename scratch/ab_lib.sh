#!/bin/bash
# A-B-B-A of scratch/gemm_bench.py between two variant libraries (scratch/build_variant.sh), after the exactness check of B:
#   bash scratch/ab_lib.sh base4 ph2 [log tag]
set -e
a=$1; b=$2; tag=${3:-ab}
HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$b.so timeout -k 10 120 python scratch/check_lib.py > gpurun_out/${tag}_check.log 2>&1
grep -c exact gpurun_out/${tag}_check.log
rm -f gpurun_out/${tag}.log
for v in $a $b $b $a; do
  echo "== $v" >> gpurun_out/${tag}.log
  HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$v.so timeout -k 10 200 python scratch/gemm_bench.py 10 2>/dev/null >> gpurun_out/${tag}.log
done
grep "==\|layer\|mm w\|km dqkv\|kk qkv" gpurun_out/${tag}.log
