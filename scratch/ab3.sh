#!/bin/bash
# A-B-C-C-B-A of scratch/gemm_bench.py between variant libraries (scratch/build_variant.sh), after the exactness check of each:
#   bash scratch/ab3.sh tag base dr1 dr2
set -e
tag=$1; shift
mkdir -p gpurun_out
rm -f gpurun_out/${tag}.log
for v in "$@"; do
  HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$v.so timeout -k 10 120 python scratch/check_lib.py > gpurun_out/${tag}_check_$v.log 2>&1
  echo "$v exact: $(grep -c exact gpurun_out/${tag}_check_$v.log)"
done
rev=$(printf '%s\n' "$@" | tac | tr '\n' ' ')
for v in "$@" $rev; do
  echo "== $v" >> gpurun_out/${tag}.log
  HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$v.so timeout -k 10 200 python scratch/gemm_bench.py 10 2>/dev/null >> gpurun_out/${tag}.log
done
grep "==\|layer\|kk \|km " gpurun_out/${tag}.log
