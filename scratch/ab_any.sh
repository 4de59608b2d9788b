#!/bin/bash
# A-B-..-B-A of one scratch benchmark between variant libraries: bash scratch/ab_any.sh tag "python scratch/attn_bench.py" base v1 v2
set -e
tag=$1; cmd=$2; shift 2
mkdir -p gpurun_out; rm -f gpurun_out/${tag}.log
rev=$(printf '%s\n' "$@" | tac | tr '\n' ' ')
for v in "$@" $rev; do
  echo "== $v" >> gpurun_out/${tag}.log
  HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$v.so timeout -k 10 200 $cmd 2>/dev/null >> gpurun_out/${tag}.log
done
cat gpurun_out/${tag}.log
