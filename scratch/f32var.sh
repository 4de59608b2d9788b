#!/bin/bash
# bash scratch/f32var.sh tokens pick libs...
T=$1; pk=$2; shift 2
for lib in "$@" $(printf '%s\n' "$@" | tac); do
  echo "== $lib pick $pk"
  HMMC_F32_PICK=$pk HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$lib.so timeout -k 10 300 python scratch/gemm32_dma.py $T 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "total\|qkv\|fc \|mlm \|moco"
done
