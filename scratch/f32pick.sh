#!/bin/bash
# bash scratch/f32pick.sh <lib name> <tokens> picks...
lib=$1; T=$2; shift 2
mkdir -p gpurun_out
for pk in "$@"; do
  echo "== pick $pk"
  HMMC_F32_PICK=$pk HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$lib.so timeout -k 10 300 python scratch/gemm32_dma.py $T 2>&1 | grep -v "Warning\|amdgpu.ids"
done
