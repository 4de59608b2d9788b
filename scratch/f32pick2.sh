#!/bin/bash
# dispatcher with and without the LDS-DMA kernel at several token counts: bash scratch/f32pick2.sh <lib> tokens...
lib=$1; shift
for T in "$@"; do for off in 1 0; do
  echo "== T=$T $( [ $off = 1 ] && echo without || echo with ) DMA kernel"
  if [ $off = 1 ]; then export HMMC_NO_F32_DMA=1; else unset HMMC_NO_F32_DMA; fi
  HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$lib.so timeout -k 10 300 python scratch/gemm32_dma.py $T 2>&1 | grep -v "Warning\|amdgpu.ids"
done; done
