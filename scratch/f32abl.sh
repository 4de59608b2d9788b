#!/bin/bash
for lib in "$@"; do
  echo "== $lib"
  HMMC_F32_PICK=7 HMMC_LIB=$PWD/scratch/_dbg/libhmmc_$lib.so timeout -k 10 300 python scratch/gemm32_dma.py 3072 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "qkv \|fc \|dproj\|mlp1\|mlm \|moco\|wfc"
done
