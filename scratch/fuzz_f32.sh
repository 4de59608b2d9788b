#!/bin/bash
for pk in 7 9 0; do echo "== pick $pk"; HMMC_F32_PICK=$pk HMMC_LIB=$PWD/scratch/_dbg/libhmmc_f32dma.so timeout -k 10 500 python scratch/fuzz_f32_dma.py 150 $pk 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -12; done
