#!/bin/bash
# fp32 LDS-DMA kernel on / off (HMMC_NO_F32_DMA) in whole steps: fine-tune b = 256 and pre-training, A-B-B-A
mkdir -p gpurun_out
for mode in "" "--mode pretrain"; do
  for v in 1 0 0 1; do
    if [ $v = 1 ]; then export HMMC_NO_F32_DMA=1; else unset HMMC_NO_F32_DMA; fi
    echo "== mode '$mode' NO_F32_DMA=$v"
    timeout -k 10 400 python bench.py $mode --no-cpu-baseline --no-hbm-roofline --steps 6 --warmup 3 --roofline-steps 0 --vit-forward-iters 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
