#!/bin/bash
# A-B-B-A of bench.py between VAR unset and VAR=1: bash scratch/ab_env2.sh VAR "bench args"
var=$1; shift
for v in 1 0 0 1; do
  if [ $v = 1 ]; then export $var=1; else unset $var; fi
  echo "== $var=$v"
  timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-hbm-roofline --roofline-steps 0 --vit-forward-iters 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
