#!/bin/bash
# A-B-B-A of bench.py between two values of an environment variable: bash scratch/ab_env.sh VAR A B [bench args]
var=$1; a=$2; b=$3; shift 3
for v in $a $b $b $a; do
  env $var=$v python bench.py --no-cpu-baseline --no-hbm-roofline --vit-forward-iters 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
