#!/bin/bash
# A-B-B-A of the default bench (or "$@") with an environment switch: bash scratch/ab_step.sh VAR [bench args]
var=$1; shift
for v in 0 1 1 0; do
  if [ $v = 1 ]; then export $var=1; else unset $var; fi
  python bench.py --no-cpu-baseline --vit-forward-iters 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_layout'])"
done
