#!/bin/bash
# A-B-B-A of bench.py between this tree (B) and the snapshot of an earlier commit under scratch/_base (A): bash scratch/ab_tree.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/ab_$TAG; mkdir -p $O
cd /tmp
i=0
for t in A B B A; do
  i=$((i+1))
  if [ $t = A ]; then D=$R/scratch/_base; else D=$R; fi
  (cd $D && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-hbm-roofline --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0 "$@" > $O/$t$i.log 2>&1) || { echo "run $t$i failed"; tail -5 $O/$t$i.log; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$O/$t$i.log') if l.startswith('{')][-1]); print('$TAG $t', d['ms_per_step'])"
done
