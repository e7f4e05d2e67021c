#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/s14; mkdir -p $O
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  for m in 0 f 1; do
    OTGAN_DENSE16_CHAIN=$m python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet chain=$m" | tee -a $O/dn.txt
  done
done
