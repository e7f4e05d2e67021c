#!/bin/bash
# loss after 1, 2, 3, 6, 12, 24 steps of the DenseNet configuration with and without the dense-block split (two fp32
# evaluation orders of the same step: the trajectories must agree to rounding at first and stay close)
for k in 1 2 3 6 12 24; do
  a=$(python bench.py --model densenet --nr_sinkhorn_iter 200 --steps $k --warmup 0 --no_cpu_baseline --no_prof --no_secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['last_distance'], d['config']['last_entropy'])")
  b=$(OTGAN_DENSE_SPLIT=0 OTGAN_DISABLE_X_OPERAND=1 OTGAN_WN_BATCHED=0 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps $k --warmup 0 --no_cpu_baseline --no_prof --no_secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['last_distance'], d['config']['last_entropy'])")
  echo "steps=$k split=$a chain=$b"
done
