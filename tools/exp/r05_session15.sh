#!/bin/bash
mkdir -p gpurun_out/s15
for rep in 1 2; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  OTGAN_FORK_REAL=$1 OTGAN_FORK_WGRAD=$2 timeout 600 python bench.py --steps 24 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('fork_real=$1 fork_wgrad=$2', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'], d['config']['last_distance'])"
done; done > gpurun_out/s15/ab.txt 2>&1
for v in "0 0" "1 1"; do set -- $v
  OTGAN_FORK_REAL=$1 OTGAN_FORK_WGRAD=$2 timeout 600 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('densenet fork_real=$1 fork_wgrad=$2', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'], d['config']['last_distance'])"
done >> gpurun_out/s15/ab.txt 2>&1
cat gpurun_out/s15/ab.txt
