#!/bin/bash
mkdir -p gpurun_out/s14
for rep in 1 2; do for v in 1 0; do
  OTGAN_FORK_REAL=$v timeout 600 python bench.py --steps 24 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('fork_real=$v', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'], d['config']['last_distance'])"
done; done > gpurun_out/s14/ab.txt 2>&1
cat gpurun_out/s14/ab.txt
