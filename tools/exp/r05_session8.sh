#!/bin/bash
mkdir -p gpurun_out/s8
export OMP_NUM_THREADS=16
for rep in 1 2; do
for g in 1 0; do
  for model in dcgan densenet; do
    it=100; [ $model = densenet ] && it=200
    OTGAN_STEP_GRAPH=$g timeout 600 python bench.py --model $model --nr_sinkhorn_iter $it --steps 24 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('graph=$g', '$model', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'])"
  done
done
done > gpurun_out/s8/ab.txt 2>&1
cat gpurun_out/s8/ab.txt
