#!/bin/bash
mkdir -p gpurun_out/s18
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_side_stream_gpu.py tests/test_layers_gpu.py tests/test_models_gpu.py -x -q -k "stream or dense or densenet" > gpurun_out/s18/tests.txt 2>&1
for rep in 1 2; do for v in 0 1; do
  OTGAN_SIDE_STREAM=$v timeout 600 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('densenet side_stream=$v', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'], d['config']['last_distance'])"
done; done > gpurun_out/s18/ab.txt 2>&1
tail -4 gpurun_out/s18/tests.txt; cat gpurun_out/s18/ab.txt
