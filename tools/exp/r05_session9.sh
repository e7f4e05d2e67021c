#!/bin/bash
mkdir -p gpurun_out/s9
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_train_step_gpu.py tests/test_dist_gpu.py tests/test_rccl_gpu.py tests/test_step_graph_gpu.py tests/test_sinkhorn_soak_gpu.py tests/test_train_main_gpu.py tests/test_cfg5_gpu.py -x -q > gpurun_out/s9/tests.txt 2>&1
for rep in 1 2; do for v in 1 0; do
  OTGAN_ONE_CRITIC_PASS=$v timeout 600 python bench.py --steps 24 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('one_pass=$v', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'])"
done; done > gpurun_out/s9/ab.txt 2>&1
tail -4 gpurun_out/s9/tests.txt; cat gpurun_out/s9/ab.txt
