#!/bin/bash
mkdir -p gpurun_out/s17
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_side_stream_gpu.py tests/test_step_graph_gpu.py tests/test_train_step_gpu.py tests/test_train_main_gpu.py tests/test_amax_fused_gpu.py -x -q > gpurun_out/s17/tests.txt 2>&1
for rep in 1 2; do for v in "0 1" "1 0" "1 1"; do set -- $v
  OTGAN_SIDE_STREAM=$1 OTGAN_FORK_OPT=$2 timeout 600 python bench.py --steps 24 --warmup 6 --no_cpu_baseline --no_secondary --no_prof 2>&1 | grep "^{" | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('side_stream=$1 fork_opt=$2', d['value'], d['ms_per_step'], d['config']['step_mix']['critic_ms'], d['config']['step_mix']['generator_ms'], d['config']['last_distance'])"
done; done > gpurun_out/s17/ab.txt 2>&1
timeout 600 python tools/soak_sinkhorn.py --steps 1500 --batch 64 --out gpurun_out/s17/soak_two.json > gpurun_out/s17/soak.txt 2>&1
OTGAN_SIDE_STREAM=0 timeout 600 python tools/soak_sinkhorn.py --steps 1500 --batch 64 --out gpurun_out/s17/soak_one.json >> gpurun_out/s17/soak.txt 2>&1
python tools/soak_sinkhorn.py --compare gpurun_out/s17/soak_two.json gpurun_out/s17/soak_one.json >> gpurun_out/s17/soak.txt 2>&1
tail -4 gpurun_out/s17/tests.txt; cat gpurun_out/s17/ab.txt; tail -15 gpurun_out/s17/soak.txt | cut -c1-160
