#!/bin/bash
mkdir -p gpurun_out/s4
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_step_graph_gpu.py -x -q --durations=10 > gpurun_out/s4/graph_test.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 12 --no_cpu_baseline > gpurun_out/s4/bench.txt 2>&1
OTGAN_STEP_GRAPH=0 timeout 600 python bench.py --steps 30 --warmup 12 --no_cpu_baseline --no_secondary > gpurun_out/s4/bench_eager.txt 2>&1
tail -15 gpurun_out/s4/graph_test.txt
