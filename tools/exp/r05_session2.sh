#!/bin/bash
mkdir -p gpurun_out/s2
export OMP_NUM_THREADS=16
timeout 1700 python -m pytest tests/test_matching_engine_accuracy_gpu.py tests/test_matching_grad_gpu.py tests/test_matching_gpu.py tests/test_multirank_gpu.py tests/test_cfg5_gpu.py -q -s --durations=15 > gpurun_out/s2/matching.txt 2>&1
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_bench_launch_gpu.py -q --durations=10 > gpurun_out/s2/dist.txt 2>&1
timeout 600 python -m pytest tests/test_layers_gpu.py -q -k "dense_block" --durations=10 > gpurun_out/s2/layers.txt 2>&1
tail -4 gpurun_out/s2/matching.txt gpurun_out/s2/dist.txt gpurun_out/s2/layers.txt
