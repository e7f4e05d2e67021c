#!/bin/bash
mkdir -p gpurun_out/s3
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/s3/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/s3/gpu_tests.txt
timeout 600 python bench.py --steps 30 --warmup 6 --no_cpu_baseline > gpurun_out/s3/bench.txt 2>&1
tail -3 gpurun_out/s3/gpu_tests.txt
