#!/bin/bash
mkdir -p gpurun_out/s19
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_rccl_gpu.py tests/test_bench_launch_gpu.py tests/test_side_stream_gpu.py tests/test_train_step_gpu.py tests/test_train_main_gpu.py -x -q --durations=8 > gpurun_out/s19/tests.txt 2>&1
tail -14 gpurun_out/s19/tests.txt
