#!/bin/bash
mkdir -p gpurun_out/s11
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests/test_matching_gpu.py tests/test_matching_grad_gpu.py tests/test_multirank_gpu.py tests/test_sinkhorn_modes_gpu.py tests/test_cfg5_gpu.py tests/test_corun_gpu.py -x -q > gpurun_out/s11/tests.txt 2>&1
for v in 1 0; do
OTGAN_PANEL_XCD=$v timeout 300 python tools/bench_matching.py > gpurun_out/s11/bm_$v.txt 2>&1
done
tail -5 gpurun_out/s11/tests.txt; tail -25 gpurun_out/s11/bm_1.txt; echo ----; tail -25 gpurun_out/s11/bm_0.txt
