#!/bin/bash
mkdir -p gpurun_out/s12
export OMP_NUM_THREADS=16
{ for rep in 1 2; do for v in 1 0; do echo "== OTGAN_COST_XCD=$v"; OTGAN_COST_XCD=$v python tools/exp/rank_time.py 32768 100 2>&1 | grep -v amdgpu; OTGAN_COST_XCD=$v python tools/exp/rank_time.py 7296 200 2>&1 | grep -v amdgpu; done; done; } > gpurun_out/s12/rank.txt 2>&1
timeout 900 python -m pytest tests/test_matching_gpu.py tests/test_matching_grad_gpu.py tests/test_multirank_gpu.py tests/test_cfg5_gpu.py -x -q > gpurun_out/s12/tests.txt 2>&1
cat gpurun_out/s12/rank.txt; tail -3 gpurun_out/s12/tests.txt
