#!/bin/bash
mkdir -p gpurun_out/s7
export OMP_NUM_THREADS=16
timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/s7/bench.txt 2>&1
timeout 600 python tools/soak_sinkhorn.py --steps 2000 --batch 64 --out gpurun_out/s7/soak_default.json > gpurun_out/s7/soak.txt 2>&1
OTGAN_SINKHORN_LINEAR=0 timeout 600 python tools/soak_sinkhorn.py --steps 2000 --batch 64 --out gpurun_out/s7/soak_log_only.json >> gpurun_out/s7/soak.txt 2>&1
timeout 600 python tools/soak_sinkhorn.py --steps 2000 --batch 64 --eager --out gpurun_out/s7/soak_default_eager.json >> gpurun_out/s7/soak.txt 2>&1
python tools/soak_sinkhorn.py --compare gpurun_out/s7/soak_default.json gpurun_out/s7/soak_log_only.json >> gpurun_out/s7/soak.txt 2>&1
python tools/soak_sinkhorn.py --compare gpurun_out/s7/soak_default.json gpurun_out/s7/soak_default_eager.json >> gpurun_out/s7/soak.txt 2>&1
timeout 900 python -m pytest tests/test_sinkhorn_soak_gpu.py tests/test_step_graph_gpu.py tests/test_train_main_gpu.py tests/test_train_step_gpu.py -x -q -s > gpurun_out/s7/tests.txt 2>&1
tail -5 gpurun_out/s7/tests.txt; grep -v amdgpu gpurun_out/s7/soak.txt | cut -c1-600
