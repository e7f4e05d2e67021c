#!/bin/bash
mkdir -p gpurun_out/s6
echo "=== no_prof" > gpurun_out/s6/out.txt
timeout 600 python bench.py --steps 12 --warmup 12 --no_cpu_baseline --no_prof 2>&1 | grep -v amdgpu.ids | tail -c 1800 >> gpurun_out/s6/out.txt
echo "=== prof" >> gpurun_out/s6/out.txt
timeout 600 python bench.py --steps 12 --warmup 12 --no_cpu_baseline 2>&1 | grep -v amdgpu.ids | tail -c 1800 >> gpurun_out/s6/out.txt
cat gpurun_out/s6/out.txt | cut -c1-1500
