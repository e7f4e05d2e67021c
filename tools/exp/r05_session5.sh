#!/bin/bash
mkdir -p gpurun_out/s5
for mode in plain keep reset nocache; do echo "=== $mode"; timeout 300 python tools/exp/graph_teardown.py $mode 2>&1 | grep -v amdgpu.ids | tail -12; done > gpurun_out/s5/teardown.txt 2>&1
cat gpurun_out/s5/teardown.txt
