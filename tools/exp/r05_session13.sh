#!/bin/bash
mkdir -p gpurun_out/s13
for v in 1 0; do echo "== OTGAN_PANEL_XCD=$v"; OTGAN_PANEL_XCD=$v OTGAN_LIB_PATH=tools/debug/bin/libotgan_panel_timing.so python tools/matching_workload.py 1024 7296 200 0 grad 2>&1 | grep "panel timing" | tail -3; done > gpurun_out/s13/panel_timing.txt 2>&1
cat gpurun_out/s13/panel_timing.txt
