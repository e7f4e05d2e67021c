#!/bin/bash
# per-kernel totals of the timed steps of the DCGAN bench (dev tool, GPU box): 12 steps at ~13.3 ms = 160 ms; window 10 steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/win_trace -- python $R/bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary > $R/gpurun_out/win_trace.json 2> $R/gpurun_out/win_trace.err
f=$(find $R/gpurun_out/win_trace -name "*.db" | head -1)
ms=$(python -c "import json; print(json.loads(open('$R/gpurun_out/win_trace.json').read().strip().splitlines()[-1])['ms_per_step'])")
python $R/tools/rocpd_window_stats.py $f $(python -c "print(12*$ms)") 12 > $R/gpurun_out/window_stats.txt
rm -rf $R/gpurun_out/win_trace
