#!/bin/bash
# per-kernel totals of the timed steps of the DenseNet bench (dev tool, GPU box): 12 timed steps = two 5:1 cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/dwin_trace -- python $R/bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary > $R/gpurun_out/dwin_trace.json 2> $R/gpurun_out/dwin_trace.err
f=$(find $R/gpurun_out/dwin_trace -name "*.db" | head -1)
ms=$(python -c "import json; print(json.loads(open('$R/gpurun_out/dwin_trace.json').read().strip().splitlines()[-1])['ms_per_step'])")
python $R/tools/rocpd_window_stats.py $f $(python -c "print(12*$ms)") 12 > $R/gpurun_out/window_stats_densenet.txt
python $R/tools/rocpd_gaps.py $f 30 130 > $R/gpurun_out/gaps_densenet.txt 2>&1
rm -rf $R/gpurun_out/dwin_trace
