#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dc_trace -- python $R/bench.py --steps 6 --warmup 6 --no_cpu_baseline > $R/gpurun_out/dc_trace.json 2> $R/gpurun_out/dc_trace.err
f=$(find $R/gpurun_out/dc_trace -name "*.db" | head -1)
python $R/tools/rocpd_shapes.py $f 60 > $R/gpurun_out/dc_shapes.txt
python $R/tools/rocpd_stats.py $f $R/gpurun_out/dc_kernel_stats.csv > /dev/null
rm -rf $R/gpurun_out/dc_trace
