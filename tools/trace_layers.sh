#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ly_trace -- python $R/tools/bench_layers.py 256 > $R/gpurun_out/ly_trace.log 2>&1
f=$(find $R/gpurun_out/ly_trace -name "*.db" | head -1)
python $R/tools/rocpd_shapes.py $f 40 > $R/gpurun_out/ly_shapes.txt
rm -rf $R/gpurun_out/ly_trace
