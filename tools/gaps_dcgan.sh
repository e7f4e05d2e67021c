#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/gap_trace -- python $R/bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary > $R/gpurun_out/gap_trace.json 2> $R/gpurun_out/gap_trace.err
f=$(find $R/gpurun_out/gap_trace -name "*.db" | head -1)
python $R/tools/rocpd_gaps.py $f 30 130 > $R/gpurun_out/gaps.txt
rm -rf $R/gpurun_out/gap_trace
