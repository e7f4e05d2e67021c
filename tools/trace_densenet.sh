#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dn_trace -- python $R/bench.py --model densenet --steps 6 --warmup 3 --no_cpu_baseline > $R/gpurun_out/dn_trace.json 2> $R/gpurun_out/dn_trace.err
f=$(find $R/gpurun_out/dn_trace -name "*.db" | head -1)
python $R/tools/rocpd_shapes.py $f 70 > $R/gpurun_out/dn_shapes.txt
python $R/tools/class_stats.py $f > $R/gpurun_out/dn_class.txt 2>&1
cat $R/gpurun_out/dn_trace.json
