#!/bin/bash
# per (kernel, grid) totals of the DenseNet bench (dev tool, GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/dshape_trace -- env ${SHAPE_ENV:-X=0} python $R/bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary > /dev/null 2> $R/gpurun_out/dshape.err
f=$(find $R/gpurun_out/dshape_trace -name "*.db" | head -1)
python $R/tools/rocpd_shapes.py $f 90 > $R/gpurun_out/shapes_densenet.txt
rm -rf $R/gpurun_out/dshape_trace
