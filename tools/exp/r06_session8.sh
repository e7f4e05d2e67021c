#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OTGAN_STEP_GRAPH=0 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --model densenet --nr_sinkhorn_iter 200 --steps 6 --warmup 6 --no_cpu_baseline --no_secondary --no_prof > $O/bench.json 2> $O/err.txt
f=$(find $O/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/debug/absmax_consumers.py $f > $O/absmax_consumers.txt 2>&1
python $GRAFT_REPO_ROOT/tools/debug/absmax_consumers.py $f slab_reduce > $O/slab_consumers.txt 2>&1
rm -rf $O/trace
cat $O/absmax_consumers.txt; echo; cat $O/slab_consumers.txt | head -20
