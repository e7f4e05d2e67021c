#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/s10; mkdir -p $O
export OMP_NUM_THREADS=16
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_grow_buffers_gpu.py tests/test_models_gpu.py tests/test_step_graph_gpu.py tests/test_reference_run_gpu.py tests/test_amax_fused_gpu.py tests/test_train_step_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['config']['step_graph'][:40])"; }
for rep in 1 2; do
  python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet" | tee -a $O/dn.txt
done
cd /tmp && export TMPDIR=/tmp
OTGAN_STEP_GRAPH=0 rocprofv3 --kernel-trace -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --model densenet --nr_sinkhorn_iter 200 --steps 6 --warmup 6 --no_cpu_baseline --no_secondary --no_prof > $O/bench.json 2> $O/err.txt
f=$(find $O/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/debug/absmax_consumers.py $f > $O/absmax_consumers.txt 2>&1
rm -rf $O/trace
cat $O/absmax_consumers.txt
