#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/s13; mkdir -p $O
export OMP_NUM_THREADS=16
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dense16_h2_gpu.py -x -q > $O/t1.txt 2>&1; tail -12 $O/t1.txt
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_models_gpu.py tests/test_step_graph_gpu.py tests/test_reference_run_gpu.py tests/test_train_step_gpu.py tests/test_grow_buffers_gpu.py -x -q > $O/t2.txt 2>&1; tail -4 $O/t2.txt
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  OTGAN_DENSE16_CHAIN=0 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet per-layer chain" | tee -a $O/dn.txt
  python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet one-launch chain" | tee -a $O/dn.txt
done
