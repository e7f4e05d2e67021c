#!/bin/bash
O=gpurun_out/s6; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_grow_buffers_gpu.py tests/test_models_gpu.py tests/test_step_graph_gpu.py tests/test_reference_run_gpu.py -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['config']['step_graph'][:40])"; }
for rep in 1 2 3; do
  python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet" | tee -a $O/dn.txt
done
OTGAN_STEP_GRAPH=0 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet eager" | tee -a $O/dn.txt
