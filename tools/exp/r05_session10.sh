#!/bin/bash
mkdir -p gpurun_out/s10
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_dist_gpu.py tests/test_sinkhorn_soak_gpu.py -x -q > gpurun_out/s10/tests.txt 2>&1
timeout 600 python bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof > gpurun_out/s10/bench.txt 2>&1
tail -5 gpurun_out/s10/tests.txt
python - <<'PY'
import json
l=[x for x in open('gpurun_out/s10/bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['ms_per_step'])
    for c in d['secondary']['matching_block']['cases']: print({k:v for k,v in c.items() if k!='note'})
else: print(open('gpurun_out/s10/bench.txt').read()[-1500:])
PY
