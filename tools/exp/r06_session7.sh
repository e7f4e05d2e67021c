#!/bin/bash
O=gpurun_out/s7; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_matching_gpu.py tests/test_matching_grad_gpu.py tests/test_cfg5_gpu.py tests/test_matching_engine_accuracy_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
python bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s7/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
for c in d['secondary']['matching_block']['cases']:
    print({k:v for k,v in c.items() if k not in('note',)})
PY
bash tools/pmc_matching.sh s7 1024 32768 100 256 rank > /dev/null 2>&1; head -8 gpurun_out/s7_pmc_kernels_matching_N1024_D32768_rows256_rank.txt | cut -c1-200
