#!/bin/bash
# round 6, session 1: batch slabs (OTGAN_WINO_SLAB_MB) -- bit identity, per-layer times, bench A/B (baseline first and last)
O=gpurun_out/s1; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_gemm_engines_gpu.py -x -q -k "slab or narrow" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for mb in 0 160 80 240; do
  OTGAN_WINO_SLAB_MB=$mb python tools/bench_layers.py 256 > $O/layers_$mb.txt 2>&1
done
for mb in 0 160 80 240 0; do
  OTGAN_WINO_SLAB_MB=$mb python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab_mb=$mb', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))" | tee -a $O/bench_ab.txt
done
for mb in 0 160; do
  OTGAN_WINO_SLAB_MB=$mb OTGAN_SIDE_STREAM=0 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one-stream slab_mb=$mb', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))" | tee -a $O/bench_ab.txt
done
paste $O/layers_0.txt $O/layers_160.txt | cut -c1-260
