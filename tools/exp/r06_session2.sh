#!/bin/bash
# round 6, session 2: (a) step graphs after the ADVICE r5 fix (replay no longer recomputes the critic's operands): bit identity +
# A/B eager/graph, DCGAN and DenseNet; (b) power / clock traces (tools/power_trace.py); (c) multi-rank == single-process error sweep
O=gpurun_out/s2; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_step_graph_gpu.py -x -q > $O/graph_tests.txt 2>&1; tail -3 $O/graph_tests.txt
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for g in 0 1; do
    OTGAN_STEP_GRAPH=$g python bench.py --steps 30 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "dcgan graph=$g" | tee -a $O/graph_ab.txt
  done
done
for rep in 1 2; do
  for g in 0 1; do
    OTGAN_STEP_GRAPH=$g python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>/dev/null | line "densenet graph=$g" | tee -a $O/graph_ab.txt
  done
done
# (b)
python tools/power_trace.py $O --tag two_piece > $O/power_two_piece.log 2>&1
OTGAN_WINO_PIECES=3 python tools/power_trace.py $O --tag three_piece --phases gemm_real,step > $O/power_three_piece.log 2>&1
OTGAN_SIDE_STREAM=0 python tools/power_trace.py $O --tag one_stream --phases step > $O/power_one_stream.log 2>&1
tail -1 $O/power_two_piece.log | cut -c1-1500
# (c)
for seed in 5 6 7 8; do
  OTGAN_TEST_DIST_SEED=$seed timeout 900 python -m pytest tests/test_dist_gpu.py -q -s -k "equal_single_process" 2>&1 | grep -a "dist-tolerance\|passed\|failed" >> $O/dist_tolerance.txt
done
cat $O/dist_tolerance.txt | cut -c1-200
