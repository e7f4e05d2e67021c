#!/bin/bash
# round 6, session 4: the whole GPU suite after the graph / collectives / switch changes + the multi-rank error sweep with flip counts
O=gpurun_out/s4; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/gpu_tests.txt 2>&1; tail -30 $O/gpu_tests.txt
for seed in 5 6 7 8 9 10; do
  OTGAN_TEST_DIST_SEED=$seed timeout 900 python -m pytest tests/test_dist_gpu.py -q -s -k "equal_single_process" 2>&1 | grep -a "dist-tolerance\|passed\|failed\|Error" >> $O/dist_tolerance.txt
done
cut -c1-230 $O/dist_tolerance.txt
python bench.py --steps 30 --warmup 12 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
