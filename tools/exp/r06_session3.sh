#!/bin/bash
# round 6, session 3: step graphs WITH the second stream captured (parallel branches) -- bit identity, A/B eager / graph / graph+fork
O=gpurun_out/s3; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_step_graph_gpu.py -x -q > $O/graph_tests.txt 2>&1; tail -15 $O/graph_tests.txt
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['config']['step_graph'][:60])"; }
for rep in 1 2; do
  OTGAN_STEP_GRAPH=0 python bench.py --steps 30 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_d0.txt | line "dcgan eager" | tee -a $O/graph_ab.txt
  OTGAN_STEP_GRAPH=1 OTGAN_GRAPH_FORK=0 python bench.py --steps 30 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_d1.txt | line "dcgan graph one-stream" | tee -a $O/graph_ab.txt
  OTGAN_STEP_GRAPH=1 python bench.py --steps 30 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_d2.txt | line "dcgan graph two-stream" | tee -a $O/graph_ab.txt
done
for rep in 1 2; do
  OTGAN_STEP_GRAPH=0 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_n0.txt | line "densenet eager" | tee -a $O/graph_ab.txt
  OTGAN_STEP_GRAPH=1 OTGAN_GRAPH_FORK=0 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_n1.txt | line "densenet graph one-stream" | tee -a $O/graph_ab.txt
  OTGAN_STEP_GRAPH=1 python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 18 --warmup 24 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_n2.txt | line "densenet graph two-stream" | tee -a $O/graph_ab.txt
done
OTGAN_STEP_GRAPH=1 python bench.py --image_size 64 --batch_per_gpu 512 --steps 12 --warmup 18 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_64.txt | line "dcgan64 graph two-stream" | tee -a $O/graph_ab.txt
OTGAN_STEP_GRAPH=0 python bench.py --image_size 64 --batch_per_gpu 512 --steps 12 --warmup 18 --no_cpu_baseline --no_secondary --no_prof 2>$O/err_64e.txt | line "dcgan64 eager" | tee -a $O/graph_ab.txt
grep -h -i "warn\|error\|fail" $O/err_*.txt | sort | uniq -c | head
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_side_stream_gpu.py -x -q > $O/rccl_tests.txt 2>&1; tail -15 $O/rccl_tests.txt
