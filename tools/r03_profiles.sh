#!/bin/bash
# Round-3 evidence on one GPU box: PMC + trace passes of both models, the timed window per kernel, the bench line.
#   tools/r03_profiles.sh   -> gpurun_out/r03_*  (copy the summaries to profiles/)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_r02.sh r03 dcgan > /dev/null 2>&1
bash tools/pmc_r02.sh r03 densenet > /dev/null 2>&1
bash tools/window_ab.sh "OTGAN_X3_NARROW=1" > /dev/null 2>&1; cp gpurun_out/window_0.txt gpurun_out/r03_window_stats_dcgan.txt
BENCH_FLAGS="--model densenet" bash tools/window_ab.sh "OTGAN_X3_NARROW=1" > /dev/null 2>&1; cp gpurun_out/window_0.txt gpurun_out/r03_window_stats_densenet.txt
cd $R
python bench.py --steps 24 --warmup 6 > gpurun_out/r03_bench_dcgan.json 2> gpurun_out/r03_bench_dcgan.err
ls -la gpurun_out | grep r03_
