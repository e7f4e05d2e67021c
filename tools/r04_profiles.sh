#!/bin/bash
# Round-4 evidence on one GPU box: PMC + trace passes of both models, the timed window per kernel, SQ-level counters of the
# growth-layer kernels, PMC of the matching block through the TRAINING-MODE entry (N = 128 / 256 / rank of 8 at N = 1024), the
# bench line.   tools/r04_profiles.sh   -> gpurun_out/r04_*  (copy the summaries to profiles/)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_r02.sh r04 dcgan > /dev/null 2>&1
bash tools/pmc_r02.sh r04 densenet > /dev/null 2>&1
bash tools/window_dcgan.sh > /dev/null 2>&1; cp gpurun_out/window_stats.txt gpurun_out/r04_window_stats_dcgan.txt
bash tools/window_densenet.sh > /dev/null 2>&1; cp gpurun_out/window_stats_densenet.txt gpurun_out/r04_window_stats_densenet.txt
bash tools/pmc_dense.sh r04 > /dev/null 2>&1
bash tools/pmc_matching.sh r04 128 32768 100 0 grad > /dev/null 2>&1
bash tools/pmc_matching.sh r04 256 131072 100 0 grad > /dev/null 2>&1
bash tools/pmc_matching.sh r04 1024 32768 100 256 rank > /dev/null 2>&1
bash tools/pmc_matching.sh r04 1024 7296 200 256 rank > /dev/null 2>&1
cd $R
python bench.py --steps 24 --warmup 6 > gpurun_out/r04_bench_dcgan.json 2> gpurun_out/r04_bench_dcgan.err
ls -la gpurun_out | grep r04_
