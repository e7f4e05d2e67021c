#!/bin/bash
# Round-5 evidence on one GPU box: PMC + trace passes of the DCGAN headline, the DenseNet shape and (new: VERDICT r4 item 7)
# the 64x64 configuration; the timed window per kernel; PMC of the matching block through the training-mode entry and of a
# rank of eight (one split of the features per step, XCD-local Sinkhorn problems); per-layer times; the bench line.
#   tools/r05_profiles.sh   -> gpurun_out/r05_*  (copy the summaries to profiles/)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_r02.sh r05 dcgan > /dev/null 2>&1
bash tools/pmc_r02.sh r05 densenet > /dev/null 2>&1
bash tools/pmc_r02.sh r05 dcgan64 > /dev/null 2>&1
bash tools/window_dcgan.sh > /dev/null 2>&1; cp gpurun_out/window_stats.txt gpurun_out/r05_window_stats_dcgan.txt
bash tools/window_densenet.sh > /dev/null 2>&1; cp gpurun_out/window_stats_densenet.txt gpurun_out/r05_window_stats_densenet.txt
bash tools/pmc_matching.sh r05 128 32768 100 0 grad > /dev/null 2>&1
bash tools/pmc_matching.sh r05 256 131072 100 0 grad > /dev/null 2>&1
bash tools/pmc_matching.sh r05 1024 32768 100 256 rank > /dev/null 2>&1
bash tools/pmc_matching.sh r05 1024 7296 200 256 rank > /dev/null 2>&1
cd $R
python tools/bench_layers.py 256 > gpurun_out/r05_bench_layers.txt 2>&1
python bench.py --steps 24 --warmup 6 > gpurun_out/r05_bench_dcgan.json 2> gpurun_out/r05_bench_dcgan.err
ls -la gpurun_out | grep r05_
