#!/bin/bash
# Round-6 evidence on one GPU box (the code after the two-stream schedule, the one-tile matching kernels on the fp16
# pipe and the in-place dense blocks): PMC + trace passes of the DCGAN headline, the DenseNet shape and the 64x64 configuration;
# the timed window per kernel; the kernel stats of the headline with the second stream OFF (what `roofline.avg_ms` must agree
# with: bench.py measures the roofline in a one-stream pass); PMC of the matching block (N = 128 training-mode entry, a rank
# of eight); per-layer times; the bench line.
#   tools/r06_profiles.sh   -> gpurun_out/r06_*  (copy the summaries to profiles/)
R=$GRAFT_REPO_ROOT
cd $R
# (DenseNet's PMC passes run eagerly -- OTGAN_STEP_GRAPH=0 -- so that every kernel is a stream launch the counters can bracket)
# (per-kernel figures with the second stream OFF: every kernel alone on the device, durations comparable with rounds 1 - 4 and
# with `roofline.avg_ms`; the timed windows below show the default two-stream schedule)
OTGAN_SIDE_STREAM=0 bash tools/pmc_r02.sh r06 dcgan > /dev/null 2>&1
OTGAN_SIDE_STREAM=0 OTGAN_STEP_GRAPH=0 bash tools/pmc_r02.sh r06 densenet > /dev/null 2>&1
OTGAN_SIDE_STREAM=0 bash tools/pmc_r02.sh r06 dcgan64 > /dev/null 2>&1
bash tools/window_dcgan.sh > /dev/null 2>&1; cp gpurun_out/window_stats.txt gpurun_out/r06_window_stats_dcgan.txt
bash tools/window_densenet.sh > /dev/null 2>&1; cp gpurun_out/window_stats_densenet.txt gpurun_out/r06_window_stats_densenet.txt
( cd /tmp && export TMPDIR=/tmp && OTGAN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_one_trace -- python $R/bench.py --steps 6 --warmup 6 --no_cpu_baseline --no_secondary > $R/gpurun_out/r06_bench_under_rocprof_dcgan_one_stream.json 2> /dev/null
  t=$(find $R/gpurun_out/r06_one_trace -name "*.db" | head -1); python $R/tools/rocpd_stats.py $t $R/gpurun_out/r06_kernel_stats_dcgan_one_stream.csv > /dev/null; rm -rf $R/gpurun_out/r06_one_trace )
bash tools/pmc_matching.sh r06 128 32768 100 0 grad > /dev/null 2>&1
bash tools/pmc_matching.sh r06 1024 32768 100 256 rank > /dev/null 2>&1
cd $R
python tools/bench_layers.py 256 > gpurun_out/r06_bench_layers.txt 2>&1
python bench.py --steps 24 --warmup 6 > gpurun_out/r06_bench_dcgan.json 2> gpurun_out/r06_bench_dcgan.err
ls -la gpurun_out | grep r06_
