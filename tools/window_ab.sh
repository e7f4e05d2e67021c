#!/bin/bash
# dev: per-kernel totals of the timed DCGAN steps under two environment settings on one box
#   tools/window_ab.sh "OTGAN_FUSED_AMAX=0" "OTGAN_FUSED_AMAX=1"  -> gpurun_out/window_<i>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for setting in "$@"; do
  rm -rf $R/gpurun_out/win_trace
  env $setting rocprofv3 --kernel-trace -d $R/gpurun_out/win_trace -- python $R/bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary ${BENCH_FLAGS} > $R/gpurun_out/win_trace.json 2> $R/gpurun_out/win_trace.err
  f=$(find $R/gpurun_out/win_trace -name "*.db" | head -1)
  ms=$(python -c "import json; print(json.loads(open('$R/gpurun_out/win_trace.json').read().strip().splitlines()[-1])['ms_per_step'])")
  echo "# $setting   ${ms} ms/step under rocprofv3 --kernel-trace" > $R/gpurun_out/window_$i.txt
  python $R/tools/rocpd_window_stats.py $f $(python -c "print(12*$ms)") 12 >> $R/gpurun_out/window_$i.txt
  i=$((i+1))
done
rm -rf $R/gpurun_out/win_trace
