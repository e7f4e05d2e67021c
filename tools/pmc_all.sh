#!/bin/bash
# final-state PMC + trace passes for both model families (GPU box)
R=$GRAFT_REPO_ROOT
for M in dcgan densenet; do
  bash $R/tools/pmc_bench.sh r01f_$M $M > /dev/null 2>&1
  f=$(find $R/gpurun_out/r01f_${M}_fetch -name "*.db" | head -1)
  w=$(find $R/gpurun_out/r01f_${M}_write -name "*.db" | head -1)
  m=$(find $R/gpurun_out/r01f_${M}_mfma -name "*.db" | head -1)
  t=$(find $R/gpurun_out/r01f_${M}_trace -name "*.db" | head -1)
  python $R/tools/traffic_summary.py $f $w $m $R/gpurun_out/r01f_pmc_summary_$M.json > /dev/null
  python $R/tools/rocpd_stats.py $t $R/gpurun_out/r01f_kernel_stats_$M.csv > /dev/null
  python $R/tools/class_stats.py $t > $R/gpurun_out/r01f_class_stats_$M.json
  cp $R/gpurun_out/r01f_${M}_trace.json $R/gpurun_out/r01f_bench_under_rocprof_$M.json
  # keep the merge-back small: the sqlite traces are not needed locally
  rm -rf $R/gpurun_out/r01f_${M}_fetch $R/gpurun_out/r01f_${M}_write $R/gpurun_out/r01f_${M}_mfma $R/gpurun_out/r01f_${M}_trace
done
cat $R/gpurun_out/r01f_pmc_summary_dcgan.json $R/gpurun_out/r01f_pmc_summary_densenet.json
