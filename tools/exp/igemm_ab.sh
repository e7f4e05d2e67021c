#!/bin/bash
# per-kernel times of the implicit-GEMM transitions of a DenseNet step: default library against another build (dev tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" "$1"; do
  rm -rf $R/gpurun_out/ig_trace
  OTGAN_SIDE_STREAM=0 OTGAN_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ig_trace -- python $R/bench.py --model densenet --nr_sinkhorn_iter 200 --steps 6 --warmup 6 --no_secondary --no_cpu_baseline --no_prof > /dev/null 2>&1
  t=$(find $R/gpurun_out/ig_trace -name "*.db" | head -1)
  echo "== lib=[$lib]"
  python $R/tools/rocpd_stats.py $t /tmp/ig.csv > /dev/null
  grep "conv_igemm_kernel" /tmp/ig.csv | cut -c1-200
done
rm -rf $R/gpurun_out/ig_trace
