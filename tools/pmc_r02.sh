#!/bin/bash
# Round-2 PMC + trace passes of bench.py (GPU box): kernel trace, FETCH_SIZE, WRITE_SIZE and MFMA-busy passes
# (separate --pmc runs, kernel-trace only), summarised per kernel class (traffic_summary.py) AND per kernel name
# (pmc_kernels.py: the Winograd transform kernels get their own rows).   usage: tools/pmc_r02.sh <tag> <model> [bench args]
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}; M=${2:-dcgan}; shift; shift
bash $R/tools/pmc_bench.sh ${TAG}_$M $M "$@" > /dev/null 2>&1
f=$(find $R/gpurun_out/${TAG}_${M}_fetch -name "*.db" | head -1)
w=$(find $R/gpurun_out/${TAG}_${M}_write -name "*.db" | head -1)
m=$(find $R/gpurun_out/${TAG}_${M}_mfma -name "*.db" | head -1)
t=$(find $R/gpurun_out/${TAG}_${M}_trace -name "*.db" | head -1)
python $R/tools/traffic_summary.py $f $w $m $R/gpurun_out/${TAG}_pmc_summary_$M.json > /dev/null
python $R/tools/pmc_kernels.py $f $w $m $t $R/gpurun_out/${TAG}_pmc_kernels_$M.json 0.05 > $R/gpurun_out/${TAG}_pmc_kernels_$M.txt
python $R/tools/rocpd_stats.py $t $R/gpurun_out/${TAG}_kernel_stats_$M.csv > /dev/null
python $R/tools/class_stats.py $t > $R/gpurun_out/${TAG}_class_stats_$M.json
cp $R/gpurun_out/${TAG}_${M}_trace.json $R/gpurun_out/${TAG}_bench_under_rocprof_$M.json
rm -rf $R/gpurun_out/${TAG}_${M}_fetch $R/gpurun_out/${TAG}_${M}_write $R/gpurun_out/${TAG}_${M}_mfma $R/gpurun_out/${TAG}_${M}_trace
head -40 $R/gpurun_out/${TAG}_pmc_kernels_$M.txt
