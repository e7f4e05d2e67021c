#!/bin/bash
# HBM-traffic PMC passes + kernel trace of bench.py (GPU box).  Separate passes for FETCH_SIZE
# and WRITE_SIZE (TCC slot limits); no sys/hip/hsa trace domains together with --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
MODEL=${2:-dcgan}
ARGS="--steps 6 --warmup 6 --no_cpu_baseline --no_secondary --model $MODEL"
# "dcgan64": BASELINE configs[4] shape on one GPU (64x64 images, 512 per GPU = two halves of 256, D = 131072)
[ "$MODEL" = "dcgan64" ] && ARGS="--steps 6 --warmup 6 --no_cpu_baseline --no_secondary --model dcgan --image_size 64 --batch_per_gpu 512"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace -- python $R/bench.py $ARGS > $R/gpurun_out/${TAG}_trace.json 2> $R/gpurun_out/${TAG}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py $ARGS --no_prof > /dev/null 2> $R/gpurun_out/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_write -- python $R/bench.py $ARGS --no_prof > /dev/null 2> $R/gpurun_out/${TAG}_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/${TAG}_mfma -- python $R/bench.py $ARGS --no_prof > /dev/null 2> $R/gpurun_out/${TAG}_mfma.err
ls $R/gpurun_out/${TAG}_*/*/ 
