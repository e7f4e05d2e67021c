#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, separate runs, kernel-trace only) of the matching block at one
# problem size -> gpurun_out/<tag>_pmc_kernels_matching_N<N>_D<D>.{json,txt}.   usage: tools/pmc_matching.sh <tag> N D L [rows|0] [arrays|grad|rank]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; N=$2; D=$3; L=$4; ROWS=$5; MODE=${6:-arrays}
W="python $R/tools/matching_workload.py $N $D $L ${ROWS:-0} $MODE"
[ "$ROWS" = "0" ] && ROWS=""
O=$R/gpurun_out/${TAG}_m
rocprofv3 --kernel-trace --stats -d ${O}_trace -- $W > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${O}_fetch -- $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${O}_write -- $W > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d ${O}_mfma -- $W > /dev/null 2>&1
f=$(find ${O}_fetch -name "*.db" | head -1); w=$(find ${O}_write -name "*.db" | head -1)
m=$(find ${O}_mfma -name "*.db" | head -1); t=$(find ${O}_trace -name "*.db" | head -1)
S=N${N}_D${D}${ROWS:+_rows$ROWS}_$MODE
python $R/tools/pmc_kernels.py $f $w $m $t $R/gpurun_out/${TAG}_pmc_kernels_matching_$S.json 0.0 > $R/gpurun_out/${TAG}_pmc_kernels_matching_$S.txt
rm -rf ${O}_trace ${O}_fetch ${O}_write ${O}_mfma
grep -v "at::\|rocclr\|elementwise\|Cat" $R/gpurun_out/${TAG}_pmc_kernels_matching_$S.txt | head -14
