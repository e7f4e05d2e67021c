#!/bin/bash
# SQ-level PMC passes of the DenseNet growth-layer kernels (tools/bench_dense.py): what bounds dense16_fwd_* / dgrad / wgrad
# -- LDS issue stalls, VALU work, waits -- next to the MFMA-busy fraction.  usage: tools/pmc_dense.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/${TAG}_dense_sq1 -- python $R/tools/bench_dense.py 256 > $R/gpurun_out/${TAG}_dense_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_dense_sq2 -- python $R/tools/bench_dense.py 256 > $R/gpurun_out/${TAG}_dense_sq2.log 2>&1
a=$(find $R/gpurun_out/${TAG}_dense_sq1 -name "*.db" | head -1)
b=$(find $R/gpurun_out/${TAG}_dense_sq2 -name "*.db" | head -1)
python $R/tools/pmc_sq.py $R/gpurun_out/${TAG}_pmc_sq_dense.txt $a $b --match dense16 | head -150
rm -rf $R/gpurun_out/${TAG}_dense_sq1 $R/gpurun_out/${TAG}_dense_sq2
