#!/bin/bash
# PMC counters for the conv kernels (separate passes; no sys-trace domains) -- dev tool, GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; fi
  if [ $pass = 2 ]; then C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAVES"; fi
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc$pass -- python $R/tools/bench_layers.py 256 > $R/gpurun_out/pmc$pass.log 2>&1
done
ls -R $R/gpurun_out/pmc1 | head
