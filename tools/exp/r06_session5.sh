#!/bin/bash
O=gpurun_out/s5; mkdir -p $O
export OMP_NUM_THREADS=16
OTGAN_STEP_GRAPH=0 python tools/debug/absmax_sites.py densenet > $O/absmax_sites_densenet.txt 2>&1
OTGAN_STEP_GRAPH=0 python tools/debug/glue_profile.py densenet > $O/glue_densenet.txt 2>&1
cat $O/absmax_sites_densenet.txt | tail -30; cat $O/glue_densenet.txt | tail -50 | cut -c1-260
