#!/bin/bash
# Sinkhorn panel kernel A/B over library builds (dev): tools/exp/sinkhorn_ab.sh lib1 lib2 ...
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" "$@"; do echo "== lib=[$(basename "$lib")]"; OTGAN_LIB_PATH=$lib python $R/tools/bench_matching.py 2>&1 | grep -v amdgpu | grep "rows= all" | grep -v "N=  128" | sed -E 's/.*(N= *[0-9]+ D= *[0-9]+ L=[0-9]+).*sinkhorn +([0-9.]+).*/\1 sinkhorn \2 ms/'; done; done
