# dev: the phase tool on the Winograd-domain GEMM shapes of the DCGAN layers (two-piece build)
cd ${GRAFT_REPO_ROOT:-.}
B=tools/ablate/bin/x3_phase2
for s in ${SHAPES:-"4096 1024 256" "1024 2048 512" "256 4096 1024" "4096 256 1024" "1024 512 2048" "256 1024 4096"}; do
  echo "=== $s"; timeout 120 $B $s 10 32 2>&1 | grep -v "^  workgroup\|^workgroup\|start at\|span"
done
