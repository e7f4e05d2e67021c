#!/bin/bash
# round 5, first GPU session: prototype measurements
mkdir -p gpurun_out/s1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export OMP_NUM_THREADS=16
{
for b in fused_out_proto_f9_e1 fused_out_proto_f8_e1 fused_out_proto_f8_e0 fused_out_proto_f9_e0; do
  for shape in "4096 1024 256" "1024 2048 512" "256 4096 1024" "4096 256 1024"; do
    timeout 120 tools/ablate/bin/$b $shape
  done
done
} > gpurun_out/s1/fused_proto.txt 2>&1
timeout 300 python tools/bench_layers.py 256 > gpurun_out/s1/bench_layers.txt 2>&1
timeout 300 python tools/exp/critic_batch.py dcgan > gpurun_out/s1/critic_batch.txt 2>&1
timeout 300 python tools/exp/critic_batch.py densenet >> gpurun_out/s1/critic_batch.txt 2>&1
timeout 1500 python -m pytest tests/test_matching_engine_accuracy_gpu.py -x -q -s > gpurun_out/s1/match_engine.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 6 --no_cpu_baseline > gpurun_out/s1/bench.txt 2>&1
tail -5 gpurun_out/s1/fused_proto.txt gpurun_out/s1/critic_batch.txt gpurun_out/s1/match_engine.txt
