#!/bin/bash
# dev: libotgan_hip.so variants whose 256 x 128 GEMM lacks one ingredient (its results are garbage; only its presence as a
# neighbour matters) + the probe library.  Output: tools/debug/bin/ (git-ignored, travels with gpurun).
#   nodma   no global_load_lds (LDS-DMA) issued        nomfma   no matrix instructions        noread   no fragment ds_reads
set -e
cd "$(dirname "$0")/../.."
C=ot-gan_amd/csrc
B=tools/debug/bin
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value"
make -s -C $C
for v in NODMA NOMFMA NOREAD; do
  l=$(echo $v | tr A-Z a-z)
  /opt/rocm/bin/hipcc $FLAGS -DX3_PIECES=2 -DWINO_NS=wino_p2 -DX3N_DBG_$v -c $C/winograd.hip -o $B/winograd_p2_$l.o 2>/dev/null &
done
/opt/rocm/bin/hipcc $FLAGS -shared tools/debug/corun_probe.hip -o $B/libcorun_probe.so 2>/dev/null &
wait
for l in nodma nomfma noread; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libotgan_hip_$l.so $C/runtime.o $C/sinkhorn.o $C/conv.o $C/pointwise.o $C/dense16.o $B/winograd_p2_$l.o $C/winograd_p3.o
done
ls -la $B/*.so
