#!/bin/bash
# dev: libotgan_hip.so with phase stamps in sinkhorn_panel_kernel (-DPANEL_TIMING: thread 0 of workgroup 0 prints where its
# linear row half-sweeps spent their time) -> tools/debug/bin/libotgan_panel_timing.so; use with OTGAN_LIB_PATH.
set -e
cd "$(dirname "$0")/../../ot-gan_amd/csrc"
mkdir -p ../../tools/debug/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DPANEL_TIMING -c sinkhorn.hip -o /tmp/sinkhorn_timing.o 2> >(grep -v "is not a recognized feature" >&2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/debug/bin/libotgan_panel_timing.so runtime.o /tmp/sinkhorn_timing.o conv.o pointwise.o dense16.o winograd_p2.o winograd_p3.o
ls -la ../../tools/debug/bin/libotgan_panel_timing.so
