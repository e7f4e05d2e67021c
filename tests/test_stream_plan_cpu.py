"""Work partition of the persistent ("stream") Winograd-domain GEMM, checked on the host.

tools/stream_plan_check.hip plans ~2300 shapes (plain, strided forward / dgrad / wgrad; 1..38 workgroups per XCD)
with the planner winograd.hip uses and walks every workgroup's range with the function the kernel itself walks it
with (both live in csrc/gemm_x3.h): pieces are even, >= 4 stages, inside their tiles; every (frequency, tile) is
covered exactly once in workgroup order; a workgroup parks at most its first piece.  No GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_stream_plan_invariants(tmp_path):
    exe = tmp_path / "stream_plan_check"
    subprocess.run([HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "ot-gan_amd", "csrc"),
                    os.path.join(ROOT, "tools", "stream_plan_check.hip"), "-o", str(exe)],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    last = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 0 and last.startswith("ok "), out.stdout[-2000:]
    # the shapes of the DCGAN layers at 256 images (forward, dgrad, wgrad of the strided critic layers included)
    dcgan = ("4096 1024 256 0 0 0 32  1024 2048 512 0 0 0 32  256 4096 1024 0 0 0 32  4096 256 1024 1 256 0 32 "
             "1024 512 2048 1 512 0 32  256 1024 4096 1 1024 0 32  4096 1024 256 2 256 5 32  1024 2048 512 2 512 5 32 "
             "1024 256 4096 3 256 5 32  2048 512 1024 3 512 5 32").split()
    out = subprocess.run([str(exe)] + dcgan, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok 10 shapes"), out.stdout[-2000:]
