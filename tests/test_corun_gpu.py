"""GPU: kernels of the library running BESIDE the Winograd-domain GEMM on one GPU (second stream of the same process).

Round 3's default GEMM (wino_bgemm_x3n_kernel: 256 x 128 tile, two workgroups per compute unit) is the first one that
leaves room for other workgroups on its compute unit, which is what happens whenever two ranks share a GPU (the
gloo tests) or a collective's kernels overlap the backward pass.  Found the hard way: the RGB-in forward kernel's
packed fp32 FMAs returned wrong values in lanes 48-63 in 60 % of the launches with a GEMM wave on the same SIMD
(it now uses scalar FMAs; round 3's RGB-out input-gradient kernel reproduced the same signature and got the same cure,
and every translation unit except the Winograd transforms is built without packed fp32, csrc/Makefile).  Round 4 found
the mechanism (tools/debug/corun_probe.*: a packed fp32 instruction with OP_SEL on SRC1 is the one
form that returns wrong low halves beside a wave with MFMAs and LDS-DMA loads in flight; tests/test_isa_cpu.py asserts
that the built library contains none); this test keeps every kernel family of a DCGAN / DenseNet step honest as the
neighbour all the same."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("victim", ["rgbin", "rgbin3", "rgbin_grad", "rgbout", "growth", "s2", "up", "matching", "matching256", "adam",
                                    "wn", "glu", "head"])
def test_neighbour_of_the_gemm_computes_the_same(victim):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "corun_repro.py"), victim, "8"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = [l for l in r.stdout.splitlines() if l.startswith("CORUN")][-1]
    assert f"CORUN {victim} mismatching results: 0 of" in last, r.stdout[-2000:]
    assert "GEMM-side mismatches: 0 of" in last, r.stdout[-2000:]      # the GEMM layers are victims too
