"""CPU: the shipped gfx950 code contains no packed fp32 instruction of the form that miscomputes beside the GEMM.

Round 4 root-caused round 3's co-residency miscompute (DESIGN section 3 "Four hazards", item 3; evidence in
profiles/r04_corun_*.txt): v_pk_fma_f32 / v_pk_mul_f32 with OP_SEL set on SRC1 (the low result half reads the HIGH register
of the source pair -- the compiler's code for `acc2 += w2 * x4.y`) returns wrong low halves in lanes 48-63 while a wave with
MFMAs and LDS-DMA loads in flight shares the SIMD; every other packed / scalar form is clean.  The library avoids the form
(no packed fp32 outside the Winograd transforms, whose code only uses op_sel_hi); this test keeps it that way for every
future kernel and compiler."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "ot-gan_amd", "csrc", "libotgan_hip.so")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs ROCm's llvm-objdump")
def test_no_packed_fp32_instruction_selects_the_high_register_of_src1():
    import isa_check
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    forms, bad = isa_check.scan(LIB)
    assert sum(forms.values()) > 1000          # the scan saw the Winograd transforms' packed code at all
    assert not bad, bad[:10]
