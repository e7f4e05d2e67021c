"""Representation error of the split-precision operands of csrc/gemm_x3.h, modelled in numpy
(oracle/split_precision_np.py): products exact, fp64 accumulation, so only the splits are measured."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import split_precision_np as SP  # noqa: E402


def _case(name, K, rng):
    A, B = rng.standard_normal((64, K)), rng.standard_normal((64, K))
    if name == "heavy":          # log-normal magnitudes over ~e^12
        A *= np.exp(2 * rng.standard_normal(A.shape)); B *= np.exp(2 * rng.standard_normal(B.shape))
    if name == "tiny":           # gradients: far below the fp16 range before scaling
        A *= 1e-7; B *= 1e-7
    if name == "outlier":        # one element 3e4 times the rest sets the scale
        A[0, 0] = 3e4
    return A.astype(np.float32), B.astype(np.float32)


@pytest.mark.parametrize("name", ["gauss", "heavy", "tiny", "outlier"])
@pytest.mark.parametrize("K", [256, 2048])
def test_split_errors_below_fp32_accumulation(name, K):
    rng = np.random.default_rng(K + len(name))
    A, B = _case(name, K, rng)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    e3 = SP.rel(SP.dots_bf16x3(A, B), ref)
    e2 = SP.rel(SP.dots_f16x2(A, B), ref)
    e2h = SP.rel(SP.dots_f16x2(A, B, headroom=64.0), ref)      # a Winograd gain bound that is not attained
    e32 = SP.rel((A @ B.T).astype(np.float64), ref)            # what fp32 accumulation alone costs
    assert e3 < 2e-8                       # 24 bits, terms down to 2^-16 kept
    assert e2 < 3e-7 and e2h < 3e-7        # 22 bits; at or below the accumulation error of an fp32 GEMM
    assert e2 < 2 * e32 + 1e-7


def test_scale_never_overflows_fp16():
    rng = np.random.default_rng(1)
    for mag in (1e-30, 1e-8, 1.0, 3e4, 1e20):
        x = (rng.standard_normal(4096) * mag).astype(np.float32)
        h, lo, s = SP.split_f16x2(x)
        assert np.isfinite(h).all() and np.abs(h).max() < 2.0 ** 14 + 1
        assert float(np.log2(s)) == round(float(np.log2(s)))   # a power of two: scaling and un-scaling are exact
    z = np.zeros(16, np.float32)
    assert SP.split_f16x2(z)[2] == 2.0 ** 14                   # all-zero tensor: e = 0
