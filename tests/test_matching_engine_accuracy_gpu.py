"""GPU: what the matching GEMM engines cost on the TRAINING SIGNAL.

The step injects `f_aa - f_ab` and `f_bb - f_ba` as upstream gradients (reference train.py:111,125-126): differences of
matched features, lambda-amplified through the log-kernel (lambda = 500).  The cost and plan-application GEMMs run on two
scaled fp16 pieces per operand at every size (N >= 256 since round 4; the one-tile problems N <= 128 since round 5:
cost128_h2_kernel / plan_apply128_h2_kernel) (22 significand bits, three MFMAs per product); `OTGAN_MATCH_FP32=1` keeps
them on the exact-fp32 MFMA engine (the yardstick: `v_mfma_f32_32x32x2_f32`, what the N = 128 headline problem ran on until round 5).
One process per engine (the library reads the switch once), every case against the fp64 oracle:

    N = 128  / D = 32768  / 100 sweeps   configs[1]
    N = 256  / D = 7296   / 200 sweeps   configs[3] width
    N = 256  / D = 131072 / 100 sweeps   configs[4]
    N = 1024 / D = 32768 and D = 7296 (200 sweeps): the rows of ranks 0 and 5 of eight, configs[2] / configs[3]

Asserted: the default engine's error on both injected differences is <= 1.25 x the fp32 engine's (+ 2e-6: both at rounding
level), and the tolerance of the matching tests for the injected gradients (`conftest.REL_DIFF_INJECTED`, used by
test_matching_grad_gpu.py / test_matching_gpu.py / test_multirank_gpu.py) is within ~3 x of what is measured here -- the numbers are printed into the test log and copied
into DESIGN.md section 3.  (A three-bf16-piece build of the matching GEMMs no longer exists: sinkhorn.hip compiles the
engine with two pieces only since round 4; the 24-bit yardstick is the fp32 engine.)"""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from conftest import REL_DIFF_INJECTED

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

ENGINES = {"fp16x2": {}, "fp32_mfma": {"OTGAN_MATCH_FP32": "1"}}
CASES = ["N128_D32768", "N256_D7296", "N256_D131072", "N1024_D32768_rank", "N1024_D7296_rank"]
# Measured (round 5, DESIGN section 3 "What the matching engines cost on the injected gradients"): the shipped engines
# leave 3.7e-6 ... 6.0e-6 on both differences at every size (N = 128 on the exact-fp32 engine included); the fp32 MFMA
# engine at N = 1024 is the WORST of the lot (2.4e-5 ... 4.2e-5: its fp32 accumulation chains over D are eight times
# longer per K split).  The matching tests' tolerance for the injected gradients (conftest.REL_DIFF_INJECTED) is 3 x the
# shipped engines' worst.
REL_DIFF = REL_DIFF_INJECTED


def _run(refdir, env):
    e = {k: v for k, v in os.environ.items() if k != "OTGAN_MATCH_FP32"}
    e.update(env)
    e.setdefault("OMP_NUM_THREADS", "16")
    r = subprocess.run([sys.executable, os.path.join(HERE, "matching_engine_worker.py"), refdir] + CASES, env=e,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("MATCH_ENGINE_JSON ")][-1]
    return json.loads(line[len("MATCH_ENGINE_JSON "):])


@pytest.fixture(scope="module")
def runs():
    with tempfile.TemporaryDirectory(prefix="otgan_match_ref_") as refdir:
        # the first worker computes the oracle's rows (fp64, CPU) and leaves them in refdir for the second
        return {name: _run(refdir, env) for name, env in ENGINES.items()}


def test_injected_gradients_per_engine(runs):
    d, w = runs["fp16x2"], runs["fp32_mfma"]
    print("\nrelative L2 error of the injected gradients against the fp64 oracle, lambda = 500")
    print(f"{'case':22s} {'fp16x2 gen':>11s} {'fp16x2 dat':>11s} {'fp32 gen':>11s} {'fp32 dat':>11s} {'dist fp16x2':>12s} {'dist fp32':>10s}")
    for c in CASES:
        print(f"{c:22s} {d[c]['grad_gen']:11.2e} {d[c]['grad_dat']:11.2e} {w[c]['grad_gen']:11.2e} {w[c]['grad_dat']:11.2e} "
              f"{d[c]['distance_rel']:12.2e} {w[c]['distance_rel']:10.2e}")
    worst = 0.0
    for c in CASES:
        for k in ("grad_gen", "grad_dat"):
            assert d[c][k] <= 1.25 * w[c][k] + 2e-6, (c, k, d[c][k], w[c][k])
            worst = max(worst, d[c][k])
        for r in (d, w):
            assert r[c]["distance_rel"] < 1e-4, (c, r[c]["distance_rel"])      # the north star's loss tolerance
    # the tolerance of the matching tests has at most ~3 x headroom over the shipped engines' worst case (3.5: box-to-box
    # the worst case moved by a few per cent)
    print(f"worst measured on the shipped engines {worst:.2e}; REL_DIFF of the matching tests {REL_DIFF:.1e}")
    assert worst <= REL_DIFF <= 3.5 * worst, (worst, REL_DIFF)
