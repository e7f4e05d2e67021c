"""GPU: what the Winograd transform and the operand representation cost in accuracy, measured on the device on a REAL
step.  The well-conditioned whole step (ELU, lambda = 20: no CReLU sign flips, no lambda-amplified
cancellation) runs under four GEMM engines, one process each:
    default             Winograd F(4x4,3x3), operands as two scaled fp16 pieces (22 bits), three MFMAs per product
    OTGAN_WINO_PIECES=3 the same transform, three bf16 pieces (24 bits), six MFMAs
    OTGAN_WINO_FP32=1   the same transform on the fp32 MFMA instruction (the fp32 yardstick of the transform)
    OTGAN_DISABLE_WINOGRAD=1  no Winograd: direct implicit GEMMs (the yardstick of "a plain fp32 convolution")
and every gradient tensor is compared with the fp64 oracle step.  Asserted: the default engine is no worse than 1.25 x
the fp32-Winograd engine per tensor (+1e-6: tensors where both are at rounding level), i.e. the fp16 pieces cost
nothing measurable; the price of the Winograd transform itself against direct convolution is recorded and bounded."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

ENGINES = {"default": {}, "bf16x3": {"OTGAN_WINO_PIECES": "3"}, "fp32_winograd": {"OTGAN_WINO_FP32": "1"},
           "direct": {"OTGAN_DISABLE_WINOGRAD": "1"}}


def _run(env, size=32, seed=5):
    e = {k: v for k, v in os.environ.items() if not k.startswith("OTGAN_WINO") and k != "OTGAN_DISABLE_WINOGRAD"}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "engine_step_worker.py"), str(size), str(seed)], env=e,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("ENGINE_JSON ")][-1]
    return json.loads(line[len("ENGINE_JSON "):])


SEEDS = (5, 6, 7)


@pytest.fixture(scope="module")
def runs():
    """engine -> kind -> tensor -> MEDIAN over three seeds of the relative error.  The median, because the critic's
    feature head is a CReLU (models/dcgan.py:16,19): one of its ~1e5 pre-activations landing on the other side of zero
    than in fp64 moves every gradient of that step by 2e-4 .. 5e-4, whichever engine computed it (seed 5: the direct and
    the three-piece engine flip the same unit, the other two do not) -- a coin flip, not a property of an engine."""
    import statistics
    from concurrent.futures import ThreadPoolExecutor
    # twelve worker processes, most of their time in the fp64 CPU oracle: four at a time (round 4: 129 s -> a third)
    jobs = [(name, s) for name in ENGINES for s in SEEDS]
    with ThreadPoolExecutor(max_workers=4) as pool:
        res = dict(zip(jobs, pool.map(lambda j: _run(dict(ENGINES[j[0]], OMP_NUM_THREADS="8"), seed=j[1]), jobs)))
    out = {}
    for name, env in ENGINES.items():
        per_seed = [res[(name, s)] for s in SEEDS]
        out[name] = {kind: {n: statistics.median(r[kind][n] for r in per_seed) for n in per_seed[0][kind]}
                     for kind in ("disc", "gen")}
        out[name]["dist"] = {kind: max(r["dist"][kind] for r in per_seed) for kind in ("disc", "gen")}
        out[name]["worst_single"] = {kind: max(max(r[kind].values()) for r in per_seed) for kind in ("disc", "gen")}
    return out


@pytest.mark.parametrize("kind", ["disc", "gen"])
def test_default_engine_is_no_worse_than_fp32_winograd(runs, kind):
    d, w, b3, direct = (runs[k][kind] for k in ("default", "fp32_winograd", "bf16x3", "direct"))
    print(f"\n{kind} step, rel. L2 error of every gradient tensor against the fp64 oracle step (median of {len(SEEDS)} seeds)")
    print(f"{'tensor':32s} {'fp16x2':>10s} {'bf16x3':>10s} {'fp32 wino':>10s} {'direct':>10s}")
    for n in d:
        print(f"{n:32s} {d[n]:10.2e} {b3[n]:10.2e} {w[n]:10.2e} {direct[n]:10.2e}")
    print("worst single (seed, tensor):", {k: f"{runs[k]['worst_single'][kind]:.2e}" for k in runs})
    for n in d:
        assert d[n] <= 1.25 * w[n] + 1e-6, (n, d[n], w[n])      # the shipped engine against the fp32 yardstick of its transform
        assert b3[n] <= 2.0 * w[n] + 2e-6, (n, b3[n], w[n])     # (the 24-bit reference build: same class)
    # the price of the Winograd transform itself (any operand representation) against direct fp32 convolution is
    # recorded above (measured: 1.0 - 1.6 x on the weight tensors, < 1 on the biases) and bounded here, so that a
    # regression of the transforms shows
    worst_ratio = max(d[n] / max(direct[n], 1e-7) for n in d)
    assert worst_ratio < 3.0, worst_ratio
    for k in ("default", "bf16x3", "fp32_winograd", "direct"):
        assert runs[k]["dist"][kind] < 1e-4, (k, runs[k]["dist"][kind])
        assert runs[k]["worst_single"][kind] < 3e-3, (k, runs[k]["worst_single"][kind])   # (a flipped head unit: bounded)
