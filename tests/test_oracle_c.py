"""CPU: the plain-C oracle (fp32) against the reference-generated golden vectors (fp64).
Tolerances are fp32-roundoff class amplified by lambda."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import sinkhorn_c as C


def _flat(x):
    return np.concatenate(list(x), 0).astype(np.float32)


def test_c_two_batch(list_case):
    g = list_case
    fa, fb = _flat(g["fa"]), _flat(g["fb"])
    aa, bb, ab, ba, ent, dist = C.two_batch(fa, fb, float(g["lam"]), int(g["iters"]))
    S, B, D = g["fa"].shape
    for got, k in ((aa, "aa"), (bb, "bb"), (ab, "ab"), (ba, "ba")):
        ref = g["two_" + k].reshape(S * B, D)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 2e-3, (k, err)
    assert ent == pytest.approx(float(g["two_entropy"]), rel=2e-3)
    assert dist == pytest.approx(float(g["two_distance"]), rel=5e-3, abs=1e-6)


def test_c_single_batch(list_case):
    g = list_case
    fa, fb = _flat(g["fa"]), _flat(g["fb"])
    aa, bb, ab, ba, ent, dist = C.single_batch(fa, fb, float(g["lam"]), int(g["iters"]))
    S, B, D = g["fa"].shape
    for got, k in ((aa, "aa"), (bb, "bb"), (ab, "ab"), (ba, "ba")):
        ref = g["single_" + k].reshape(S * B, D)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 2e-3, (k, err)
    assert ent == pytest.approx(float(g["single_entropy"]), rel=2e-3)
    assert dist == pytest.approx(float(g["single_distance"]), rel=5e-3, abs=1e-6)


@pytest.mark.parametrize("name", ["survey", "toy_gauss2d_b64"])
def test_c_toy(name):
    g = load_golden(name)
    a, b = g["toy_a"].astype(np.float32), g["toy_b"].astype(np.float32)
    aa, bb, ab, ba, ent, dist = C.two_batch(a, b, float(g["toy_lam"]), int(g["toy_iters"]),
                                           cost="sqeuclid_mean")
    for got, k in ((aa, "aa"), (bb, "bb"), (ab, "ab"), (ba, "ba")):
        ref = g["toy_" + k]
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-4, k
    assert ent == pytest.approx(float(g["toy_entropy"]), rel=1e-4)
    assert dist == pytest.approx(float(g["toy_distance"]), rel=1e-3)
