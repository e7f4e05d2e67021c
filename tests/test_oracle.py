"""CPU: the oracle (oracle/matching_np.py) against the golden vectors that were generated
from the reference's own matching code (oracle/make_golden.py), plus the sanity values
recorded in SURVEY.md section 8c and the algebraic properties of section 3.4."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import matching_np as M


def _lists(g):
    return list(g["fa"]), list(g["fb"])


def test_survey_sanity_values():
    # SURVEY.md section 8c, fp64
    g = load_golden("survey")
    assert float(g["two_entropy"]) == pytest.approx(0.312991720506, abs=1e-11)
    assert float(g["two_distance"]) == pytest.approx(0.00692139257042, abs=1e-13)
    assert float(g["single_entropy"]) == pytest.approx(0.427481100387, abs=1e-11)
    assert float(g["single_distance"]) == pytest.approx(0.00416410837530, abs=1e-13)
    assert float(g["random_distance"]) == pytest.approx(0.0198237253905, abs=1e-12)
    assert float(g["toy_entropy"]) == pytest.approx(0.701236515962, abs=1e-11)
    assert float(g["toy_distance"]) == pytest.approx(0.0219625198030, abs=1e-12)


def test_two_batch_matches_reference(list_case):
    g = list_case
    fa, fb = _lists(g)
    out = M.get_matched_features(fa, fb, float(g["lam"]), int(g["iters"]))
    for k, got in zip(("aa", "bb", "ab", "ba"), out[:4]):
        np.testing.assert_allclose(np.stack(got), g["two_" + k], rtol=1e-10, atol=1e-12)
    assert out[4] == pytest.approx(float(g["two_entropy"]), rel=1e-10)
    d = M.calc_distance(fa, fb, out)
    assert d == pytest.approx(float(g["two_distance"]), rel=1e-9, abs=1e-14)


def test_single_batch_matches_reference(list_case):
    g = list_case
    fa, fb = _lists(g)
    out = M.get_matched_features_single_batch(fa, fb, float(g["lam"]), int(g["iters"]))
    for k, got in zip(("aa", "bb", "ab", "ba"), out[:4]):
        np.testing.assert_allclose(np.stack(got), g["single_" + k], rtol=1e-10, atol=1e-12)
    assert out[4] == pytest.approx(float(g["single_entropy"]), rel=1e-10)
    d = M.calc_distance(fa, fb, out)
    assert d == pytest.approx(float(g["single_distance"]), rel=1e-9, abs=1e-14)


def test_random_matches_reference(list_case):
    g = list_case
    fa, fb = _lists(g)
    out = M.get_matched_features_random(fa, fb)
    assert out[4] == 0.0
    d = M.calc_distance(fa, fb, out)
    assert d == pytest.approx(float(g["random_distance"]), rel=1e-10)


@pytest.mark.parametrize("name", ["survey", "toy_gauss2d_b64"])
def test_toy_matches_reference(name):
    g = load_golden(name)
    out = M.toy_get_matched_features(g["toy_a"], g["toy_b"], float(g["toy_lam"]), int(g["toy_iters"]))
    for k, got in zip(("aa", "bb", "ab", "ba"), out[:4]):
        np.testing.assert_allclose(got, g["toy_" + k], rtol=1e-10, atol=1e-12)
    assert out[4] == pytest.approx(float(g["toy_entropy"]), rel=1e-10)
    d = M.toy_calc_distance(g["toy_a"], g["toy_b"], out)
    assert d == pytest.approx(float(g["toy_distance"]), rel=1e-9)


def test_closed_form_equals_calc_distance(list_case):
    # cancellation-free closed form == reference formula (fp64)
    g = list_case
    fa, fb = _lists(g)
    d = M.closed_form_distance(fa, fb, float(g["lam"]), int(g["iters"]))
    assert d == pytest.approx(float(g["two_distance"]), rel=1e-8, abs=1e-13)


def test_potential_form_equals_inplace_form():
    rng = np.random.RandomState(11)
    x = np.abs(rng.randn(24, 40)); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.abs(rng.randn(24, 40) + 1.0); y /= np.linalg.norm(y, axis=1, keepdims=True)
    C = M.cosine_cost(x, y)
    Mi, ent, log_a = M.sinkhorn_plan(C, 500.0, 23)
    K = -500.0 * C
    f, g, f_fin = M.sinkhorn_potentials(K, 23)
    np.testing.assert_allclose(K + f[:, None] + g[None, :], log_a, rtol=0, atol=1e-9)
    Mp = np.exp(K + f_fin[:, None] + g[None, :])
    np.testing.assert_allclose(Mp, Mi, rtol=1e-9, atol=1e-15)
    # rows of the plan sum to one, total mass N; columns only approximately (ends on a row softmax)
    np.testing.assert_allclose(Mi.sum(1), 1.0, rtol=1e-12)


def test_exact_iteration_count_matters():
    # F10b: loss is not monotone / converged in L -- the kernel must run exactly L sweeps
    g = load_golden("clustered_s2_b16_d64")
    fa, fb = _lists(g)
    d37 = M.closed_form_distance(fa, fb, float(g["lam"]), 37)
    d38 = M.closed_form_distance(fa, fb, float(g["lam"]), 38)
    assert abs(d37 - d38) > 1e-9 * abs(d37)


def test_permutation_equivariance():
    g = load_golden("clustered_s2_b16_d64")
    fa, fb = _lists(g)
    out = M.get_matched_features(fa, fb, 500.0, 20)
    rng = np.random.RandomState(0)
    p = rng.permutation(fa[0].shape[0])
    fa2 = [fa[0][p], fa[1]]
    out2 = M.get_matched_features(fa2, fb, 500.0, 20)
    np.testing.assert_allclose(out2[0][0], out[0][0][p], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out2[2][0], out[2][0][p], rtol=1e-9, atol=1e-12)
    assert out2[4] == pytest.approx(out[4], rel=1e-10)


def test_row_helpers_agree_with_full_two_batch():
    """two_batch_plans / matched_rows / closed_form_from (used by the N = 1024 rank tests) are the same
    arithmetic as get_matched_features / calc_distance."""
    rng = np.random.RandomState(3)
    S, B, D = 4, 6, 40
    ca, cb = rng.randn(4, D), rng.randn(4, D)
    fa = [M.clustered_features(rng, B, D, ca) for _ in range(S)]
    fb = [M.clustered_features(rng, B, D, cb) for _ in range(S)]
    ref = M.get_matched_features(fa, fb, 200.0, 17)
    fa1, fa2 = np.concatenate(fa[:2]), np.concatenate(fa[2:])
    fb1, fb2 = np.concatenate(fb[:2]), np.concatenate(fb[2:])
    plans, costs, ent = M.two_batch_plans(fa1, fa2, fb1, fb2, 200.0, 17)
    assert ent == pytest.approx(ref[4], rel=1e-13)
    N = 2 * B
    for half in (0, 1):
        rows = M.matched_rows(plans, fa1, fa2, fb1, fb2, half, 3, 9)
        for got, full in zip(rows, ref[:4]):
            want = np.concatenate(full)[half * N + 3: half * N + 9]
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)
    d = M.closed_form_from(plans, costs, N)
    assert d == pytest.approx(M.calc_distance(fa, fb, ref), rel=1e-9)
