"""CPU oracle (NumPy) for the OT-GAN mini-batch Sinkhorn energy distance.

TEST INFRASTRUCTURE ONLY -- imported by `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py`; never by the product path under `ot-gan_amd/`.

This is an independent restatement (not a copy) of the algorithm in the reference:

  * two-batch matching        /root/reference/utils/matching.py:11-85
  * single-batch matching     /root/reference/utils/matching.py:88-136
  * random matching           /root/reference/utils/matching.py:3-9
  * distance                  /root/reference/utils/matching.py:139-153
  * toy (sq-Euclid) variants  /root/reference/toy_example/matching_cpu.py:4-95,98-152,155-164

Pinning: `oracle/make_golden.py` runs the reference's own, unmodified files over
`oracle/tf_standin.py` in the build container and stores inputs/outputs under
`tests/golden/`; `tests/test_oracle.py` checks this module against those vectors
(and against the sanity values recorded in SURVEY.md section 8c).

Everything is evaluated in `dtype` (float64 by default: the authoritative truth, see
SURVEY.md F10a -- two legal fp32 evaluations of the reference's loss already differ by
2e-4 relative).
"""
import numpy as np


# ----------------------------------------------------------------------------- helpers
def _lse(x, axis):
    """Max-shifted log-sum-exp, keepdims (semantics of tf.reduce_logsumexp)."""
    m = np.max(x, axis=axis, keepdims=True)
    return np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)) + m


def sinkhorn_plan(cost, sinkhorn_lambda, nr_sinkhorn_iter):
    """Log-domain Sinkhorn-Knopp exactly as utils/matching.py:50-57.

    log_a = -lambda*C; repeat L times {rows -= LSE_rows; cols -= LSE_cols}; the plan is
    the *row* softmax of the final log_a (rows sum to 1, total mass N) and the entropy
    is the mean row entropy of the plan.
    Returns (M, entropy, log_a).
    """
    log_a = -sinkhorn_lambda * cost
    for _ in range(int(nr_sinkhorn_iter)):
        log_a = log_a - _lse(log_a, 1)
        log_a = log_a - _lse(log_a, 0)
    log_m = log_a - _lse(log_a, 1)
    M = np.exp(log_m)
    entropy = np.mean(-np.sum(M * log_m, axis=1))
    return M, entropy, log_a


def sinkhorn_potentials(K, nr_sinkhorn_iter):
    """Potential form of the same iteration.

    log_a == K + f[:,None] + g[None,:] at every point of the reference loop, with
    f_i <- f_i - LSE_j(K_ij+f_i+g_j) == -LSE_j(K_ij + g_j), then g_j likewise.
    This is the formulation the HIP kernels use; kept here so tests can verify the two
    forms agree in fp64.  Returns (f, g) after L sweeps and the final row-softmax f.
    """
    n, m = K.shape
    f = np.zeros(n, K.dtype)
    g = np.zeros(m, K.dtype)
    for _ in range(int(nr_sinkhorn_iter)):
        f = -_lse(K + g[None, :], 1)[:, 0]
        g = -_lse(K + f[:, None], 0)[0, :]
    f_final = -_lse(K + g[None, :], 1)[:, 0]
    return f, g, f_final


def cosine_cost(x, y):
    """utils/matching.py:31 -- C = 1 - X.Y^T (rows are unit L2 norm by construction)."""
    return 1.0 - x @ y.T


def sqeuclid_mean_cost(x, y):
    """toy_example/matching_cpu.py:17-21 -- C = 0.5*mean(x^2) + 0.5*mean(y^2) - x.y^T/n."""
    n = x.shape[1]
    return (0.5 * np.mean(np.square(x), axis=1, keepdims=True)
            + 0.5 * np.mean(np.square(y), axis=1)[None, :] - (x @ y.T) / n)


# ----------------------------------------------------------------------------- two-batch
def _two_batch_core(fa1, fa2, fb1, fb2, lam, iters, cost_fn):
    """Six problems, twelve products (utils/matching.py:29-83). Plain [N,D] arrays."""
    pairs = [("a1a2", fa1, fa2), ("b2b1", fb2, fb1), ("a1b1", fa1, fb1),
             ("a1b2", fa1, fb2), ("a2b1", fa2, fb1), ("a2b2", fa2, fb2)]
    plans, ents, costs = {}, [], {}
    for name, x, y in pairs:
        C = cost_fn(x, y)
        M, e, _ = sinkhorn_plan(C, lam, iters)
        plans[name], costs[name] = M, C
        ents.append(e)
    entropy = sum(ents) / len(ents)
    a1_a2 = plans["a1a2"] @ fa2
    a2_a1 = plans["a1a2"].T @ fa1
    b1_b2 = plans["b2b1"].T @ fb2
    b2_b1 = plans["b2b1"] @ fb1
    a1_b1 = plans["a1b1"] @ fb1
    a1_b2 = plans["a1b2"] @ fb2
    a2_b1 = plans["a2b1"] @ fb1
    a2_b2 = plans["a2b2"] @ fb2
    b1_a1 = plans["a1b1"].T @ fa1
    b2_a1 = plans["a1b2"].T @ fa1
    b1_a2 = plans["a2b1"].T @ fa2
    b2_a2 = plans["a2b2"].T @ fa2
    f_aa = np.concatenate([a1_a2, a2_a1], 0)
    f_bb = np.concatenate([b1_b2, b2_b1], 0)
    f_ab = 0.5 * (np.concatenate([a1_b1, a2_b1], 0) + np.concatenate([a1_b2, a2_b2], 0))
    f_ba = 0.5 * (np.concatenate([b1_a1, b2_a1], 0) + np.concatenate([b1_a2, b2_a2], 0))
    return f_aa, f_bb, f_ab, f_ba, entropy, plans, costs


def two_batch_plans(fa1, fa2, fb1, fb2, lam, iters, threads=6):
    """The six plans / costs / mean entropy of the two-batch matching (matching.py:29-61) without the
    twelve products -- for tests at the 8-GPU problem size (N = 1024), where only some rows of the
    matched features are compared.  The six independent problems run on a small thread pool (NumPy
    releases the GIL inside its ufuncs); the arithmetic is sinkhorn_plan()'s, unchanged."""
    from concurrent.futures import ThreadPoolExecutor
    pairs = [("a1a2", fa1, fa2), ("b2b1", fb2, fb1), ("a1b1", fa1, fb1),
             ("a1b2", fa1, fb2), ("a2b1", fa2, fb1), ("a2b2", fa2, fb2)]

    def solve(item):
        name, x, y = item
        C = cosine_cost(x, y)
        M, e, _ = sinkhorn_plan(C, lam, iters)
        return name, M, C, e

    with ThreadPoolExecutor(max_workers=threads) as ex:
        res = list(ex.map(solve, pairs))
    plans = {n: M for n, M, _, _ in res}
    costs = {n: C for n, _, C, _ in res}
    entropy = sum(e for *_, e in res) / len(res)
    return plans, costs, entropy


def matched_rows(plans, fa1, fa2, fb1, fb2, half, r0, r1):
    """Rows [r0, r1) of mini-batch `half` (0: a1/b1, 1: a2/b2) of f_aa, f_bb, f_ab, f_ba
    (matching.py:64-83) from the six plans."""
    sl = slice(r0, r1)
    if half == 0:
        aa = plans["a1a2"][sl] @ fa2
        bb = plans["b2b1"].T[sl] @ fb2
        ab = 0.5 * (plans["a1b1"][sl] @ fb1 + plans["a1b2"][sl] @ fb2)
        ba = 0.5 * (plans["a1b1"].T[sl] @ fa1 + plans["a2b1"].T[sl] @ fa2)
    else:
        aa = plans["a1a2"].T[sl] @ fa1
        bb = plans["b2b1"][sl] @ fb1
        ab = 0.5 * (plans["a2b1"][sl] @ fb1 + plans["a2b2"][sl] @ fb2)
        ba = 0.5 * (plans["a1b2"].T[sl] @ fa1 + plans["a2b2"].T[sl] @ fa2)
    return aa, bb, ab, ba


def closed_form_from(plans, costs, N):
    """SURVEY 3.4 closed form of calc_distance from plans and costs."""
    W = {k: np.sum(plans[k] * costs[k]) for k in plans}
    return (W["a1b1"] + W["a1b2"] + W["a2b1"] + W["a2b2"] - 2 * W["a1a2"] - 2 * W["b2b1"]) / (4 * N)


def get_matched_features(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter,
                         dtype=np.float64):
    """Reference contract (utils/matching.py:11-85): lists of S shards of [B,D] in,
    four lists of S shards + scalar entropy out.  `a` = generated, `b` = data."""
    S = len(features_a)
    assert S % 2 == 0 and len(features_b) == S
    h = S // 2
    fa = [np.asarray(x, dtype) for x in features_a]
    fb = [np.asarray(x, dtype) for x in features_b]
    fa1, fa2 = np.concatenate(fa[:h], 0), np.concatenate(fa[h:], 0)
    fb1, fb2 = np.concatenate(fb[:h], 0), np.concatenate(fb[h:], 0)
    f_aa, f_bb, f_ab, f_ba, ent, _, _ = _two_batch_core(
        fa1, fa2, fb1, fb2, dtype(sinkhorn_lambda), nr_sinkhorn_iter, cosine_cost)
    sp = lambda z: list(np.split(z, S, 0))
    return sp(f_aa), sp(f_bb), sp(f_ab), sp(f_ba), ent


def get_matched_features_single_batch(features_a, features_b, sinkhorn_lambda,
                                      nr_sinkhorn_iter, dtype=np.float64):
    """utils/matching.py:88-136 -- three problems of size S*B, +999 on the a-a / b-b
    diagonal to forbid self matches."""
    S = len(features_a)
    fa = np.concatenate([np.asarray(x, dtype) for x in features_a], 0)
    fb = np.concatenate([np.asarray(x, dtype) for x in features_b], 0)
    n = fa.shape[0]
    lam = dtype(sinkhorn_lambda)
    eye = 999.0 * np.eye(n, dtype=dtype)
    Maa, eaa, _ = sinkhorn_plan(cosine_cost(fa, fa) + eye, lam, nr_sinkhorn_iter)
    Mbb, ebb, _ = sinkhorn_plan(cosine_cost(fb, fb) + eye, lam, nr_sinkhorn_iter)
    Mab, eab, _ = sinkhorn_plan(cosine_cost(fa, fb), lam, nr_sinkhorn_iter)
    ent = (eaa + ebb + eab) / 3
    sp = lambda z: list(np.split(z, S, 0))
    return sp(Maa @ fa), sp(Mbb @ fb), sp(Mab @ fb), sp(Mab.T @ fa), ent


def get_matched_features_random(features_a, features_b):
    """utils/matching.py:3-9 -- rotate the shard lists by one; entropy 0."""
    fa, fb = list(features_a), list(features_b)
    return fa[1:] + fa[:1], fb[1:] + fb[:1], fb, fa, 0.0


def calc_distance(features_a, features_b, matched_features, dtype=np.float64):
    """utils/matching.py:139-153: sum_i (nd_bb + nd_aa - 2 nd_ab) / (2*B*S)."""
    S = len(features_a)
    B = np.asarray(features_a[0]).shape[0]
    f_aa, f_bb, f_ab, _f_ba, _ = matched_features
    tot = dtype(0)
    for i in range(S):
        a = np.asarray(features_a[i], dtype)
        b = np.asarray(features_b[i], dtype)
        nd_aa = np.sum(a * np.asarray(f_aa[i], dtype))
        nd_bb = np.sum(b * np.asarray(f_bb[i], dtype))
        nd_ab = np.sum(a * np.asarray(f_ab[i], dtype))
        tot = tot + (nd_bb + nd_aa - 2.0 * nd_ab)
    return tot / (2 * B * S)


def closed_form_distance(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter,
                         dtype=np.float64):
    """Cancellation-free form of the two-batch distance:
    [W(a1,b1)+W(a1,b2)+W(a2,b1)+W(a2,b2) - 2W(a1,a2) - 2W(b2,b1)] / (4N),
    W(P,Q) = <M_PQ, C_PQ>.  Equal to calc_distance(get_matched_features(...)) for
    unit-norm rows; this is what the HIP path accumulates (in fp64)."""
    S = len(features_a)
    h = S // 2
    fa = [np.asarray(x, dtype) for x in features_a]
    fb = [np.asarray(x, dtype) for x in features_b]
    fa1, fa2 = np.concatenate(fa[:h], 0), np.concatenate(fa[h:], 0)
    fb1, fb2 = np.concatenate(fb[:h], 0), np.concatenate(fb[h:], 0)
    *_, plans, costs = _two_batch_core(fa1, fa2, fb1, fb2, dtype(sinkhorn_lambda),
                                       nr_sinkhorn_iter, cosine_cost)
    W = {k: np.sum(plans[k] * costs[k]) for k in plans}
    N = fa1.shape[0]
    return (W["a1b1"] + W["a1b2"] + W["a2b1"] + W["a2b2"]
            - 2 * W["a1a2"] - 2 * W["b2b1"]) / (4 * N)


# ----------------------------------------------------------------------------- toy (cfg 1)
def toy_get_matched_features(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter,
                             dtype=np.float64):
    """toy_example/matching_cpu.py:4-95 -- plain [2N,n] tensors, sq-Euclid/(2n) cost."""
    fa = np.asarray(features_a, dtype)
    fb = np.asarray(features_b, dtype)
    fa1, fa2 = np.split(fa, 2, 0)
    fb1, fb2 = np.split(fb, 2, 0)
    f_aa, f_bb, f_ab, f_ba, ent, _, _ = _two_batch_core(
        fa1, fa2, fb1, fb2, dtype(sinkhorn_lambda), nr_sinkhorn_iter, sqeuclid_mean_cost)
    return f_aa, f_bb, f_ab, f_ba, ent


def toy_get_matched_features_single_batch(features_a, features_b, sinkhorn_lambda,
                                          nr_sinkhorn_iter, batch_size, dtype=np.float64):
    """toy_example/matching_cpu.py:98-152 (lists of shards; eye size passed explicitly)."""
    S = len(features_a)
    fa = np.concatenate([np.asarray(x, dtype) for x in features_a], 0)
    fb = np.concatenate([np.asarray(x, dtype) for x in features_b], 0)
    lam = dtype(sinkhorn_lambda)
    eye = 999.0 * np.eye(int(batch_size), dtype=dtype)
    Maa, eaa, _ = sinkhorn_plan(sqeuclid_mean_cost(fa, fa) + eye, lam, nr_sinkhorn_iter)
    Mbb, ebb, _ = sinkhorn_plan(sqeuclid_mean_cost(fb, fb) + eye, lam, nr_sinkhorn_iter)
    Mab, eab, _ = sinkhorn_plan(sqeuclid_mean_cost(fa, fb), lam, nr_sinkhorn_iter)
    ent = (eaa + ebb + eab) / 3
    sp = lambda z: list(np.split(z, S, 0))
    return sp(Maa @ fa), sp(Mbb @ fb), sp(Mab @ fb), sp(Mab.T @ fa), ent


def toy_calc_distance(features_a, features_b, matched_features, dtype=np.float64):
    """toy_example/matching_cpu.py:155-164: (mean(b*bb)+mean(a*aa)-2 mean(a*ab))/2."""
    f_aa, f_bb, f_ab, _f_ba, _ = matched_features
    a = np.asarray(features_a, dtype)
    b = np.asarray(features_b, dtype)
    nd_aa = np.mean(a * np.asarray(f_aa, dtype))
    nd_bb = np.mean(b * np.asarray(f_bb, dtype))
    nd_ab = np.mean(a * np.asarray(f_ab, dtype))
    return (nd_bb + nd_aa - 2.0 * nd_ab) / 2.0


# ----------------------------------------------------------------------------- inputs
def clustered_features(rng, n, d, centres, sigma=0.1):
    """Non-negative unit-norm rows drawn around given centres (SURVEY.md section 8d:
    i.i.d. features give a distance ~ 0; benchmark/fixture inputs use two different
    centre sets for `a` and `b`)."""
    k = rng.randint(0, centres.shape[0], size=n)
    x = np.abs(centres[k] + sigma * rng.randn(n, d))
    return x / np.sqrt(np.sum(x * x, axis=1, keepdims=True))
