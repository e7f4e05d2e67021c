"""GPU: the one-tile matching GEMMs on two fp16 pieces (round 5: cost128_h2_kernel / plan_apply128_h2_kernel, the N <= 128
problems of a single-GPU step; reference utils/matching.py:29-39 and :64-83) at the shapes the other suites do not reach --
ragged rows (n != m, n < 128), feature widths that are not multiples of 16 or 128 (k tail, last d tile), one k step per
split, row ranges of a rank -- against the fp64 oracle; run-to-run bit identity; and that shapes outside the kernels'
contract (contraction length not a multiple of 16, D % 4 != 0) still go through the exact-fp32 kernels."""
import numpy as np
import pytest
import torch

from conftest import REL_DIFF_INJECTED
from oracle import matching_np as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from otgan_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _t(x, dev):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device=dev)


def _rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)


@pytest.mark.parametrize("n,m,D", [(128, 128, 32768),    # configs[1]: 64 K splits of 32 k steps
                                   (128, 96, 36),        # two k steps + a tail of 4; n != m
                                   (17, 128, 1028),      # ragged rows, k tail
                                   (64, 64, 16),         # ONE k step in all
                                   (128, 128, 7296)])    # configs[3] width
def test_cost_h2_vs_fp64(dev, n, m, D):
    from otgan_amd.utils import matching
    rng = np.random.RandomState(n + m + D)
    nrm = lambda z: z / np.linalg.norm(z, axis=1, keepdims=True)
    X, Y = nrm(np.abs(rng.randn(n, D)) + 0.1).astype(np.float32), nrm(rng.randn(m, D) ** 2).astype(np.float32)
    x, y = _t(X, dev), _t(Y, dev)
    lam = 500.0
    K = matching.cost_log_kernel(x, y, lam)
    ref = -lam * M.cosine_cost(X.astype(np.float64), Y.astype(np.float64))
    assert np.abs(K.cpu().numpy() - ref).max() < 2e-6 * lam          # measured <= 3e-7 * lambda
    assert torch.equal(matching.cost_log_kernel(x, y, lam), K)        # fixed summation order: the same bits every run
    K3 = matching.cost_log_kernels([x, x], [y, y], lam)
    assert torch.equal(K3[0], K3[1])


@pytest.mark.parametrize("N,D,iters", [(128, 1000, 30),   # D % 128 != 0: the last d tile is ragged; D % 16 != 0
                                       (48, 260, 20),     # three k steps per term, rows < 128
                                       (16, 64, 10)])     # one k step per term
def test_grads_h2_vs_oracle(dev, N, D, iters):
    from otgan_amd.utils import matching
    lam = 500.0
    rng = np.random.RandomState(N + D)
    ca, cb = rng.randn(8, D), rng.randn(8, D)
    fa = M.clustered_features(rng, 2 * N, D, ca).astype(np.float32)
    fb = M.clustered_features(rng, 2 * N, D, cb).astype(np.float32)
    A, B = _t(fa, dev), _t(fb, dev)
    ga, gb, ent, dist = matching.matched_feature_grads(A, B, lam, iters)
    f64 = lambda z: z.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa[:N]), f64(fa[N:]), f64(fb[:N]), f64(fb[N:])
    plans, costs, ent_ref = M.two_batch_plans(fa1, fa2, fb1, fb2, lam, iters)
    ra, rb = [], []
    for half in (0, 1):
        aa, bb, ab, ba = M.matched_rows(plans, fa1, fa2, fb1, fb2, half, 0, N)
        ra.append(aa - ab)
        rb.append(bb - ba)
    assert _rel(ga.cpu().numpy(), np.concatenate(ra)) < REL_DIFF_INJECTED
    assert _rel(gb.cpu().numpy(), np.concatenate(rb)) < REL_DIFF_INJECTED
    assert float(ent) == pytest.approx(float(ent_ref), rel=2e-4)
    ga2, gb2, _, _ = matching.matched_feature_grads(A, B, lam, iters)
    assert torch.equal(ga2, ga) and torch.equal(gb2, gb)
    # a rank's rows (second mini-batch, generator step): the same values as the full call's rows
    if N >= 32:
        r0, cnt = N + N // 2, N // 4
        gr, none, _, _ = matching.matched_feature_grads(A, B, lam, iters, need_b=False, rows=(r0, cnt))
        assert none is None and _rel(gr.cpu().numpy(), ga[r0:r0 + cnt].cpu().numpy()) < 1e-6
    # the reference's operator (eight blocks of one or two terms) on the same kernels
    out = matching.get_matched_features([A[:N], A[N:]], [B[:N], B[N:]], lam, iters)
    da = (torch.cat(out[0]) - torch.cat(out[2])).cpu().numpy()
    assert _rel(ga.cpu().numpy(), da) < 1e-4


@pytest.mark.parametrize("N,D", [(40, 100), (24, 66)])      # contraction length 40 / 24: not a multiple of 16; D % 4 != 0
def test_outside_the_contract_runs_the_fp32_kernels(dev, N, D):
    from otgan_amd.utils import matching
    lam, iters = 500.0, 12
    rng = np.random.RandomState(N * D)
    ca, cb = rng.randn(4, D), rng.randn(4, D)
    fa = M.clustered_features(rng, 2 * N, D, ca).astype(np.float32)
    fb = M.clustered_features(rng, 2 * N, D, cb).astype(np.float32)
    ga, gb, _, _ = matching.matched_feature_grads(_t(fa, dev), _t(fb, dev), lam, iters)
    f64 = lambda z: z.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa[:N]), f64(fa[N:]), f64(fb[:N]), f64(fb[N:])
    plans, _, _ = M.two_batch_plans(fa1, fa2, fb1, fb2, lam, iters)
    aa, bb, ab, ba = M.matched_rows(plans, fa1, fa2, fb1, fb2, 0, 0, N)
    assert _rel(ga[:N].cpu().numpy(), aa - ab) < REL_DIFF_INJECTED
    assert _rel(gb[:N].cpu().numpy(), bb - ba) < REL_DIFF_INJECTED


def test_cost_h2_row_pitch(dev):
    """features inside a wider buffer (ldf > D), through the C ABI: the same log-kernel as from a packed copy, bit for bit"""
    from otgan_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(9)
    n, m, D, ld = 64, 96, 100, 112
    nrm = lambda z: z / np.linalg.norm(z, axis=1, keepdims=True)
    X, Y = nrm(rng.randn(n, D)).astype(np.float32), nrm(rng.randn(m, D)).astype(np.float32)
    xw, yw = torch.full((n, ld), 7.0, device=dev), torch.full((m, ld), -3.0, device=dev)     # (the padding must not be read)
    xw[:, :D], yw[:, :D] = _t(X, dev), _t(Y, dev)
    out = []
    for x, y, l in ((xw, yw, ld), (_t(X, dev), _t(Y, dev), D)):
        K = torch.empty(n, m, device=dev)
        need = L.otgan_cost_matrix_workspace_bytes(n, m, D)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        _lib.check(L.otgan_cost_matrix_f32(x.data_ptr(), y.data_ptr(), n, m, D, l, 500.0, 0, 0.0, K.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _lib.stream_ptr()), "cost")
        out.append(K)
    assert torch.equal(out[0], out[1])
    ref = -500.0 * M.cosine_cost(X.astype(np.float64), Y.astype(np.float64))
    assert np.abs(out[0].cpu().numpy() - ref).max() < 1e-3


@pytest.mark.parametrize("N,D", [(128, 512), (256, 512)], ids=["one_tile_kernels", "pre_split_operands"])
def test_features_outside_the_cosine_contract_are_loud(dev, N, D):
    """include/otgan.h: the default matching engine splits its operands into two fp16 pieces with an a-priori scale for rows of
    unit length (what the reference's critics return, models/dcgan.py:16-19); the reference itself has no such precondition
    (utils/matching.py:29-39 multiplies whatever it is given).  The contract is therefore: |x| < 8 is computed correctly -- rows
    that are NOT of unit length included -- and an element of magnitude >= 8 poisons the result with NaN instead of returning a
    wrong finite number.  Both engines of the split path: N <= 128 (cost128_h2 / plan_apply128_h2) and N >= 256 (pre-split)."""
    from otgan_amd.utils import matching
    lam, iters = 20.0, 10
    rng = np.random.RandomState(N + D)
    ca, cb = rng.randn(4, D), rng.randn(4, D)
    fa = M.clustered_features(rng, 2 * N, D, ca).astype(np.float32)
    fb = M.clustered_features(rng, 2 * N, D, cb).astype(np.float32)
    # inside the contract but not of unit length: rows scaled by 0.5 .. 3 (largest element well below 8)
    sa = (0.5 + 2.5 * rng.rand(2 * N, 1)).astype(np.float32)
    fa_s = fa * sa
    assert np.abs(fa_s).max() < 8.0
    ga, gb, ent, dist = matching.matched_feature_grads(_t(fa_s, dev), _t(fb, dev), lam, iters)
    f64 = lambda z: z.astype(np.float64)
    a1, a2, b1, b2 = f64(fa_s[:N]), f64(fa_s[N:]), f64(fb[:N]), f64(fb[N:])
    plans, _, _ = M.two_batch_plans(a1, a2, b1, b2, lam, iters)
    aa, bb, ab, ba = M.matched_rows(plans, a1, a2, b1, b2, 0, 0, N)
    assert np.isfinite(float(dist)) and np.isfinite(float(ent))
    assert _rel(ga[:N].cpu().numpy(), aa - ab) < 2e-4
    # outside: ONE element of 9.0
    bad = fa.copy()
    bad[3, 5] = 9.0
    ga, gb, ent, dist = matching.matched_feature_grads(_t(bad, dev), _t(fb, dev), lam, iters)
    loud = not np.isfinite(float(dist)) or not bool(torch.isfinite(ga).all())
    if N <= 128:
        assert loud, "the one-tile kernels' a-priori scale cannot hold 9.0: the result must be NaN, never a wrong finite number"
    elif not loud:
        # the pre-split engine measures its operands (speculative split, re-split when the largest magnitude leaves the expected
        # band): it may take the element -- then the answer has to be RIGHT
        a1, a2 = f64(bad[:N]), f64(bad[N:])
        b1, b2 = f64(fb[:N]), f64(fb[N:])
        plans, _, _ = M.two_batch_plans(a1, a2, b1, b2, lam, iters)
        aa, bb, ab, ba = M.matched_rows(plans, a1, a2, b1, b2, 0, 0, N)
        assert _rel(ga[:N].cpu().numpy(), aa - ab) < 2e-4
