"""GPU: the rank-level workload of the 8-GPU configs (BASELINE configs[2] DCGAN / configs[3] DenseNet) on ONE GPU.

The reference forms the global mini-batch halves across towers (utils/matching.py:16-19), row-shards the
cost GEMMs over the devices (:29-39) and applies the plans (:64-83).  The build's rank does the same in
`trainer._sharded_log_kernels` + `matching.get_matched_features_rows` (otgan_matching_two_batch_rows_f32 with
K_pre).  Here "rank r of 8" is emulated on one device with the SAME functions the trainer calls
(`trainer.rank_log_kernel_slices`, `trainer.assemble_log_kernels`): S = 16 shards x 128 -> N = 1024 rows per
half, 256 rows per rank, D = 32768 (DCGAN, L = 100) and D = 7296 (DenseNet, L = 200), lambda = 500; rows,
entropy and distance of ranks 0, 3, 4, 7 are compared with the fp64 oracle (oracle/matching_np.py) at the
matching tolerances of tests/test_matching_gpu.py (features 2e-4 rel. L2, loss 1e-4 rel.)."""
import numpy as np
import pytest
import torch

from conftest import REL_DIFF_INJECTED
from oracle import matching_np as M

pytestmark = pytest.mark.gpu

REL_FEAT = 2e-4
REL_LOSS = 1e-4
WORLD = 8


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)


def _features(seed, S, B, D):
    rng = np.random.RandomState(seed)
    ca, cb = rng.randn(32, D), rng.randn(32, D)
    fa = np.concatenate([M.clustered_features(rng, B, D, ca) for _ in range(S)]).astype(np.float32)
    fb = np.concatenate([M.clustered_features(rng, B, D, cb) for _ in range(S)]).astype(np.float32)
    return fa, fb


@pytest.mark.parametrize("D,iters", [(32768, 100), (7296, 200)], ids=["cfg3_dcgan", "cfg4_densenet"])
def test_rank_of_8_rows_vs_oracle(dev, D, iters):
    from otgan_amd import trainer
    from otgan_amd.utils import matching
    S, B, lam = 16, 128, 500.0
    shards_per_rank = S // WORLD
    nb = shards_per_rank * B                      # 256 rows per rank
    N = S * B // 2                                # 1024
    fa_h, fb_h = _features(11, S, B, D)
    fa_d, fb_d = torch.as_tensor(fa_h, device=dev), torch.as_tensor(fb_h, device=dev)
    fa = list(torch.chunk(fa_d, S, 0))            # what gather_feature_shards returns on every rank
    fb = list(torch.chunk(fb_d, S, 0))
    # every rank's [3, nb, N] slices, then the all-gather + assembly the trainer performs
    slices = [trainer.rank_log_kernel_slices(r, WORLD, fa_d[r * nb:(r + 1) * nb], fb_d[r * nb:(r + 1) * nb],
                                             fa, fb, lam) for r in range(WORLD)]
    K = trainer.assemble_log_kernels(torch.stack(slices, 0), WORLD)
    assert tuple(K.shape) == (6, N, N)

    f64 = lambda z: z.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa_h[:N]), f64(fa_h[N:]), f64(fb_h[:N]), f64(fb_h[N:])
    plans, costs, ent_ref = M.two_batch_plans(fa1, fa2, fb1, fb2, lam, iters)
    dist_ref = M.closed_form_from(plans, costs, N)
    # the assembled log-kernels are the reference's six cost matrices (matching.py:41-43,50)
    for p, name in enumerate(["a1a2", "b2b1", "a1b1", "a1b2", "a2b1", "a2b2"]):
        assert np.abs(K[p].cpu().numpy() + lam * costs[name]).max() < 2e-5 * lam, name

    for r in (0, 3, 4, 7):
        outs, ent, dist = matching.get_matched_features_rows(fa, fb, lam, iters, r * nb, nb, K)
        half, r0 = divmod(r * nb, N)
        ref = M.matched_rows(plans, fa1, fa2, fb1, fb2, half, r0, r0 + nb)
        for k, got, want in zip("aa bb ab ba".split(), outs, ref):
            assert _rel(got.cpu().numpy(), want) < REL_FEAT, (r, k)
        # the injected gradients of the rank's samples (train.py:111,125-126)
        g_gen = (outs[0] - outs[2]).cpu().numpy()
        assert _rel(g_gen, ref[0] - ref[2]) < 2 * REL_DIFF_INJECTED, r     # (difference of two separately rounded fp32 arrays)
        assert float(ent) == pytest.approx(float(ent_ref), rel=2e-4)
        assert abs(float(dist) - dist_ref) <= REL_LOSS * abs(dist_ref) + 1e-7, (r, float(dist), dist_ref)


def test_rank_of_8_rows_without_precomputed_kernels(dev):
    """K_pre = NULL: the library computes the six log-kernels itself; rows of rank 5 at N = 1024."""
    from otgan_amd.utils import matching
    S, B, D, lam, iters = 16, 128, 2048, 500.0, 40
    N, nb = S * B // 2, 256
    fa_h, fb_h = _features(12, S, B, D)
    fa = list(torch.chunk(torch.as_tensor(fa_h, device=dev), S, 0))
    fb = list(torch.chunk(torch.as_tensor(fb_h, device=dev), S, 0))
    outs, ent, dist = matching.get_matched_features_rows(fa, fb, lam, iters, 5 * nb, nb, None)
    f64 = lambda z: z.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa_h[:N]), f64(fa_h[N:]), f64(fb_h[:N]), f64(fb_h[N:])
    plans, costs, ent_ref = M.two_batch_plans(fa1, fa2, fb1, fb2, lam, iters)
    ref = M.matched_rows(plans, fa1, fa2, fb1, fb2, 1, nb, 2 * nb)
    for k, got, want in zip("aa bb ab ba".split(), outs, ref):
        assert _rel(got.cpu().numpy(), want) < REL_FEAT, k
    assert float(ent) == pytest.approx(float(ent_ref), rel=2e-4)
    dref = M.closed_form_from(plans, costs, N)
    assert abs(float(dist) - dref) <= REL_LOSS * abs(dref) + 1e-7


def test_single_batch_global_scope_rank_shards(dev):
    """--single_batch in the global scope (matching.py:88-136 over ALL shards; trainer._match picks the rank's
    shards): S = 8 shards x 128 -> three problems of 1024 rows (+999 on the a-a / b-b diagonals), D = 7296."""
    from otgan_amd.utils import matching
    S, B, D, lam, iters = 8, 128, 7296, 500.0, 50
    fa_h, fb_h = _features(13, S, B, D)
    fa = list(torch.chunk(torch.as_tensor(fa_h, device=dev), S, 0))
    fb = list(torch.chunk(torch.as_tensor(fb_h, device=dev), S, 0))
    out = matching.get_matched_features_single_batch(fa, fb, lam, iters)
    ref = M.get_matched_features_single_batch(list(np.split(fa_h, S)), list(np.split(fb_h, S)), lam, iters)
    dref = float(M.calc_distance(list(np.split(fa_h, S)), list(np.split(fb_h, S)), ref))
    world, shards = 4, 2
    for r in (0, 3):                                 # the shards rank r of 4 picks (trainer._match: lo = rank*shards)
        lo = r * shards
        for k, got, want in zip("aa bb ab ba".split(), out[:4], ref[:4]):
            g = torch.cat(got[lo:lo + shards], 0).cpu().numpy()
            assert _rel(g, np.concatenate(want[lo:lo + shards], 0)) < REL_FEAT, (r, k)
    assert float(out[4]) == pytest.approx(float(ref[4]), rel=2e-4)
    d = float(matching.calc_distance(fa, fb, out))
    assert abs(d - dref) <= REL_LOSS * abs(dref) + 1e-7, (d, dref)
    d3 = float(matching.closed_form_distance(out))
    assert abs(d3 - dref) <= REL_LOSS * abs(dref) + 1e-7


def test_panel_kernel_falls_back_when_grid_cannot_be_resident(dev, monkeypatch):
    """The persistent Sinkhorn kernel needs all its workgroups co-resident; on a device that cannot host
    them (emulated: OTGAN_PANEL_MAX_WG) the launcher must take the multi-launch path and give the same plan."""
    from otgan_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(1)
    P, n, iters, lam = 2, 256, 15, 500.0
    Kh = (-lam * rng.rand(P, n, n) * 0.3).astype(np.float32)
    K = torch.as_tensor(Kh, device=dev)

    def run():
        plan = torch.empty(P, n, n, device=dev)
        planT = torch.empty(P, n, n, device=dev)
        stats = torch.empty(P, 4, dtype=torch.float64, device=dev)
        need = max(L.otgan_sinkhorn_workspace_bytes(P, n, n), 256)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _lib.check(L.otgan_sinkhorn_plan_f32(K.data_ptr(), P, n, n, iters, lam, plan.data_ptr(), planT.data_ptr(),
                                             stats.data_ptr(), ws.data_ptr(), need, _lib.stream_ptr()), "sinkhorn")
        return plan.cpu().numpy(), stats.cpu().numpy()

    a, sa = run()
    monkeypatch.setenv("OTGAN_PANEL_MAX_WG", "1")
    b, sb = run()
    assert _rel(b, a) < 1e-5
    np.testing.assert_allclose(sb[:, :3], sa[:, :3], rtol=1e-5)
    assert np.all(sa[:, 3] == 0) and np.all(sb[:, 3] == 0)
    for p in range(P):
        Mref, _, _ = M.sinkhorn_plan(-Kh[p].astype(np.float64) / lam, lam, iters)
        assert _rel(b[p], Mref) < 1e-4


@pytest.mark.parametrize("D,iters", [(32768, 100), (7296, 200)], ids=["cfg3_dcgan", "cfg4_densenet"])
def test_rank_of_8_feature_stack(dev, D, iters):
    """Round 5: a rank splits the gathered features into the GEMM engine's operand ONCE per step (matching.FeatureStack,
    otgan_matching_stack_split_f32) and both of its library calls -- cost row slices (matching.py:29-39) and plans applied to
    its rows (:64-83) -- read that stack.  Every rank's slices from its stack assemble to the log-kernels of the
    one-split-per-call path, and the injected gradients of ranks 0, 3 (first half), 4, 7 (second half), generator and
    critic steps, match the fp64 oracle and the older path."""
    from otgan_amd import trainer
    from otgan_amd.utils import matching
    S, B, lam = 16, 128, 500.0
    nb, N = 2 * B, S * B // 2
    fa_h, fb_h = _features(11, S, B, D)
    fa_d, fb_d = torch.as_tensor(fa_h, device=dev), torch.as_tensor(fb_h, device=dev)
    assert trainer.rank_stack_ok(nb, fa_d)
    fa, fb = list(torch.chunk(fa_d, S, 0)), list(torch.chunk(fb_d, S, 0))
    old = torch.stack([trainer.rank_log_kernel_slices(r, WORLD, fa_d[r * nb:(r + 1) * nb], fb_d[r * nb:(r + 1) * nb], fa, fb, lam)
                       for r in range(WORLD)], 0)
    mines = []
    for r in range(WORLD):
        for need_b in (False, True):        # the stack's ranges (and where the rank's own rows sit) depend on the step kind
            ranges, own_gen, own_dat = matching.FeatureStack.rank_plan(r * nb, nb, N, need_b)
            st = matching.FeatureStack(fa_d, fb_d, ranges)
            if r < WORLD // 2:
                m = st.cost_slices([own_gen] * 3, [3 * N, N, 2 * N], nb, lam)
            else:
                m = st.cost_slices([own_dat, own_gen, own_gen], [N, N, 2 * N], nb, lam)
            assert (m - old[r]).abs().max() <= 2e-5 * lam, (r, need_b)
        mines.append(m)
    allk = torch.stack(mines, 0)
    K_old = trainer.assemble_log_kernels(old, WORLD)
    f64 = lambda z: z.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa_h[:N]), f64(fa_h[N:]), f64(fb_h[:N]), f64(fb_h[N:])
    plans, costs, ent_ref = M.two_batch_plans(fa1, fa2, fb1, fb2, lam, iters)
    dref = M.closed_form_from(plans, costs, N)
    for r in (0, 3, 4, 7):
        half, r0 = divmod(r * nb, N)
        aa, bb, ab, ba = M.matched_rows(plans, fa1, fa2, fb1, fb2, half, r0, r0 + nb)
        for need_b in (False, True):
            ga, gb, ent, dist = trainer.rank_matching_stack(r, WORLD, nb, fa_d, fb_d, lam, iters, need_b, gather=allk)
            assert _rel(ga.cpu().numpy(), aa - ab) < REL_DIFF_INJECTED, (r, need_b)
            assert (gb is None) == (not need_b)
            if need_b:
                assert _rel(gb.cpu().numpy(), bb - ba) < REL_DIFF_INJECTED, r
            assert float(ent) == pytest.approx(float(ent_ref), rel=2e-4)
            assert abs(float(dist) - dref) <= REL_LOSS * abs(dref) + 1e-7, (r, float(dist), dref)
            ga0, gb0, _, _ = matching.matched_feature_grads(fa_d, fb_d, lam, iters, need_b=need_b, rows=(r * nb, nb), log_kernels=K_old)
            assert _rel(ga.cpu().numpy(), ga0.cpu().numpy()) < 1e-5, (r, need_b)
