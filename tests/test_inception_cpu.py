"""The evaluation hook's score (reference utils/inception.py:43-51) on known answers, and the hook's contract."""
import numpy as np
import pytest

from otgan_amd.utils.inception import get_inception_score, inception_score_from_probs


def test_known_answers():
    n, k = 1000, 10
    uniform = np.full((n, k), 1.0 / k)
    m, s = inception_score_from_probs(uniform)
    assert m == pytest.approx(1.0, abs=1e-12) and s == pytest.approx(0.0, abs=1e-12)   # p(y|x) = p(y): KL = 0
    onehot = np.eye(k)[np.arange(n) % k]                 # confident and balanced inside every split: score = classes
    m, s = inception_score_from_probs(onehot)
    assert m == pytest.approx(float(k), rel=1e-12) and s == pytest.approx(0.0, abs=1e-9)
    collapsed = np.eye(k)[np.zeros(n, int)]              # confident but one class only: score = 1
    assert inception_score_from_probs(collapsed)[0] == pytest.approx(1.0, abs=1e-12)


def test_matches_a_direct_restatement():
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((700, 37)) * 2
    p = np.exp(logits - logits.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    splits = 7
    ref = []
    for i in range(splits):
        part = p[i * 100:(i + 1) * 100]
        ref.append(np.exp(np.mean(np.sum(part * (np.log(part) - np.log(part.mean(0, keepdims=True))), 1))))
    m, s = inception_score_from_probs(p, splits)
    assert m == pytest.approx(np.mean(ref), rel=1e-12) and s == pytest.approx(np.std(ref), rel=1e-10)


def test_hook_contract():
    imgs = [np.full((32, 32, 3), 127.5, np.float32) + i for i in range(40)]
    with pytest.raises(RuntimeError, match="no Inception classifier"):
        get_inception_score(imgs)
    seen = []

    def clf(batch):                                   # brightness -> two classes
        seen.append(batch.shape)
        b = batch.mean((1, 2, 3)) > 147.0
        return np.stack([b, ~b], 1).astype(np.float64)
    m, s = get_inception_score(imgs, splits=2, classifier=clf, batch_size=16)
    assert seen == [(16, 32, 32, 3), (16, 32, 32, 3), (8, 32, 32, 3)]
    assert 1.0 <= m <= 2.0
    with pytest.raises(AssertionError):               # the reference's input checks (values in 0..255, not -1..1)
        get_inception_score([np.zeros((32, 32, 3), np.float32)], classifier=clf)


# ---- the training driver's hook (train.py inception_hook): every rank classifies its share, scores agree ----------
class _FakeModel:
    """`sample(n, ema)` of a trainer: images in (-1, 1) whose brightness depends on a per-rank stream."""
    def __init__(self, seed):
        import torch
        self.device = torch.device("cpu")
        self.g = torch.Generator().manual_seed(seed)
        self.calls = []

    def sample(self, n, ema=False):
        import torch
        self.calls.append((n, ema))
        level = torch.rand((n, 1, 1, 1), generator=self.g) * 2 - 1
        return (level * (0.5 if ema else 1.0)).expand(n, 8, 8, 3).contiguous()


def _brightness_classifier(batch):
    b = np.clip(batch.mean((1, 2, 3)) / 255.0, 1e-3, 1 - 1e-3)
    return np.stack([b, 1 - b], 1)


def test_training_hook_single_rank():
    from types import SimpleNamespace
    from otgan_amd.train import inception_hook
    m = _FakeModel(3)
    state = {"max": 0.0, "iter": 0, "epoch": 7}
    out = inception_hook(m, SimpleNamespace(eval_samples=250), _brightness_classifier, state)
    assert sum(n for n, e in m.calls if not e) == 250 and sum(n for n, e in m.calls if e) == 250
    assert set(out) == {"live", "EMA"} and out["live"][0] > out["EMA"][0] >= 1.0
    assert state["max"] == out["live"][0] and state["iter"] == 7


def _hook_worker(rank, world, port, q):
    import os
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from otgan_amd import parallel
    from otgan_amd.train import inception_hook
    parallel.init_from_env(backend="gloo")
    m = _FakeModel(10 + rank)
    state = {"max": 0.0, "iter": 0, "epoch": 1}
    out = inception_hook(m, SimpleNamespace(eval_samples=251), _brightness_classifier, state, rank, world)
    q.put((rank, out["live"], out["EMA"], sum(n for n, e in m.calls if not e), state["max"]))
    parallel.barrier()


def test_training_hook_two_ranks_share_the_work():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_hook_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, live0, ema0, n0, mx0), (_, live1, ema1, n1, mx1) = got
    assert n0 == n1 == 126                       # ceil(251 / 2) samples drawn and classified per rank
    assert live0 == live1 and ema0 == ema1 and mx0 == mx1     # every rank holds the same gathered probabilities
