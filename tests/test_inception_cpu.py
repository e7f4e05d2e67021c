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
