"""Evaluation hook of the reference's training driver: the Inception score of generated samples
(reference utils/inception.py:24-52, called from train.py:245-272 every 100 epochs on 50 000 samples of the
generator and of its EMA copy).

The reference downloads the 2015 Inception graph at import time (utils/inception.py:18,55-70); there is no network
here, so the CLASSIFIER is an input: any callable `images [n, H, W, 3] float32 in 0..255 -> class probabilities
[n, classes]`, or the path of a TorchScript module with that signature (`--inception_model`).  The score itself --
exp of the mean KL divergence between p(y|x) and the split's marginal p(y), mean and standard deviation over
`splits` equal parts -- is computed here exactly as the reference does it.  Without a classifier the hook is a
loud no-op (train.py prints that it is skipped); nothing in the training step depends on it.
"""
import math

import numpy as np


def inception_score_from_probs(preds, splits=10):
    """(mean, std) over `splits` consecutive parts of exp(mean_x KL(p(y|x) || p(y))) -- utils/inception.py:43-51.
    preds: [n, classes] rows of class probabilities."""
    preds = np.asarray(preds, dtype=np.float64)
    if preds.ndim != 2 or preds.shape[0] < splits:
        raise ValueError("need a [n, classes] array with at least `splits` rows")
    scores = []
    n = preds.shape[0]
    for i in range(splits):
        part = preds[(i * n // splits):((i + 1) * n // splits), :]
        marginal = np.mean(part, 0, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            kl = part * (np.log(part) - np.log(marginal))
        kl = np.where(part > 0, kl, 0.0)          # 0 log 0 = 0 (the reference's softmax outputs are never exactly 0)
        scores.append(math.exp(float(np.mean(np.sum(kl, 1)))))
    return float(np.mean(scores)), float(np.std(scores))


def load_classifier(path, device="cpu"):
    """A TorchScript module mapping float32 images [n, H, W, 3] in 0..255 to class probabilities, as a callable."""
    import torch
    mod = torch.jit.load(path, map_location=device).eval()

    def run(images):
        with torch.no_grad():
            return mod(torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32)).to(device)).float().cpu().numpy()
    return run


def get_inception_score(images, splits=10, classifier=None, batch_size=100):
    """Same contract as the reference (utils/inception.py:24-31): `images` is a list of [H, W, 3] arrays with values in
    0..255 (asserted the same way).  `classifier`: see the module docstring."""
    assert type(images) == list
    assert type(images[0]) == np.ndarray
    assert len(images[0].shape) == 3
    assert np.max(images[0]) > 10
    assert np.min(images[0]) >= 0.0
    if classifier is None:
        raise RuntimeError("no Inception classifier: the reference downloads one (utils/inception.py:18); pass "
                           "`classifier=` or train.py --inception_model <TorchScript file>")
    return inception_score_from_probs(class_probabilities(images, classifier, batch_size), splits)


def class_probabilities(images, classifier, batch_size=100):
    """[n, classes] class probabilities of a list of [H, W, 3] images in 0..255, `batch_size` at a time
    (utils/inception.py:35-42, bs = 100)."""
    preds = []
    for i in range(0, len(images), batch_size):
        batch = np.stack([im.astype(np.float32) for im in images[i:i + batch_size]], 0)
        preds.append(np.asarray(classifier(batch)))
    return np.concatenate(preds, 0)
