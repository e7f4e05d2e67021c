"""NumPy stand-in for the 16 TensorFlow-1.x symbols the reference's matching code touches.

TEST INFRASTRUCTURE ONLY.  This module exists so that the *unmodified* reference files
`/root/reference/utils/matching.py` and `/root/reference/toy_example/matching_cpu.py`
can be imported by path in the build container (TensorFlow 1.x is not installable here,
SURVEY.md F5 / section 8c) in order to

  1. pin `oracle/matching_np.py` (our independent restatement) against the reference's
     own control flow, and
  2. generate the golden vectors committed under `tests/golden/` (see
     `oracle/make_golden.py`).

Nothing in the product path (`ot-gan_amd/`) imports this file, and nothing on the GPU
box reads `/root/reference` (it does not exist there).

Semantics reproduced (TF 1.x):
  tf.matmul(a, b, transpose_a=, transpose_b=)
  tf.reduce_logsumexp(x, axis, keep_dims)   -- max-shifted: log(sum(exp(x-max))) + max
  tf.reduce_mean / tf.reduce_sum(x, axis, keep_dims)
  tf.nn.softmax(x)                          -- last axis
  tf.nn.softmax_cross_entropy_with_logits(labels=, logits=) = -sum(labels*log_softmax(logits), -1)
  tf.concat, tf.split(x, n, axis), tf.reshape, tf.square, tf.eye, tf.zeros, tf.float32
  tf.device(...)                            -- no-op context manager
  Tensor.get_shape().as_list()
"""
import contextlib
import sys
import types

import numpy as np

_DTYPE = [np.float64]


def set_dtype(dt):
    """Arithmetic type used by tensors created inside the stand-in (eye/zeros)."""
    _DTYPE[0] = np.dtype(dt).type


class _Shape(tuple):
    def as_list(self):
        return [int(s) for s in self]


class Tensor(np.ndarray):
    def get_shape(self):
        return _Shape(self.shape)


def T(x, dtype=None):
    a = np.asarray(x, dtype=dtype)
    return a.view(Tensor)


def _wrap(x):
    return np.asarray(x).view(Tensor)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a = np.asarray(a)
    b = np.asarray(b)
    if transpose_a:
        a = a.T
    if transpose_b:
        b = b.T
    return _wrap(a @ b)


def reduce_logsumexp(x, axis=None, keep_dims=False):
    x = np.asarray(x)
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0)
    r = np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)) + m
    if not keep_dims:
        r = np.squeeze(r, axis=axis)
    return _wrap(r)


def reduce_mean(x, axis=None, keep_dims=False):
    return _wrap(np.mean(np.asarray(x), axis=axis, keepdims=keep_dims))


def reduce_sum(x, axis=None, keep_dims=False):
    return _wrap(np.sum(np.asarray(x), axis=axis, keepdims=keep_dims))


def square(x):
    return _wrap(np.square(np.asarray(x)))


def reshape(x, shape):
    return _wrap(np.reshape(np.asarray(x), shape))


def concat(xs, axis=0):
    return _wrap(np.concatenate([np.asarray(x) for x in xs], axis=axis))


def split(x, n, axis=0):
    return [_wrap(p) for p in np.split(np.asarray(x), n, axis=axis)]


def eye(n):
    return _wrap(np.eye(int(n), dtype=_DTYPE[0]))


def zeros(shape=(), dtype=None, name=None):
    return _wrap(np.zeros(shape, dtype=_DTYPE[0]))


def _softmax(x):
    x = np.asarray(x)
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return _wrap(e / np.sum(e, axis=-1, keepdims=True))


def _xent(labels=None, logits=None):
    logits = np.asarray(logits)
    m = np.max(logits, axis=-1, keepdims=True)
    lse = np.log(np.sum(np.exp(logits - m), axis=-1, keepdims=True)) + m
    return _wrap(-np.sum(np.asarray(labels) * (logits - lse), axis=-1))


@contextlib.contextmanager
def device(_name):
    yield


def install():
    """Register this stand-in as `tensorflow` in sys.modules (build container only)."""
    mod = types.ModuleType("tensorflow")
    mod.float32 = np.float32
    for f in (matmul, reduce_logsumexp, reduce_mean, reduce_sum, square, reshape,
              concat, split, eye, zeros, device):
        setattr(mod, f.__name__, f)
    nn = types.ModuleType("tensorflow.nn")
    nn.softmax = _softmax
    nn.softmax_cross_entropy_with_logits = _xent
    mod.nn = nn
    sys.modules["tensorflow"] = mod
    sys.modules["tensorflow.nn"] = nn
    return mod


def import_reference(path, name):
    """Import one reference source file, unmodified, by path, over the stand-in."""
    import importlib.util
    install()
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
