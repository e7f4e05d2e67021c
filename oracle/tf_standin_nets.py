"""NumPy stand-in for the TensorFlow-1.x symbols the reference's LAYER and MODEL code touches.

TEST INFRASTRUCTURE ONLY (see oracle/tf_standin.py, which this module extends).  It lets the
*unmodified* reference files `utils/nn.py`, `models/dcgan.py` and `models/densenet.py` be
imported by path in the build container and executed eagerly on NumPy arrays, so that

  1. `oracle/nets_torch.py` (our restatement of the layers, models and optimisers) is pinned
     against the reference's own Python control flow: layer order, list/concat order, the
     CReLU interleave, scope/variable naming, weight-norm formula, GLU, the feature head, the
     optimiser update order, and
  2. golden vectors for the nets can be generated (`oracle/make_golden_nets.py`).

What is necessarily restated here rather than executed are the TensorFlow *kernels* themselves
(TF is not installable here): `tf.nn.conv2d` with 'SAME' padding (out = ceil(in/stride),
pad_total = max((out-1)*stride + k - in, 0), pad_before = pad_total // 2), nearest-neighbour
resize by integer factors, `tf.nn.l2_normalize` (x * rsqrt(max(sum x^2, 1e-12))),
`tf.nn.moments`.  The convolution is written as a tap loop over strided slices -- independent
of the torch.nn.functional.conv2d used by oracle/nets_torch.py.

Graph semantics: the reference builds a graph and runs update ops later.  Here everything is
eager; `Variable.assign` mutates only while `EAGER_ASSIGN[0]` is True (optimiser pinning) and
otherwise just returns the would-be value (model building: the data-dependent init assigns of
nn.py:141-160 are never run by train.py).

Variables are created from a name-derived seed so that a test can regenerate multi-megabyte
weight tensors from the fixture's recipe instead of storing them:
    V: 0.05 * RandomState(crc32(name)).standard_normal(shape), rounded to fp32
"""
import contextlib
import sys
import types
import zlib

import numpy as np

from . import tf_standin as base

Tensor = base.Tensor
Tensor.set_shape = lambda self, shape: None
_wrap = base._wrap

EAGER_ASSIGN = [False]
VARS = {}            # full name -> Variable
DRAWS = []           # every tf.random_uniform draw, in call order
_SCOPE = []
_ARGS = []           # arg_scope stack: list of {func: kwargs}
_RNG = [np.random.RandomState(0)]
_NOISE_QUEUE = []    # optional: values to return from random_uniform instead of drawing


def reset(seed=0):
    VARS.clear()
    DRAWS.clear()
    del _SCOPE[:]
    del _ARGS[:]
    del _NOISE_QUEUE[:]
    _RNG[0] = np.random.RandomState(seed)
    EAGER_ASSIGN[0] = False
    del _CREATED[:]
    _REPLAY[0] = None


def seeded_normal(name, shape, std):
    """The recipe tests use to regenerate a variable: fp32 rounding of std * N(0,1)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return (std * rs.standard_normal(tuple(int(s) for s in shape))).astype(np.float32)


class Variable(Tensor):
    name = None

    def assign(self, value):
        value = np.asarray(value)
        if EAGER_ASSIGN[0]:
            self[...] = value
            return self
        return _wrap(np.array(value, dtype=self.dtype))

    def assign_add(self, value):
        return self.assign(np.asarray(self) + value)


def _new_variable(value, name):
    v = np.array(value).view(Variable)
    v.name = name
    return v


_CREATED = []        # tf.Variable objects in creation order (optimiser slots)
_REPLAY = [None]     # when an int: hand the recorded objects out again instead of creating


def replay_variables():
    """Re-running an `*_updates` builder normally creates fresh zero slots; after this call the
    next run receives the SAME slot variables again, in creation order -- the eager equivalent
    of running the reference's update op a second time."""
    _REPLAY[0] = 0


def make_variable(initial_value, *_a, **_k):     # tf.Variable(initial_value, ...)
    if _REPLAY[0] is not None:
        v = _CREATED[_REPLAY[0]]
        _REPLAY[0] += 1
        return v
    v = _new_variable(np.asarray(initial_value, dtype=base._DTYPE[0]), "Variable_%d:0" % len(_CREATED))
    _CREATED.append(v)
    return v


@contextlib.contextmanager
def variable_scope(name, *_a, **_k):
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


def random_normal_initializer(mean=0.0, stddev=1.0):
    def init(name, shape):
        return mean + seeded_normal(name, shape, stddev)
    return init


def zeros_initializer():
    return lambda name, shape: np.zeros(shape, np.float32)


def ones_initializer():
    return lambda name, shape: np.ones(shape, np.float32)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **_k):
    full = "/".join(_SCOPE + [name])
    if full in VARS:
        return VARS[full]
    if shape is None or initializer is None:
        raise ValueError("variable %s does not exist" % full)
    if isinstance(shape, (int, np.integer)):
        shape = [int(shape)]
    val = initializer(full, [int(s) for s in shape]).astype(base._DTYPE[0])
    VARS[full] = _new_variable(val, full + ":0")
    return VARS[full]


def make_template(name, func, **_k):
    def call(*a, **k):
        with variable_scope(name):
            return func(*a, **k)
    return call


# ---- tf.contrib.framework arg_scope -----------------------------------------------------------
def add_arg_scope(func):
    def wrapped(*a, **k):
        kw = {}
        for frame in _ARGS:
            kw.update(frame.get(wrapped, {}))
        kw.update(k)
        return func(*a, **kw)
    wrapped.__name__ = func.__name__
    wrapped._wrapped = func
    return wrapped


@contextlib.contextmanager
def arg_scope(funcs, **kwargs):
    _ARGS.append({f: dict(kwargs) for f in funcs})
    try:
        yield
    finally:
        _ARGS.pop()


# ---- ops ------------------------------------------------------------------------------------
def concat(xs, axis=0):
    if isinstance(xs, np.ndarray):      # tf.concat of a single tensor is the identity
        return _wrap(np.asarray(xs))
    return base.concat(xs, axis)


def reshape(x, shape):
    return _wrap(np.reshape(np.asarray(x), tuple(int(s) for s in shape)))


def reduce_sum(x, axis=None, keep_dims=False):
    if isinstance(axis, list):
        axis = tuple(axis)
    return _wrap(np.sum(np.asarray(x), axis=axis, keepdims=keep_dims))


def reduce_mean(x, axis=None, keep_dims=False):
    if isinstance(axis, list):
        axis = tuple(axis)
    return _wrap(np.mean(np.asarray(x), axis=axis, keepdims=keep_dims))


def random_uniform(shape, minval=0.0, maxval=1.0, **_k):
    if _NOISE_QUEUE:
        u = np.asarray(_NOISE_QUEUE.pop(0), dtype=base._DTYPE[0])
        assert u.shape == tuple(shape)
    else:
        u = (_RNG[0].random_sample(tuple(shape)) * (maxval - minval) + minval).astype(np.float32)
        u = u.astype(base._DTYPE[0])
    DRAWS.append(np.array(u))
    return _wrap(u)


def same_pads(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def nn_conv2d(x, W, strides, padding):
    assert padding == "SAME" and strides[0] == 1 and strides[3] == 1
    x = np.asarray(x)
    W = np.asarray(W)
    sh, sw = int(strides[1]), int(strides[2])
    kh, kw, cin, cout = W.shape
    n, h, w, c = x.shape
    assert c == cin, (x.shape, W.shape)
    oh, pt, pb = same_pads(h, kh, sh)
    ow, pl, pr = same_pads(w, kw, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, cout), dtype=np.result_type(x, W))
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (oh - 1) * sh + 1:sh, j:j + (ow - 1) * sw + 1:sw, :]
            y += np.tensordot(patch, W[i, j], axes=([3], [0]))
    return _wrap(y)


def resize_nearest_neighbor(x, size):
    x = np.asarray(x)
    fh, fw = int(size[0]) // x.shape[1], int(size[1]) // x.shape[2]
    assert fh * x.shape[1] == size[0] and fw * x.shape[2] == size[1]
    return _wrap(np.repeat(np.repeat(x, fh, axis=1), fw, axis=2))


def l2_normalize(x, axis, epsilon=1e-12):
    x = np.asarray(x)
    ss = np.sum(np.square(x), axis=tuple(int(a) for a in np.atleast_1d(axis)), keepdims=True)
    return _wrap(x / np.sqrt(np.maximum(ss, epsilon)))


def moments(x, axes):
    x = np.asarray(x)
    ax = tuple(int(a) for a in axes)
    return _wrap(np.mean(x, axis=ax)), _wrap(np.var(x, axis=ax))


def _unary(f):
    return lambda x, *a, **k: _wrap(f(np.asarray(x)))


def elu(x):
    x = np.asarray(x)
    return _wrap(np.where(x > 0, x, np.expm1(np.minimum(x, 0))))


def sigmoid(x):
    return _wrap(1.0 / (1.0 + np.exp(-np.asarray(x))))


def group(*ops):
    return list(ops)


@contextlib.contextmanager
def control_dependencies(_deps):
    yield


def gradients(*_a, **_k):
    raise NotImplementedError("the stand-in runs forward passes and explicit-gradient optimisers only")


def install():
    """Register the extended stand-in as `tensorflow` (+ the contrib/python submodules the
    reference imports) in sys.modules."""
    mod = base.install()
    mod.float32 = np.float32
    mod.Variable = make_variable
    mod.get_variable = get_variable
    mod.variable_scope = variable_scope
    mod.make_template = make_template
    mod.random_normal_initializer = random_normal_initializer
    mod.zeros_initializer = zeros_initializer
    mod.ones_initializer = ones_initializer
    mod.concat = concat
    mod.reshape = reshape
    mod.reduce_sum = reduce_sum
    mod.reduce_mean = reduce_mean
    mod.random_uniform = random_uniform
    mod.sqrt = _unary(np.sqrt)
    mod.abs = _unary(np.abs)
    mod.pow = lambda a, b: _wrap(np.power(a, np.asarray(b)))
    mod.maximum = lambda a, b: _wrap(np.maximum(a, b))
    mod.stop_gradient = lambda x: x
    mod.group = group
    mod.control_dependencies = control_dependencies
    mod.gradients = gradients
    mod.nn.relu = _unary(lambda v: np.maximum(v, 0))
    mod.nn.elu = elu
    mod.nn.sigmoid = sigmoid
    mod.nn.tanh = _unary(np.tanh)
    mod.nn.l2_normalize = l2_normalize
    mod.nn.moments = moments
    mod.nn.conv2d = nn_conv2d
    mod.nn.bias_add = lambda x, b: _wrap(np.asarray(x) + np.asarray(b))
    image = types.ModuleType("tensorflow.image")
    image.resize_nearest_neighbor = resize_nearest_neighbor
    mod.image = image
    sys.modules["tensorflow.image"] = image
    # import paths used by the reference: tensorflow.contrib.framework.python.ops.{arg_scope,
    # add_arg_scope} and tensorflow.python.framework.function (imported, never used)
    chain = ["tensorflow.contrib", "tensorflow.contrib.framework", "tensorflow.contrib.framework.python",
             "tensorflow.contrib.framework.python.ops", "tensorflow.python", "tensorflow.python.framework",
             "tensorflow.python.framework.function"]
    for name in chain:
        m = types.ModuleType(name)
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules[parent], leaf, m)
    ops = sys.modules["tensorflow.contrib.framework.python.ops"]
    ops.arg_scope = arg_scope
    ops.add_arg_scope = add_arg_scope
    return mod


def import_reference_nets(ref_root):
    """Import utils/nn.py, models/dcgan.py, models/densenet.py of the reference, unmodified."""
    import importlib.util
    install()

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    import os
    utils_pkg = types.ModuleType("utils")
    utils_pkg.__path__ = []
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.nn")}
    sys.modules["utils"] = utils_pkg
    nn = load(os.path.join(ref_root, "utils", "nn.py"), "utils.nn")
    utils_pkg.nn = nn
    dcgan = load(os.path.join(ref_root, "models", "dcgan.py"), "ref_models_dcgan")
    densenet = load(os.path.join(ref_root, "models", "densenet.py"), "ref_models_densenet")
    for k, v in saved.items():       # do not leave a fake `utils` package behind
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return nn, dcgan, densenet
