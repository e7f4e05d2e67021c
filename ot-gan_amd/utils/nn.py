"""Layers, parameter handling and optimisers -- the MI355X host-side mirror of the
reference's `utils/nn.py` (same public names and argument meaning):

    conv2d, dense           nn.py:327-338, 314-325  (weight norm, list inputs, pre-activation)
    arg_scope               tensorflow.contrib.framework.arg_scope as used by models/*.py
    make_template           tf.make_template (shared parameters across calls)
    ExponentialMovingAverage  tf.train.ExponentialMovingAverage as used in train.py:63-64
    adam_updates, adamax_updates, nesterov_updates   nn.py:29-87

Tensors are NHWC float32 CUDA tensors; all arithmetic runs in the HIP library through
`otgan_amd.ops` (torch.autograd Functions).  Parameters are created on first use with the
reference's *effective* initialisation (the data-dependent init tensors of
nn.py:133-162 are built but never run by train.py, so V ~ N(0, 0.05), g = 1, b = 0).
"""
import contextlib
import os
import threading
from collections import OrderedDict

import torch

from .. import ops

# --------------------------------------------------------------------------------- arg_scope
_scope = threading.local()


def _stack():
    if not hasattr(_scope, "stack"):
        _scope.stack = []
    return _scope.stack


@contextlib.contextmanager
def arg_scope(funcs, **kwargs):
    """Default keyword arguments for the listed layer functions inside the `with` block."""
    _stack().append(({f.__name__ for f in funcs}, kwargs))
    try:
        yield
    finally:
        _stack().pop()


def _scoped(fn):
    def wrapper(*args, **kwargs):
        merged = {}
        for names, kw in _stack():
            if fn.__name__ in names:
                merged.update(kw)
        merged.update(kwargs)
        return fn(*args, **merged)
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


# --------------------------------------------------------------------------------- variables
class VariableStore:
    """Named trainable tensors of one template ('discriminator/conv2d_0/V', ...), in creation
    order (V, g, b per layer -- the order of the reference's first, init=True, call)."""

    def __init__(self, scope, device=None, seed=1):
        self.scope = scope
        self.device = device
        self.vars = OrderedDict()
        self._gen = None
        self._seed = seed

    def _generator(self):
        if self._gen is None:
            self._gen = torch.Generator(device="cpu")
            self._gen.manual_seed(self._seed + sum(map(ord, self.scope)))
        return self._gen

    def get(self, name, shape, kind, device):
        full = f"{self.scope}/{name}"
        v = self.vars.get(full)
        if v is None:
            if kind == "normal":      # tf.random_normal_initializer(0, 0.05)   nn.py:124
                t = torch.empty(shape, dtype=torch.float32).normal_(0.0, 0.05, generator=self._generator())
            elif kind == "ones":      # nn.py:143
                t = torch.ones(shape, dtype=torch.float32)
            else:                     # zeros, nn.py:160
                t = torch.zeros(shape, dtype=torch.float32)
            v = t.to(self.device or device).requires_grad_(True)
            self.vars[full] = v
        elif tuple(v.shape) != tuple(shape):
            raise ValueError(f"variable {full} has shape {tuple(v.shape)}, requested {tuple(shape)}")
        return v


class FlatGroup:
    """All variables of a template re-homed as views of ONE flat buffer, so that the optimiser
    and the EMA run as a single kernel launch over the whole parameter set instead of one launch
    per tensor (DenseNet has ~300 tensors per network).  Tensors whose size is not a multiple of
    four floats (the RGB layer's g and b) go last so that every other view stays 16-byte aligned."""

    _by_param = {}

    def __init__(self, params):
        order = [p for p in params if p.numel() % 4 == 0] + [p for p in params if p.numel() % 4 != 0]
        self.params = order
        self.total = sum(p.numel() for p in order)
        self.flat = torch.empty(self.total, dtype=order[0].dtype, device=order[0].device)
        off = 0
        with torch.no_grad():
            for p in order:
                n = p.numel()
                view = self.flat[off:off + n].view(p.shape)
                view.copy_(p)
                p.data = view
                off += n
        for p in order:
            FlatGroup._by_param[id(p)] = self

    @staticmethod
    def of(params):
        """The group whose members are exactly `params` (any order), else None."""
        params = list(params)
        grp = FlatGroup._by_param.get(id(params[0])) if params else None
        if grp is None or len(params) != len(grp.params):
            return None
        ids = {id(p) for p in params}
        return grp if all(id(p) in ids for p in grp.params) else None

    def offsets(self):
        """Start of every member inside the flat buffer, in buffer order, plus the total."""
        out, off = [], 0
        for p in self.params:
            out.append(off)
            off += p.numel()
        return out + [off]

    def in_buffer_order(self, tensors, params):
        """Per-parameter tensors (given in the order of `params`) re-ordered to buffer order, contiguous."""
        pos = {id(p): i for i, p in enumerate(params)}
        return [tensors[pos[id(p)]].contiguous() for p in self.params]

    def flatten_like(self, tensors, params):
        """Concatenate per-parameter tensors (given in the order of `params`) in buffer order."""
        pos = {id(p): i for i, p in enumerate(params)}
        return torch.cat([tensors[pos[id(p)]].reshape(-1) for p in self.params])

    def views_of(self, flat):
        out, off = {}, 0
        for p in self.params:
            out[id(p)] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return out


_current = threading.local()


def _store():
    st = getattr(_current, "store", None)
    if st is None:
        raise RuntimeError("layers must be called inside a template (nn.make_template)")
    return st


class Template:
    """Callable with shared parameters -- the role tf.make_template plays in models/*.py."""

    def __init__(self, name, fn, seed=1):
        self.name = name
        self.fn = fn
        self.store = VariableStore(name, seed=seed)

    def __call__(self, *args, **kwargs):
        prev = getattr(_current, "store", None)
        _current.store = self.store
        try:
            return self.fn(*args, **kwargs)
        finally:
            _current.store = prev

    def trainable_variables(self):
        return list(self.store.vars.values())

    def named_variables(self):
        return OrderedDict(self.store.vars)

    def reset(self, seed=1, device=None):
        self.store = VariableStore(self.name, device=device, seed=seed)

    def flatten(self):
        """Re-home the variables in one flat buffer (see FlatGroup); call after the creation pass."""
        return FlatGroup(self.trainable_variables())


def make_template(name, fn, seed=1):
    return Template(name, fn, seed)


class ExponentialMovingAverage:
    """shadow <- decay*shadow + (1-decay)*param   (train.py:63-64; shadows start at the
    parameter values like tf.train.ExponentialMovingAverage)."""

    def __init__(self, decay=0.999):
        self.decay = decay
        self._shadow = {}
        self._params = []

    def apply(self, params):
        params = list(params)
        grp = FlatGroup.of(params) if not self._params else None
        if grp is not None:      # one launch over the whole parameter set
            self._flat = (grp.flat.detach().clone(), grp)
            self._shadow.update(grp.views_of(self._flat[0]))
            self._params.extend(params)
            return self.update
        for p in params:
            if id(p) not in self._shadow:
                self._shadow[id(p)] = p.detach().clone()
                self._params.append(p)
        return self.update

    def update(self):
        flat = getattr(self, "_flat", None)
        if flat is not None:
            ops.ema_update(flat[0], flat[1].flat, self.decay)
            return
        for p in self._params:
            ops.ema_update(self._shadow[id(p)], p.detach(), self.decay)

    def average(self, p):
        return self._shadow[id(p)]

    def state(self):
        return [self._shadow[id(p)] for p in self._params]


def get_var_maybe_avg(name, shape, kind, ema, device):
    """nn.py:89-93"""
    v = _store().get(name, shape, kind, device)
    return ema.average(v) if ema is not None else v


def get_name(layer_name, counters):
    """nn.py:95-100"""
    if layer_name not in counters:
        counters[layer_name] = 0
    name = f"{layer_name}_{counters[layer_name]}"
    counters[layer_name] += 1
    return name


def _as_list(x):
    if isinstance(x, tuple):
        return list(x)
    if not isinstance(x, list):
        return [x]
    return x


def _channels_eff(xs, pre_activation):
    c = sum(int(t.shape[-1]) for t in xs)
    return c * (2 if pre_activation in ("celu", "crelu") else 1), c


# --------------------------------------------------------------------------------- data-dependent init
# The reference builds a data-dependent initialisation pass (nn.py:133-162: g <- init_scale / std, b <- -mean * g of
# every layer's output on an initial batch, layer by layer) but its train.py never runs it, so the
# default here is the EFFECTIVE behaviour g = 1, b = 0.  `data_dependent_init(True)` (train.py --data_dependent_init)
# makes an `init=True` call execute that pass: the statistics are taken from the HIP forward of the layer.
_DATA_INIT = [False]


def data_dependent_init(on=True):
    _DATA_INIT[0] = bool(on)


def _run_data_init(y0, g, b, init_scale):
    """nn.py:137-161: moments over all axes but the last of y0 = f(x, l2_normalize(V)); assigns g and b in place."""
    with torch.no_grad():
        flat = y0.reshape(-1, y0.shape[-1]).double()
        m = flat.mean(0)
        v = flat.var(0, unbiased=False)
        gi = init_scale / torch.sqrt(v)
        g.copy_(gi.to(g.dtype))
        b.copy_((-m * gi).to(b.dtype))
    ops.bump_weights_epoch()


# --------------------------------------------------------------------------------- layers
@_scoped
def dense(x, num_units, pre_activation='celu', init_scale=1., counters={}, init=False, ema=None,
          weight_norm=True, use_b=True, use_g=True, **kwargs):
    """Fully connected layer on a [B, C] tensor or list of tensors (reference nn.py:314-325)."""
    if not (weight_norm and use_g and use_b):
        raise NotImplementedError("only the weight_norm=True, use_g=True, use_b=True layer of the "
                                  "reference models is implemented")
    name = get_name('dense', counters)
    xs = _as_list(x)
    nr_in, c = _channels_eff(xs, pre_activation)
    dev = xs[0].device
    V = get_var_maybe_avg(f"{name}/V", (nr_in, num_units), "normal", ema, dev)
    g = get_var_maybe_avg(f"{name}/g", (num_units,), "ones", ema, dev)
    b = get_var_maybe_avg(f"{name}/b", (num_units,), "zeros", ema, dev)
    xin = xs[0] if len(xs) == 1 else torch.cat(xs, 1)
    if init and _DATA_INIT[0] and ema is None:
        with torch.no_grad():
            y0 = ops.dense_op(xin, V, torch.ones_like(g), torch.zeros_like(b), preact=ops.ACT[pre_activation],
                              segs=[int(t.shape[-1]) for t in xs])
        _run_data_init(y0, g, b, init_scale)
    return ops.dense_op(xin, V, g, b, preact=ops.ACT[pre_activation], segs=[int(t.shape[-1]) for t in xs])


@_scoped
def conv2d(x, num_filters, pre_activation='celu', filter_size=[3, 3], stride=[1, 1], pad='SAME',
           dilate=1, upsample=False, init_scale=1., counters={}, init=False, ema=None,
           weight_norm=True, use_b=True, use_g=True, glu_hint=False, grow=0, **kwargs):
    """2-D convolution on an NHWC tensor or list of tensors (reference nn.py:327-338):
    optional 2x nearest-neighbour upsample, pre-activation over the list, weight-normalised
    HWIO filter, TF 'SAME' padding, bias.  `glu_hint` (not in the reference): the caller passes the
    result to glu() next, see ops.conv2d_op -- values unchanged.  `grow` (not in the reference): the caller appends that
    many channels to the result next (dense_block: its list input + L * F outputs): the result is allocated as the
    channel prefix of a buffer with room for them, so the block grows there without copying its input -- values unchanged."""
    if pad != 'SAME' or dilate != 1:
        raise NotImplementedError("the reference models only use pad='SAME', dilate=1")
    if not (weight_norm and use_g and use_b):
        raise NotImplementedError("only the weight_norm=True, use_g=True, use_b=True layer of the "
                                  "reference models is implemented")
    if stride[0] != stride[1]:
        raise NotImplementedError("square strides only")
    name = get_name('conv2d', counters)
    xs = _as_list(x)
    nr_in, c = _channels_eff(xs, pre_activation)
    dev = xs[0].device
    V = get_var_maybe_avg(f"{name}/V", (filter_size[0], filter_size[1], nr_in, num_filters), "normal", ema, dev)
    g = get_var_maybe_avg(f"{name}/g", (num_filters,), "ones", ema, dev)
    b = get_var_maybe_avg(f"{name}/b", (num_filters,), "zeros", ema, dev)
    if isinstance(xs, ConcatList) and xs.buffer is not None:
        xin = xs.buffer
    else:
        xin = xs[0] if len(xs) == 1 else torch.cat(xs, 3)
    if init and _DATA_INIT[0] and ema is None:
        with torch.no_grad():
            y0 = ops.conv2d_op(xin, V, torch.ones_like(g), torch.zeros_like(b), stride=stride[0], upsample=upsample,
                               preact=ops.ACT[pre_activation], segs=[int(t.shape[-1]) for t in xs])
        _run_data_init(y0, g, b, init_scale)
    return ops.conv2d_op(xin, V, g, b, stride=stride[0], upsample=upsample,
                         preact=ops.ACT[pre_activation], segs=[int(t.shape[-1]) for t in xs], glu_hint=glu_hint,
                         grow=grow)


class ConcatList(list):
    """A list of NHWC tensors that are channel slices of ONE buffer (`.buffer`), in order --
    what a DenseNet block returns.  It behaves as the reference's Python list (indexable,
    usable as the input of conv2d) while letting the next layer read the concatenation without
    a copy."""
    buffer = None


@_scoped
def dense_block(x, layers_per_block, filters_per_layer, pre_activation='celu', filter_size=[3, 3],
                counters={}, init=False, ema=None, weight_norm=True, **kwargs):
    """`for rep in range(L): x.append(conv2d(x, F, pre_activation))` of the reference
    (models/densenet.py:11-16, 60-65) as one in-place growing block.  Layer names / variables
    are the same `conv2d_<k>/{V,g,b}` the loop would have created."""
    xs = _as_list(x)
    if init and _DATA_INIT[0] and ema is None:
        # data-dependent init: every layer's statistics depend on the initialised layers before it -- the plain loop
        # of the reference (models/densenet.py:11-16), one conv2d per layer (initialisation only)
        feats = list(xs)
        for _ in range(layers_per_block):
            feats.append(conv2d(feats, filters_per_layer, pre_activation=pre_activation, filter_size=filter_size,
                                counters=counters, init=True, ema=None, weight_norm=weight_norm))
        return feats
    dev = xs[0].device
    segs0 = [int(t.shape[-1]) for t in xs]
    if isinstance(xs, ConcatList) and xs.buffer is not None:
        x0 = xs.buffer
    elif len(xs) == 1:
        x0 = xs[0]
    else:
        # the first element inside a buffer with room for the rest of the list and the block's outputs (conv2d grow=):
        # the other elements are copied behind it there; else one concatenation
        x0 = ops.extend_channels(xs, sum(segs0) + layers_per_block * filters_per_layer)
        if x0 is None:
            x0 = ops.concat_channels(xs)
    mult = 2 if pre_activation in ("celu", "crelu") else 1
    params = []
    c = sum(segs0)
    for _ in range(layers_per_block):
        name = get_name('conv2d', counters)
        V = get_var_maybe_avg(f"{name}/V", (filter_size[0], filter_size[1], c * mult, filters_per_layer),
                              "normal", ema, dev)
        g = get_var_maybe_avg(f"{name}/g", (filters_per_layer,), "ones", ema, dev)
        b = get_var_maybe_avg(f"{name}/b", (filters_per_layer,), "zeros", ema, dev)
        params.append((V, g, b))
        c += filters_per_layer
    buf = ops.dense_block_op(x0, segs0, params, ksize=filter_size[0], preact=ops.ACT[pre_activation])
    out = ConcatList(torch.split(buf, segs0 + [filters_per_layer] * layers_per_block, dim=3))
    out.buffer = buf
    return out


def feature_head(x):
    """CReLU -> flatten -> L2 row normalisation (models/dcgan.py:16-19)."""
    return ops.feature_head(x)


def glu(x):
    """x[..., :C] * sigmoid(x[..., C:])  ("gated linear unit", models/dcgan.py:35-36)."""
    return ops.glu(x)


def tanh(x):
    return ops.tanh(x)


# --------------------------------------------------------------------------------- optimisers
class _Updates:
    """State + step of one optimiser over a fixed parameter list; gradients are supplied per
    step as a list (the reference passes precomputed gradient lists, nn.py:52-55)."""

    def __init__(self, params, lr, mom1, mom2):
        self.params = list(params)
        self.lr, self.mom1, self.mom2 = lr, mom1, mom2
        self.group = FlatGroup.of(self.params)
        self.state = [{}] if self.group is not None else [{} for _ in self.params]
        self.t = 1.0

    def fuse_ema(self, ema):
        """Let the step also update `ema`'s shadows of these parameters (same launch).  True if it will; the caller then
        skips ema.update().  Only the flat-buffer Adam takes it."""
        return False

    def __call__(self, grads, lr=None):
        lr = self.lr if lr is None else lr
        with torch.no_grad():
            if self.group is not None and self._step_gather(list(grads), lr):
                pass                        # one launch straight from the per-variable gradient tensors
            elif self.group is not None:    # the update is elementwise: run it on the flat buffer
                self._step(self.group.flat, self.group.flatten_like(list(grads), self.params), self.state[0], lr)
            else:
                for p, g, st in zip(self.params, grads, self.state):
                    self._step(p, g.contiguous(), st, lr)
        self.t += 1.0

    # checkpointing (the reference saves no optimiser state, train.py:60; SURVEY 8f-3 asks for it)
    def state_dict(self):
        return {"t": self.t, "slots": [{k: (None if v is None else v.detach().cpu()) for k, v in st.items()}
                                       for st in self.state]}

    def load_state_dict(self, sd):
        self.t = float(sd["t"])
        targets = [self.group.flat] if self.group is not None else self.params
        if len(sd["slots"]) != len(self.state):
            raise ValueError("optimiser state does not match this parameter layout")
        for st, saved, p in zip(self.state, sd["slots"], targets):
            st.clear()
            for k, v in saved.items():
                st[k] = None if v is None else v.to(p.device).clone()


    def _step_gather(self, grads, lr):
        return False


class _Adam(_Updates):
    _ema = None

    def fuse_ema(self, ema):
        flat = getattr(ema, "_flat", None)
        if self.group is None or flat is None or flat[1] is not self.group or len(self.params) > ops.ADAM_MAX_SEGMENTS:
            return False
        self._ema = ema
        return True

    def _step_gather(self, grads, lr):
        if len(self.params) > ops.ADAM_MAX_SEGMENTS:
            if self._ema is not None:       # (the caller was told the shadows are ours)
                raise RuntimeError("fused EMA needs the gathered Adam step")
            return False
        st, p = self.state[0], self.group.flat
        if not st:
            st["mg"] = torch.zeros_like(p)
            st["v"] = torch.zeros_like(p) if self.mom1 > 0 else None
        ema = self._ema
        ops.adam_step_gather(p, self.group.in_buffer_order(grads, self.params), self.group.offsets(), st["v"], st["mg"],
                             lr, self.mom1, self.mom2, self.t,
                             ema._flat[0] if ema is not None else None, ema.decay if ema is not None else 0.0,
                             coef=self.coef_dev if self.capturing else None)
        return True

    coef_dev = None      # device [2]: the step's bias corrections, written by the trainer before a captured step is replayed
    capturing = False    # True while the trainer records a step into a hipGraph: the launch then reads coef_dev

    def ensure_state(self):
        """Allocate the moment buffers now (a captured step must not be the one that creates them)."""
        if self.group is not None and not self.state[0]:
            p = self.group.flat
            self.state[0]["mg"] = torch.zeros_like(p)
            self.state[0]["v"] = torch.zeros_like(p) if self.mom1 > 0 else None

    def write_coefficients(self):
        """The bias corrections of the NEXT step (self.t) into coef_dev: the same fp32 values the eager step derives."""
        c1, c2 = ops.adam_coefficients(self.mom1, self.mom2, self.t)
        self.coef_dev.copy_(torch.tensor([c1, c2], dtype=torch.float32))

    def _step(self, p, g, st, lr):
        if not st:
            st["mg"] = torch.zeros_like(p)
            st["v"] = torch.zeros_like(p) if self.mom1 > 0 else None
        ops.adam_step(p, g, st["v"], st["mg"], lr, self.mom1, self.mom2, self.t, coef=self.coef_dev if self.capturing else None)


class _Adamax(_Updates):
    def _step(self, p, g, st, lr):
        if not st:
            st["mg"] = torch.zeros_like(p)
            st["v"] = torch.zeros_like(p) if self.mom1 > 0 else None
        ops.adamax_step(p, g, st["v"], st["mg"], lr, self.mom1, self.mom2)


class _Nesterov(_Updates):
    def _step(self, p, g, st, lr):
        if not st:
            st["v"] = torch.zeros_like(p)
        ops.nesterov_step(p, g, st["v"], lr, self.mom1)


def _maybe_apply(upd, cost_or_grads):
    if isinstance(cost_or_grads, (list, tuple)):
        upd(cost_or_grads)
    elif cost_or_grads is not None:
        upd(torch.autograd.grad(cost_or_grads, upd.params))
    return upd


def adam_updates(params, cost_or_grads=None, lr=0.001, mom1=0.9, mom2=0.999):
    """Adam as the reference writes it (nn.py:50-73): epsilon inside the square root, one
    step counter per optimiser starting at 1.  Returns the update callable `upd(grads, lr)`;
    if `cost_or_grads` is given one step is applied immediately."""
    return _maybe_apply(_Adam(params, lr, mom1, mom2), cost_or_grads)


def adamax_updates(params, cost_or_grads=None, lr=0.001, mom1=0.9, mom2=0.999):
    """nn.py:29-48"""
    return _maybe_apply(_Adamax(params, lr, mom1, mom2), cost_or_grads)


def nesterov_updates(params, cost_or_grads=None, lr=0.01, mom1=0.9):
    """nn.py:75-87"""
    return _maybe_apply(_Nesterov(params, lr, mom1, None), cost_or_grads)
