"""PyTorch restatement of the reference's layers and models (CPU or GPU, any dtype).

TEST INFRASTRUCTURE ONLY -- the "plain PyTorch reference of the same op" the HIP layer
kernels are compared against (fp64 on CPU, or fp32 on the GPU), and the nets of the
`cpu_baseline` leg of bench.py.  Never imported by the product path (ot-gan_amd/).

An independent restatement (torch.nn.functional.conv2d etc.) of:
    utils/nn.py:103-183   get_params (weight norm, non-init branch)
    utils/nn.py:190-206   apply_pre_activation (list interleave [x0,-x0,x1,-x1,...])
    utils/nn.py:234-241   conv (NHWC x HWIO, TF 'SAME', optional 2x NN upsample first)
    utils/nn.py:314-338   dense / conv2d (+ bias)
    utils/nn.py:29-87     adam / adamax / nesterov updates
    models/dcgan.py:7-52, models/densenet.py:7-88
PINNED: tests/test_oracle_nets.py checks this file against golden vectors produced by running
the reference's own utils/nn.py, models/dcgan.py and models/densenet.py unmodified over a NumPy
stand-in for TensorFlow (oracle/tf_standin_nets.py, oracle/make_golden_nets.py): features,
images, variable names/shapes and three steps of every optimiser, to 1e-12 (fp64) / 2e-7 (fp32).

Activations are NHWC like the reference; torch's conv wants NCHW, so tensors are permuted
around F.conv2d.
"""
import math

import torch
import torch.nn.functional as F


def apply_pre_activation(xs, pre, axis=3):
    """nn.py:190-206"""
    if not isinstance(xs, (list, tuple)):
        xs = [xs]
    if pre is None:
        return torch.cat(list(xs), axis)
    if pre == "celu":
        return F.elu(torch.cat([s for x in xs for s in (x, -x)], axis))
    if pre == "crelu":
        return F.relu(torch.cat([s for x in xs for s in (x, -x)], axis))
    if pre == "elu":
        return F.elu(torch.cat(list(xs), axis))
    if pre == "relu":
        return F.relu(torch.cat(list(xs), axis))
    raise ValueError("unsupported pre-activation")


def weight_norm(V, g):
    """nn.py:176-181: W = l2_normalize(V, all axes but the last) * g  (epsilon 1e-12 under max)"""
    axes = tuple(range(V.dim() - 1))
    ss = (V * V).sum(axes, keepdim=True)
    return V * torch.rsqrt(torch.clamp(ss, min=1e-12)) * g


def tf_same_pads(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d_nhwc(x, W, stride):
    """tf.nn.conv2d(x, W, [1,s,s,1], 'SAME') with NHWC x and HWIO W (nn.py:241)."""
    kh, kw = W.shape[0], W.shape[1]
    pt, pb = tf_same_pads(x.shape[1], kh, stride)
    pl, pr = tf_same_pads(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, W.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1)


def upsample2(x):
    """tf.image.resize_nearest_neighbor(x, [2H, 2W]) -- exact 2x replication"""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def _data_init(y0, p, init_scale):
    """nn.py:137-161 (the init=True branch of get_params): y0 = f(x, l2_normalize(V)); moments over all axes but the
    last; g <- init_scale / sqrt(var), b <- -mean * g  (assigned in place)."""
    axes = tuple(range(y0.dim() - 1))
    m = y0.mean(axes)
    v = y0.var(axes, unbiased=False)
    with torch.no_grad():
        g = init_scale / torch.sqrt(v)
        p["g"].copy_(g)
        p["b"].copy_(-m * g)


def conv2d(xs, p, pre=None, stride=1, upsample=False, init=False, init_scale=1.0):
    """nn.py:327-338 with weight norm.  p = {'V','g','b'}; init: the data-dependent initialisation pass"""
    if not isinstance(xs, (list, tuple)):
        xs = [xs]
    if upsample:
        xs = [upsample2(torch.cat(list(xs), 3))]
    x = apply_pre_activation(xs, pre, 3)
    if init:
        with torch.no_grad():
            _data_init(conv2d_nhwc(x, weight_norm(p["V"], torch.ones_like(p["g"])), stride), p, init_scale)
    y = conv2d_nhwc(x, weight_norm(p["V"], p["g"]), stride)
    return y + p["b"]


def dense(x, p, pre=None, init=False, init_scale=1.0):
    """nn.py:314-325"""
    x = apply_pre_activation(x, pre, 1)
    if init:
        with torch.no_grad():
            _data_init(x @ weight_norm(p["V"], torch.ones_like(p["g"])), p, init_scale)
    return x @ weight_norm(p["V"], p["g"]) + p["b"]


# Tests only (tests/test_train_step_gpu.py): a list of sign tensors (-1 / 0 / +1); while it is set, each feature_head() call takes the
# sign pattern of its CReLU from the next entry (the pattern the path under test saw) instead of from its own x.  The two
# differ only where |x| is at rounding level, so the VALUE moves by rounding-level amounts -- but a gradient comparison no
# longer depends on which side of zero such a unit fell (one such unit moves every gradient of a step by 2e-4 .. 5e-4).
FORCED_HEAD_SIGNS = None
# per forced call: (units whose forced sign differs from sign(x) of this evaluation, largest |x| among them relative to the
# RMS of its sample's pre-activations) -- the tests bound both, so that a forward kernel flipping many or large units fails
FORCED_HEAD_REPORT = []


def feature_head(x):
    """models/dcgan.py:16-19"""
    if FORCED_HEAD_SIGNS is not None:
        sign = FORCED_HEAD_SIGNS.pop(0).to(x.device)      # -1 / 0 / +1: relu'(0) = 0 on both halves, as in the reference
        with torch.no_grad():
            differ = sign.to(x.dtype) != torch.sign(x)
            rms = x.reshape(x.shape[0], -1).pow(2).mean(1).sqrt().reshape(-1, 1, 1, 1)
            rel = (x.abs() / rms)[differ]
            FORCED_HEAD_REPORT.append((int(differ.sum()), float(rel.max()) if rel.numel() else 0.0))
        zero = torch.zeros_like(x)
        x = torch.cat([torch.where(sign > 0, x, zero), torch.where(sign < 0, -x, zero)], 3)
    else:
        x = torch.cat([F.relu(x), F.relu(-x)], 3)
    x = x.reshape(x.shape[0], -1)
    return x / torch.sqrt((x * x).sum(1, keepdim=True))


def glu(x, axis):
    a, l = torch.chunk(x, 2, axis)
    return a * torch.sigmoid(l)


# ----------------------------------------------------------------------------- parameter shapes
def _crelu_mult(pre):
    return 2 if pre in ("crelu", "celu") else 1


def dcgan_disc_shapes(nonlinearity="crelu"):
    m = _crelu_mult(nonlinearity)
    return [("conv2d_0", (5, 5, 3, 128)), ("conv2d_1", (5, 5, 128 * m, 256)),
            ("conv2d_2", (5, 5, 256 * m, 512)), ("conv2d_3", (5, 5, 512 * m, 1024))]


def dcgan_gen_shapes(image_size=32):
    base = image_size // 8        # image_size other than 32: the build's added option (reference hard-codes 32)
    return [("dense_0", (100, 2 * base * base * 1024)), ("conv2d_0", (5, 5, 1024, 1024)),
            ("conv2d_1", (5, 5, 512, 512)), ("conv2d_2", (5, 5, 256, 256)),
            ("conv2d_3", (5, 5, 128, 3))]


def densenet_disc_shapes(nonlinearity="crelu", L=16, Fg=16):
    m = _crelu_mult(nonlinearity)
    shapes = [("conv2d_0", (3, 3, 3, 2 * Fg))]
    c, k = 2 * Fg, 1
    for _blk in range(3):
        for _ in range(L):
            shapes.append((f"conv2d_{k}", (3, 3, c * m, Fg)))
            c += Fg
            k += 1
        shapes.append((f"conv2d_{k}", (3, 3, c * m, c // 2)))
        c //= 2
        k += 1
    return shapes


def densenet_gen_shapes(nonlinearity="crelu", L=16, Fg=16):
    m = _crelu_mult(nonlinearity)
    shapes = [("dense_0", (100, 8 * 8 * Fg))]
    c, k = 2 * Fg, 0
    for blk in range(3):
        for _ in range(L):
            shapes.append((f"conv2d_{k}", (3, 3, c * m, Fg)))
            c += Fg
            k += 1
        if blk < 2:
            shapes.append((f"conv2d_{k}", (3, 3, c * m, c // 2)))
            c = c // 2 + Fg
            k += 1
    shapes.append((f"conv2d_{k}", (3, 3, c * m, 3)))
    return shapes


def init_params(shapes, scope, gen, dtype=torch.float32, device="cpu"):
    """Effective init of the reference: V ~ N(0, 0.05), g = 1, b = 0."""
    params = {}
    for name, shp in shapes:
        V = torch.empty(shp, dtype=dtype).normal_(0.0, 0.05, generator=gen)
        params[f"{scope}/{name}"] = {"V": V.to(device), "g": torch.ones(shp[-1], dtype=dtype, device=device),
                                     "b": torch.zeros(shp[-1], dtype=dtype, device=device)}
    return params


# ----------------------------------------------------------------------------- DCGAN
def dcgan_discriminator(x, P, nonlinearity="crelu", scope="discriminator", init=False):
    """models/dcgan.py:7-22"""
    x = conv2d(x, P[f"{scope}/conv2d_0"], None, init=init)
    x = conv2d(x, P[f"{scope}/conv2d_1"], nonlinearity, 2, init=init)
    x = conv2d(x, P[f"{scope}/conv2d_2"], nonlinearity, 2, init=init)
    x = conv2d(x, P[f"{scope}/conv2d_3"], nonlinearity, 2, init=init)
    return feature_head(x)


def dcgan_generator(u, P, scope="generator", init=False):
    """models/dcgan.py:28-52; u: [B,100] uniform(-1,1) noise"""
    B = u.shape[0]
    base = int(round(math.sqrt(P[f"{scope}/dense_0"]["V"].shape[1] // 2048)))   # 4 for 32x32 (reference), 8 for 64x64
    x = glu(dense(u, P[f"{scope}/dense_0"], None, init=init), 1).reshape(B, base, base, 1024)
    x = glu(conv2d(x, P[f"{scope}/conv2d_0"], None, 1, True, init=init), 3)
    x = glu(conv2d(x, P[f"{scope}/conv2d_1"], None, 1, True, init=init), 3)
    x = glu(conv2d(x, P[f"{scope}/conv2d_2"], None, 1, True, init=init), 3)
    return torch.tanh(conv2d(x, P[f"{scope}/conv2d_3"], None, init=init, init_scale=0.1))     # models/dcgan.py:50


# ----------------------------------------------------------------------------- DenseNet
def densenet_discriminator(x, P, nonlinearity="crelu", L=16, scope="discriminator", init=False):
    """models/densenet.py:7-45"""
    k = [0]

    def nxt():
        p = P[f"{scope}/conv2d_{k[0]}"]
        k[0] += 1
        return p

    x = conv2d(x, nxt(), None, init=init)
    for _ in range(3):
        xs = [x]
        for _r in range(L):
            xs.append(conv2d(xs, nxt(), nonlinearity, init=init))
        x = conv2d(xs, nxt(), nonlinearity, 2, init=init)
    return feature_head(x)


def densenet_generator(us, P, nonlinearity="crelu", L=16, Fg=16, scope="generator", init=False):
    """models/densenet.py:51-88; us = [u0 [B,100], u1 [B,8,8,F], u2 [B,16,16,F], u3 [B,32,32,F]]"""
    B = us[0].shape[0]
    k = [0]

    def nxt():
        p = P[f"{scope}/conv2d_{k[0]}"]
        k[0] += 1
        return p

    x = dense(us[0], P[f"{scope}/dense_0"], None, init=init).reshape(B, 8, 8, Fg)
    xs = [x, us[1]]
    for blk in range(3):
        for _r in range(L):
            xs.append(conv2d(xs, nxt(), nonlinearity, init=init))
        if blk < 2:
            x = conv2d(xs, nxt(), nonlinearity, 1, True, init=init)
            xs = [x, us[blk + 2]]
    return torch.tanh(conv2d(xs, nxt(), nonlinearity, init=init, init_scale=0.1))     # models/densenet.py:86


# ----------------------------------------------------------------------------- optimisers
def adam_update(p, g, state, lr, mom1=0.9, mom2=0.999):
    """nn.py:50-73 (epsilon inside the sqrt; t starts at 1 and is shared per optimiser)"""
    import numpy as np
    t = np.float32(state["t"])
    # `1. - tf.pow(mom, t)` lives in the fp32 graph; `(1. - mom)` is a Python double constant
    c1 = float(np.float32(1) - np.float32(mom1) ** t)
    c2 = float(np.float32(1) - np.float32(mom2) ** t)
    if mom1 > 0:
        state["v"] = mom1 * state["v"] + (1 - mom1) * g
        v_hat = state["v"] / c1
    else:
        v_hat = g
    state["mg"] = mom2 * state["mg"] + (1 - mom2) * g * g
    mg_hat = state["mg"] / c2
    return p - lr * v_hat / torch.sqrt(mg_hat + 1e-8)


def adamax_update(p, g, state, lr, mom1=0.9, mom2=0.999):
    """nn.py:29-48"""
    if mom1 > 0:
        state["v"] = mom1 * state["v"] + (1 - mom1) * g
        v_t = state["v"]
    else:
        v_t = g
    state["mg"] = torch.maximum(mom2 * state["mg"] + 1e-8, g.abs())
    return p - lr * v_t / state["mg"]


def nesterov_update(p, g, state, lr, mom1=0.9):
    """nn.py:75-87"""
    v_new = mom1 * state["v"] - lr * g
    p_new = p - mom1 * state["v"] + (1 + mom1) * v_new
    state["v"] = v_new
    return p_new
