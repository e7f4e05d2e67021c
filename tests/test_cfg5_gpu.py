"""GPU: BASELINE configs[4] -- 64x64 images, D = 8*8*2048 = 131072 critic features, 512 images per GPU as two
mini-batch halves of 256 (stacked Gram of 1024 rows).  The reference hard-codes 32x32 in its driver and
generator (train.py:52,67; models/dcgan.py:37-46) but its DCGAN critic is size-agnostic (models/dcgan.py:7-22), so:

  * the critic at 64x64 is pinned to a fixture produced by the reference's own critic code
    (tests/golden/nets_dcgan_critic64.npz, oracle/make_golden_nets.py);
  * critic and generator (the build's `image_size=64` option: stem 8x8 instead of 4x4) are compared, forward and
    every gradient, with the fp64 oracle nets (oracle/nets_torch.py, which is size-agnostic);
  * the matching at N = 256, D = 131072, lambda = 500, L = 100 is compared with the fp64 matching oracle."""
import numpy as np
import pytest
import torch

from oracle import matching_np as M
from oracle import nets_torch as NT
from conftest import REL_DIFF_INJECTED
from tests import golden_nets as GN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a, np.float64))
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b, np.float64))
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _oracle_params(template):
    P = {}
    for name, v in template.named_variables().items():
        layer, leaf = name.rsplit("/", 1)
        P.setdefault(layer, {})[leaf] = v.detach().double().cpu().requires_grad_(True)
    return P


def test_critic_64_reproduces_reference_run(dev):
    from otgan_amd import ops
    from otgan_amd.models import dcgan
    fix = GN.load("nets_dcgan_critic64.npz")
    dcgan.discriminator.reset(seed=1)
    x = torch.from_numpy(fix["x"]).float().to(dev)
    assert tuple(x.shape) == (1, 64, 64, 3)
    dcgan.discriminator(x, init=True)
    want = GN.variables(fix)
    with torch.no_grad():
        for name, v in dcgan.discriminator.named_variables().items():
            v.copy_(torch.from_numpy(want[name]).to(dev))
    ops.bump_weights_epoch()
    f = dcgan.discriminator(x)
    assert tuple(f.shape) == (1, 131072)
    assert _rel(f, fix["features"]) < 2e-5


def _critic_layerwise(dev, x, nonlinearity):
    """The critic run layer by layer on both sides (same layer calls as models/dcgan.py:7-22), keeping the
    pre-activation tensors.  CReLU's derivative is a step at 0: with ~1.5 M pre-activations per 64x64 image pair a
    few land within fp32 rounding (|x| ~ 1e-7) of zero, and a legal rounding difference in the forward pass picks
    the other branch of relu'(x) for that unit -- an O(1) change of one mask bit, not an arithmetic error.  The
    oracle therefore evaluates its derivative masks at the signs the HIP forward produced: units whose sign
    differs are nudged across zero in the oracle (|shift| < 2e-6, counted and bounded), so that both sides
    differentiate the same piecewise-linear function."""
    from otgan_amd.models import dcgan
    from otgan_amd.utils import nn
    acts = []

    def spec(z, **kw):
        with nn.arg_scope([nn.conv2d, nn.dense], counters={}, init=False, weight_norm=True, ema=None):
            for filters, s, act in dcgan._CRITIC:
                z = nn.conv2d(z, filters, filter_size=[5, 5], stride=[s, s], pre_activation=nonlinearity if act else None)
                acts.append(z)
            return nn.feature_head(z)

    t = nn.make_template("discriminator", spec)
    t.store = dcgan.discriminator.store               # shared variables, like make_template
    xg = x.to(dev).requires_grad_(True)
    f = t(xg)
    P = _oracle_params(dcgan.discriminator)
    x64 = x.double().requires_grad_(True)
    h, flips = x64, 0
    for i, (filters, s, act) in enumerate(dcgan._CRITIC):
        h = NT.conv2d(h, P[f"discriminator/conv2d_{i}"], nonlinearity if act else None, s)
        a = acts[i].detach().double().cpu()
        differ = (torch.sign(a) != torch.sign(h.detach())) & (a != 0)
        assert float((h.detach() - a).abs()[differ].max() if differ.any() else 0.0) < 2e-6   # only rounding-level units
        flips += int(differ.sum())
        h = h + torch.where(differ, a - h.detach(), torch.zeros_like(a))                      # same mask on both sides
    f_ref = NT.feature_head(h)
    return xg, f, x64, f_ref, P, flips


def test_dcgan_64_critic_parity_fwd_and_grads(dev):
    from otgan_amd.models import dcgan
    dcgan.discriminator.reset(seed=21)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 64, 64, 3, generator=gen) * 2 - 1
    with torch.no_grad():
        dcgan.discriminator(x.to(dev), init=True)                      # creates the variables
        f_plain = dcgan.discriminator(x.to(dev), nonlinearity="crelu")
        assert f_plain.shape == (2, 131072)
        assert _rel(f_plain, NT.dcgan_discriminator(x.double(), _oracle_params(dcgan.discriminator))) < 2e-5
    xg, f, x64, f_ref, P, flips = _critic_layerwise(dev, x, "crelu")
    assert torch.equal(f.detach(), f_plain)                             # the layer-wise run IS the model
    assert flips <= 16, flips                                           # a handful of 1.5 M units
    assert _rel(f, f_ref) < 2e-5
    gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
    params = dcgan.discriminator.trainable_variables()
    got = torch.autograd.grad(f, [xg] + params, gy.to(dev))
    names = list(dcgan.discriminator.named_variables())
    leaves = [x64] + [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
    ref = torch.autograd.grad(f_ref, leaves, gy.double())
    for n, a, r in zip(["dx"] + names, got, ref):
        assert _rel(a, r) < 5e-5, n


def test_dcgan_64_generator_parity_fwd_and_grads(dev):
    from otgan_amd.models import dcgan
    dcgan.generator.reset(seed=22)
    gen = torch.Generator().manual_seed(6)
    u = torch.rand(2, 100, generator=gen) * 2 - 1
    img = dcgan.generator(batch_size=2, nonlinearity="crelu", noise=u.to(dev), image_size=64)
    assert img.shape == (2, 64, 64, 3)
    P = _oracle_params(dcgan.generator)
    assert tuple(P["generator/dense_0"]["V"].shape) == (100, 2 * 8 * 8 * 1024)
    img_ref = NT.dcgan_generator(u.double(), P)
    assert _rel(img, img_ref) < 2e-5
    gy = torch.randn(img_ref.shape, generator=gen, dtype=torch.float64).float()
    params = dcgan.generator.trainable_variables()
    got = torch.autograd.grad(img, params, gy.to(dev))
    names = list(dcgan.generator.named_variables())
    leaves = [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
    ref = torch.autograd.grad(img_ref, leaves, gy.double())
    for n, a, r in zip(names, got, ref):
        assert _rel(a, r) < 5e-5, n


def test_matching_cfg5_size_vs_oracle(dev):
    """N = 256 rows per half (512 images per GPU), D = 131072, lambda = 500, 100 iterations."""
    from otgan_amd.utils import matching
    S, B, D, lam, iters = 2, 256, 131072, 500.0, 100
    rng = np.random.RandomState(17)
    ca, cb = rng.randn(32, D), rng.randn(32, D)
    fa = np.stack([M.clustered_features(rng, B, D, ca) for _ in range(S)]).astype(np.float32)
    fb = np.stack([M.clustered_features(rng, B, D, cb) for _ in range(S)]).astype(np.float32)
    A = [torch.as_tensor(x, device=dev) for x in fa]
    Bt = [torch.as_tensor(x, device=dev) for x in fb]
    out = matching.get_matched_features(A, Bt, lam, iters)
    ref = M.get_matched_features(list(fa), list(fb), lam, iters)
    dref = float(M.calc_distance(list(fa), list(fb), ref))
    for k, got, want in zip("aa bb ab ba".split(), out[:4], ref[:4]):
        assert _rel(torch.stack(got), np.stack(want)) < 2e-4, k
    ga = torch.stack(out[0]) - torch.stack(out[2])
    assert _rel(ga, np.stack(ref[0]) - np.stack(ref[2])) < 2 * REL_DIFF_INJECTED     # (difference of two separately rounded arrays)
    assert float(out[4]) == pytest.approx(float(ref[4]), rel=2e-4)
    for d in (float(out.distance), float(matching.calc_distance(A, Bt, out)), float(matching.closed_form_distance(out))):
        assert abs(d - dref) <= 1e-4 * abs(dref) + 1e-7, (d, dref)
    st = out.stats.cpu().numpy()
    np.testing.assert_allclose(st[:, 2], B, rtol=1e-5)        # plan rows sum to one
