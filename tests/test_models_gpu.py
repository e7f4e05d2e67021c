"""GPU parity of whole networks: forward features / images and parameter gradients of the
HIP models vs the PyTorch restatement (oracle/nets_torch.py) run in fp64 on the CPU with the
same parameters.  Covers the reference's gradient injection (train.py:111-128): VJPs with an
arbitrary upstream `grad_ys`."""
import pytest
import torch

from oracle import nets_torch as NT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _oracle_params(template):
    """fp64 CPU copy of a template's variables in the oracle's dict layout."""
    P = {}
    for name, v in template.named_variables().items():
        layer, leaf = name.rsplit("/", 1)
        P.setdefault(layer, {})[leaf] = v.detach().double().cpu().requires_grad_(True)
    return P


def test_dcgan_shapes_and_names(dev):
    from otgan_amd.models import dcgan
    dcgan.discriminator.reset(seed=1)
    dcgan.generator.reset(seed=1)
    x = torch.rand(2, 32, 32, 3, device=dev) * 2 - 1
    f = dcgan.discriminator(x, init=True, nonlinearity="crelu")
    assert f.shape == (2, 32768)                              # train.py:55-56 prints this width
    img = dcgan.generator(batch_size=2, init=True, nonlinearity="crelu", device=dev)
    assert img.shape == (2, 32, 32, 3) and float(img.abs().max()) < 1.0
    names = list(dcgan.discriminator.named_variables())
    assert names[:3] == ["discriminator/conv2d_0/V", "discriminator/conv2d_0/g", "discriminator/conv2d_0/b"]
    nd = sum(v.numel() for v in dcgan.discriminator.trainable_variables())
    ng = sum(v.numel() for v in dcgan.generator.trainable_variables())
    assert nd == 34419840 and ng == 37761926


def test_dcgan_critic_parity(dev):
    from otgan_amd.models import dcgan
    dcgan.discriminator.reset(seed=2)
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(3, 32, 32, 3, generator=gen) * 2 - 1)
    xg = x.to(dev).requires_grad_(True)
    f = dcgan.discriminator(xg, nonlinearity="crelu")
    P = _oracle_params(dcgan.discriminator)
    x64 = x.double().requires_grad_(True)
    f_ref = NT.dcgan_discriminator(x64, P)
    assert _rel(f, f_ref) < 2e-5
    gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
    params = dcgan.discriminator.trainable_variables()
    got = torch.autograd.grad(f, [xg] + params, gy.to(dev))
    leaves = [x64] + [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in dcgan.discriminator.named_variables()]
    ref = torch.autograd.grad(f_ref, leaves, gy.double())
    names = ["dx"] + list(dcgan.discriminator.named_variables())
    for n, a, r in zip(names, got, ref):
        assert _rel(a, r) < 5e-5, n


def test_dcgan_generator_parity(dev):
    from otgan_amd.models import dcgan
    dcgan.generator.reset(seed=3)
    gen = torch.Generator().manual_seed(1)
    u = torch.rand(2, 100, generator=gen) * 2 - 1
    img = dcgan.generator(batch_size=2, nonlinearity="crelu", noise=u.to(dev))
    P = _oracle_params(dcgan.generator)
    img_ref = NT.dcgan_generator(u.double(), P)
    assert _rel(img, img_ref) < 2e-5
    gy = torch.randn(img_ref.shape, generator=gen, dtype=torch.float64).float()
    params = dcgan.generator.trainable_variables()
    got = torch.autograd.grad(img, params, gy.to(dev))
    leaves = [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in dcgan.generator.named_variables()]
    ref = torch.autograd.grad(img_ref, leaves, gy.double())
    for n, a, r in zip(dcgan.generator.named_variables(), got, ref):
        assert _rel(a, r) < 5e-5, n


def test_dcgan_ema_generator(dev):
    from otgan_amd.models import dcgan
    from otgan_amd.utils import nn
    dcgan.generator.reset(seed=4)
    u = torch.rand(2, 100, device=dev) * 2 - 1
    a = dcgan.generator(batch_size=2, noise=u)
    ema = nn.ExponentialMovingAverage(0.999)
    upd = ema.apply(dcgan.generator.trainable_variables())
    b = dcgan.generator(batch_size=2, noise=u, ema=ema)       # shadows start at the parameters
    assert torch.equal(a.detach(), b.detach())
    with torch.no_grad():
        for p in dcgan.generator.trainable_variables():
            p.add_(0.01)
    upd()
    p0 = dcgan.generator.trainable_variables()[0]
    assert _rel(ema.average(p0), p0.detach() - 0.01 + 0.001 * 0.01) < 1e-6


# ------------------------------------------------------------------------------- DenseNet
def test_densenet_shapes(dev):
    from otgan_amd.models import densenet
    densenet.discriminator.reset(seed=1)
    densenet.generator.reset(seed=1)
    x = torch.rand(2, 32, 32, 3, device=dev) * 2 - 1
    f = densenet.discriminator(x, init=True, nonlinearity="crelu")
    assert f.shape == (2, 7296)
    img = densenet.generator(batch_size=2, init=True, nonlinearity="crelu", device=dev)
    assert img.shape == (2, 32, 32, 3)
    nd = sum(v.numel() for v in densenet.discriminator.trainable_variables())
    ng = sum(v.numel() for v in densenet.generator.trainable_variables())
    assert nd == 7453016 and ng == 6012422
    assert len(densenet.discriminator.trainable_variables()) == 52 * 3


@pytest.mark.parametrize("L", [2, 16])
def test_densenet_critic_parity(dev, L):
    from otgan_amd.models import densenet
    densenet.discriminator.reset(seed=5)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(2, 32, 32, 3, generator=gen) * 2 - 1
    xg = x.to(dev).requires_grad_(True)
    f = densenet.discriminator(xg, nonlinearity="crelu", layers_per_block=L)
    P = _oracle_params(densenet.discriminator)
    x64 = x.double().requires_grad_(True)
    f_ref = NT.densenet_discriminator(x64, P, "crelu", L)
    assert _rel(f, f_ref) < 5e-5
    gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
    params = densenet.discriminator.trainable_variables()
    got = torch.autograd.grad(f, [xg] + params, gy.to(dev))
    names = list(densenet.discriminator.named_variables())
    leaves = [x64] + [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
    ref = torch.autograd.grad(f_ref, leaves, gy.double())
    # (This parameter seed has no pre-activation within fp32 rounding of zero on the default engine: every tensor sits
    # at <= 1e-5.  Other seeds -- and this one under OTGAN_WINO_PIECES=3 -- flip one CReLU unit between two fp32
    # evaluation orders and land at 1e-3 ... 7e-2 on the tensors next to it, with or without the dense-block split:
    # tools/debug/dn_critic_err.py.  The blocks themselves are pinned at 2e-5 in test_layers_gpu.py.)
    for n, a, r in zip(["dx"] + names, got, ref):
        assert _rel(a, r) < 2e-4, n


def test_densenet_generator_parity(dev):
    from otgan_amd.models import densenet
    densenet.generator.reset(seed=6)
    gen = torch.Generator().manual_seed(1)
    L = 3
    us = [torch.rand(2, 100, generator=gen) * 2 - 1, torch.rand(2, 8, 8, 16, generator=gen) * 2 - 1,
          torch.rand(2, 16, 16, 16, generator=gen) * 2 - 1, torch.rand(2, 32, 32, 16, generator=gen) * 2 - 1]
    img = densenet.generator(batch_size=2, nonlinearity="crelu", layers_per_block=L,
                             noise=[u.to(dev) for u in us])
    P = _oracle_params(densenet.generator)
    img_ref = NT.densenet_generator([u.double() for u in us], P, "crelu", L)
    assert _rel(img, img_ref) < 5e-5
    gy = torch.randn(img_ref.shape, generator=gen, dtype=torch.float64).float()
    params = densenet.generator.trainable_variables()
    got = torch.autograd.grad(img, params, gy.to(dev))
    names = list(densenet.generator.named_variables())
    leaves = [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
    ref = torch.autograd.grad(img_ref, leaves, gy.double())
    for n, a, r in zip(names, got, ref):
        assert _rel(a, r) < 2e-4, n
