"""The HIP models and optimisers against golden vectors produced by running the REFERENCE's own
models/dcgan.py, models/densenet.py and utils/nn.py (unmodified, over the NumPy stand-in; see
oracle/make_golden_nets.py).  Weights are rebuilt from the fixtures' name-seeded recipe
(tests/golden_nets.py).  Tolerances: the reference run is fp64, the HIP path fp32."""
import numpy as np
import pytest
import torch

from tests import golden_nets as GN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu().numpy()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _load_into(template, fix):
    from otgan_amd import ops
    want = GN.variables(fix)
    with torch.no_grad():
        for name, v in template.named_variables().items():
            assert tuple(v.shape) == want[name].shape, name
            v.copy_(torch.from_numpy(want[name]).to(v.device))
    ops.bump_weights_epoch()
    return set(template.named_variables())


CASES = [("nets_dcgan.npz", "dcgan", {}),
         ("nets_densenet.npz", "densenet", {}),
         ("nets_densenet_small_celu.npz", "densenet", dict(layers_per_block=3, filters_per_layer=8, nonlinearity="celu"))]


@pytest.mark.parametrize("case", CASES, ids=[c[0][5:-4] for c in CASES])
def test_models_reproduce_reference_run(dev, case):
    import importlib
    fname, kind, opts = case
    fix = GN.load(fname)
    mod = importlib.import_module(f"otgan_amd.models.{kind}")
    mod.discriminator.reset(seed=1)
    mod.generator.reset(seed=1)
    x = torch.from_numpy(fix["x"]).float().to(dev)
    B = x.shape[0]
    mod.discriminator(x, init=True, **opts)                          # creates the variables
    mod.generator(batch_size=B, init=True, device=dev, **opts)
    names = _load_into(mod.discriminator, fix) | _load_into(mod.generator, fix)
    assert names == set(str(n) for n in fix["var_names"])           # same variable inventory
    us = [torch.from_numpy(u).float().to(dev) for u in GN.noise(fix)]
    noise = us[0] if kind == "dcgan" else us
    f = mod.discriminator(x, **opts)
    img = mod.generator(batch_size=B, noise=noise, device=dev, **opts)
    f_img = mod.discriminator(img, **opts)
    assert _rel(f, fix["features"]) < 2e-5
    assert _rel(img, fix["image"]) < 2e-5
    assert _rel(f_img, fix["features_of_image"]) < 5e-5


OPT = {"adam_m05": ("adam_updates", dict(lr=3e-4, mom1=0.5, mom2=0.999)),
       "adam_m0": ("adam_updates", dict(lr=-3e-4, mom1=0.0, mom2=0.999)),
       "adamax": ("adamax_updates", dict(lr=2e-3, mom1=0.5, mom2=0.999)),
       "nesterov": ("nesterov_updates", dict(lr=1e-2, mom1=0.9))}


@pytest.mark.parametrize("tag", sorted(OPT))
def test_optimisers_reproduce_reference_run(dev, tag):
    from otgan_amd.utils import nn
    fix = GN.load("nets_optimisers.npz")
    fname, kw = OPT[tag]
    ps = [torch.from_numpy(fix[f"p0_{i}"].copy()).to(dev) for i in range(3)]
    upd = getattr(nn, fname)(ps, **kw)
    for k in range(3):
        upd([torch.from_numpy(fix[f"grad{k}_{i}"]).to(dev) for i in range(3)])
        for i in range(3):
            ref = fix[f"{tag}_step{k + 1}_{i}"]
            got = ps[i].cpu().numpy()
            assert np.max(np.abs(got - ref)) <= 3e-7 * max(1.0, float(np.max(np.abs(ref)))), (tag, k, i)
