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


@pytest.mark.parametrize("fname,kind", [("nets_dcgan_datainit.npz", "dcgan"), ("nets_densenet_datainit.npz", "densenet")])
def test_data_dependent_init_reproduces_reference_run(dev, fname, kind):
    """--data_dependent_init: the reference's intended initialisation pass (utils/nn.py:133-162; built by
    train.py:52-54 but never fetched there) against a fixture produced by EXECUTING that branch of the reference's
    own get_params layer by layer (oracle/make_golden_nets.py::run_data_init): g and b of every layer after the pass
    and the pass's outputs.  Statistics come from the HIP forward of each layer."""
    import importlib
    from otgan_amd import ops
    from otgan_amd.utils import nn
    fix = GN.load(fname)
    mod = importlib.import_module(f"otgan_amd.models.{kind}")
    mod.discriminator.reset(seed=1)
    mod.generator.reset(seed=1)
    x = torch.from_numpy(fix["x"]).float().to(dev)
    B = x.shape[0]
    us = [torch.from_numpy(u).float().to(dev) for u in GN.noise(fix)]
    noise = us[0] if kind == "dcgan" else us
    with torch.no_grad():
        mod.discriminator(x, init=True)                                # creates the variables (g = 1, b = 0)
        mod.generator(batch_size=B, init=True, noise=noise, device=dev)
        for t in (mod.discriminator, mod.generator):                   # V from the fixture's recipe
            for name, v in t.named_variables().items():
                if name.endswith("/V"):
                    v.copy_(torch.from_numpy(GN.variable(name, v.shape)).to(dev))
                elif name.endswith("/g"):
                    v.fill_(1.0)
                else:
                    v.zero_()
        ops.bump_weights_epoch()
        nn.data_dependent_init(True)
        try:
            f = mod.discriminator(x, init=True)
            img = mod.generator(batch_size=B, init=True, noise=noise, device=dev)
        finally:
            nn.data_dependent_init(False)
        # DenseNet: 52 layers each re-normalised by statistics of a 2-image batch -- rounding differences of the
        # fp32 forward compound through the chain of 1 / std factors (DCGAN: 4 - 5 layers)
        tol_f, tol_img, tol_p = (5e-5, 5e-5, 2e-4) if kind == "dcgan" else (2e-4, 5e-4, 1e-3)
        assert _rel(f, fix["features_init"].astype(np.float64)) < tol_f
        assert _rel(img, fix["image_init"].astype(np.float64)) < tol_img
        named = {}
        named.update(mod.discriminator.named_variables())
        named.update(mod.generator.named_variables())
        # g relative; b = -mean / std is in units of the layer's (unit) output scale and is often ~0: absolute
        worst_g = max((_rel(named[k[4:]], fix[k]), k) for k in fix if k.startswith("val:") and k.endswith("/g"))
        worst_b = max((float(np.abs(named[k[4:]].double().cpu().numpy() - fix[k]).max()), k)
                      for k in fix if k.startswith("val:") and k.endswith("/b"))
        assert worst_g[0] < tol_p, worst_g
        assert worst_b[0] < tol_p, worst_b
        # a plain forward after the pass reproduces the pass (the in-place dense blocks see the same parameters)
        f2 = mod.discriminator(x)
        assert _rel(f2, fix["features_init"].astype(np.float64)) < tol_f


def test_trainer_data_dependent_init_flag(dev):
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model="dcgan", batch_size=4, nr_gpu=2, nr_sinkhorn_iter=5, nr_gen_per_disc=1, data_dependent_init=True)
    with pytest.raises(ValueError):
        OTGAN(args, dev)                                               # needs an initial batch
    x = torch.rand(8, 32, 32, 3, device=dev) * 2 - 1
    m = OTGAN(args, dev, init_batch=x)
    g1 = m.discriminator.named_variables()["discriminator/conv2d_1/g"]
    assert not torch.allclose(g1, torch.ones_like(g1))                 # initialised from the batch statistics
    r = m.step(x)
    assert torch.isfinite(r["distance"]).item()
    m2 = OTGAN(default_args(model="dcgan", batch_size=4, nr_gpu=2), dev)   # default: effective reference behaviour
    g1 = m2.discriminator.named_variables()["discriminator/conv2d_1/g"]
    assert torch.equal(g1, torch.ones_like(g1))
