"""The nets oracle (oracle/nets_torch.py) against golden vectors produced by the REFERENCE's own
utils/nn.py, models/dcgan.py and models/densenet.py (run unmodified over the NumPy stand-in,
oracle/make_golden_nets.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nets_torch as NT
from tests import golden_nets as GN


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _t64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


CASES = [("nets_dcgan.npz", "dcgan", {}),
         ("nets_densenet.npz", "densenet", dict(L=16, F=16, nonlinearity="crelu")),
         ("nets_densenet_small_celu.npz", "densenet", dict(L=3, F=8, nonlinearity="celu"))]


@pytest.mark.parametrize("case", CASES, ids=[c[0][5:-4] for c in CASES])
def test_models_match_reference_run(case):
    fname, kind, o = case
    fix = GN.load(fname)
    P = GN.layer_params(fix, _t64)
    x = _t64(fix["x"])
    us = [_t64(u) for u in GN.noise(fix)]
    if kind == "dcgan":
        f = NT.dcgan_discriminator(x, P)
        img = NT.dcgan_generator(us[0], P)
        f_img = NT.dcgan_discriminator(img, P)
    else:
        f = NT.densenet_discriminator(x, P, o["nonlinearity"], o["L"])
        img = NT.densenet_generator(us, P, o["nonlinearity"], o["L"], o["F"])
        f_img = NT.densenet_discriminator(img, P, o["nonlinearity"], o["L"])
    assert f.shape == fix["features"].shape and img.shape == fix["image"].shape
    assert _rel(f, fix["features"]) < 1e-12
    assert _rel(img, fix["image"]) < 1e-12
    assert _rel(f_img, fix["features_of_image"]) < 1e-12
    # unit feature rows (models/dcgan.py:19)
    assert np.allclose(np.linalg.norm(fix["features"], axis=1), 1.0, atol=1e-12)


def test_variable_inventory_matches_reference():
    """Names and shapes of every variable the reference creates == our shape tables."""
    for fname, shapes in (("nets_dcgan.npz", [("discriminator", NT.dcgan_disc_shapes()), ("generator", NT.dcgan_gen_shapes())]),
                          ("nets_densenet.npz", [("discriminator", NT.densenet_disc_shapes()), ("generator", NT.densenet_gen_shapes())])):
        fix = GN.load(fname)
        ref = {str(n): tuple(int(v) for v in str(s).split(",")) for n, s in zip(fix["var_names"], fix["var_shapes"])}
        ours = {}
        for scope, lst in shapes:
            for layer, shp in lst:
                ours[f"{scope}/{layer}/V"] = tuple(shp)
                ours[f"{scope}/{layer}/g"] = (shp[-1],)
                ours[f"{scope}/{layer}/b"] = (shp[-1],)
        assert ours == ref


OPT = {"adam_m05": ("adam", dict(lr=3e-4, mom1=0.5, mom2=0.999)),
       "adam_m0": ("adam", dict(lr=-3e-4, mom1=0.0, mom2=0.999)),
       "adamax": ("adamax", dict(lr=2e-3, mom1=0.5, mom2=0.999)),
       "nesterov": ("nesterov", dict(lr=1e-2, mom1=0.9))}


@pytest.mark.parametrize("tag", sorted(OPT))
def test_optimisers_match_reference_run(tag):
    fix = GN.load("nets_optimisers.npz")
    kind, kw = OPT[tag]
    assert str(fix[f"{tag}_kw"]) == repr(sorted(kw.items()))
    fn = {"adam": NT.adam_update, "adamax": NT.adamax_update, "nesterov": NT.nesterov_update}[kind]
    ps = [torch.from_numpy(fix[f"p0_{i}"].copy()) for i in range(3)]
    states = [{"v": torch.zeros_like(p), "mg": torch.zeros_like(p), "t": 1.0} for p in ps]
    for k in range(3):
        for i in range(3):
            g = torch.from_numpy(fix[f"grad{k}_{i}"])
            ps[i] = fn(ps[i], g, states[i], **kw)
            states[i]["t"] += 1.0
            ref = fix[f"{tag}_step{k + 1}_{i}"]
            assert ps[i].dtype == torch.float32
            assert np.max(np.abs(ps[i].numpy() - ref)) <= 2e-7 * max(1.0, float(np.max(np.abs(ref)))), (tag, k, i)


@pytest.mark.parametrize("fname,kind", [("nets_dcgan_datainit.npz", "dcgan"), ("nets_densenet_datainit.npz", "densenet")])
def test_data_dependent_init_matches_reference_run(fname, kind):
    """The reference's intended data-dependent initialisation (utils/nn.py:133-162, executed eagerly over the stand-in
    by oracle/make_golden_nets.py::run_data_init): g and b of every layer after the pass, and the pass's outputs."""
    fix = GN.load(fname)
    P = {}
    for n, sh in zip(fix["var_names"], fix["var_shapes"]):
        n = str(n)
        shape = [int(v) for v in str(sh).split(",")]
        layer, leaf = n.rsplit("/", 1)
        # before the pass: V from the recipe, g = 1, b = 0 (the reference's initialisers, nn.py:124,143,160)
        t = _t64(GN.variable(n, shape)) if leaf == "V" else (torch.ones(shape, dtype=torch.float64) if leaf == "g"
                                                              else torch.zeros(shape, dtype=torch.float64))
        P.setdefault(layer, {})[leaf] = t
    x = _t64(fix["x"])
    us = [_t64(u) for u in GN.noise(fix)]
    if kind == "dcgan":
        f = NT.dcgan_discriminator(x, P, init=True)
        img = NT.dcgan_generator(us[0], P, init=True)
    else:
        f = NT.densenet_discriminator(x, P, init=True)
        img = NT.densenet_generator(us, P, init=True)
    assert _rel(f, fix["features_init"]) < 1e-6 and _rel(img, fix["image_init"]) < 1e-6       # stored as fp32
    for key in fix:
        if key.startswith("val:"):
            layer, leaf = key[4:].rsplit("/", 1)
            assert _rel(P[layer][leaf], fix[key]) < 1e-10, key
    # the initialised layers are normalised: unit variance / zero mean pre-activations -> g is not all ones any more
    assert not np.allclose(P["discriminator/conv2d_1"]["g"].numpy(), 1.0)
