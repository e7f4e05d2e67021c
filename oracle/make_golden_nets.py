#!/usr/bin/env python3
"""Generate the golden vectors of the LAYER / MODEL / OPTIMISER code from the REFERENCE's own
Python (build container only; needs /root/reference).

The reference files `utils/nn.py`, `models/dcgan.py`, `models/densenet.py` are imported
*unmodified, by path* over the NumPy stand-in `oracle/tf_standin_nets.py` and executed on
seeded inputs.  Stored: inputs, outputs and the variable-name list -- no reference source.
Weights are NOT stored (the DCGAN critic alone has 40 M of them): every variable follows the
name-seeded recipe of `tf_standin_nets.seeded_normal` / `perturbed_gain_bias` below, which
`tests/golden_nets.py` re-implements to rebuild the identical tensors.

    python oracle/make_golden_nets.py       # rewrites tests/golden/nets_*.npz
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import tf_standin as base            # noqa: E402
from oracle import tf_standin_nets as S          # noqa: E402

REF = os.environ.get("OTGAN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def perturbed_gain_bias(name, shape):
    """Non-trivial g and b (the effective init g=1, b=0 would hide gain/bias mistakes)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    if name.endswith("/g"):
        return (0.5 + rs.random_sample(tuple(shape))).astype(np.float32)
    return (0.1 * rs.standard_normal(tuple(shape))).astype(np.float32)


def perturb():
    for name, v in S.VARS.items():
        if name.endswith("/g") or name.endswith("/b"):
            v[...] = perturbed_gain_bias(name, v.shape)


def fp32_uniform(rs, shape):
    return (rs.random_sample(shape) * 2 - 1).astype(np.float32).astype(np.float64)


def run_model(mod, tag, B, opts):
    base.set_dtype(np.float64)
    S.reset(seed=zlib.crc32(tag.encode()) & 0xffff)
    rs = np.random.RandomState(zlib.crc32(("x" + tag).encode()) & 0xffff)
    x = base.T(fp32_uniform(rs, (B, 32, 32, 3)))
    mod.discriminator(x, init=True, **opts)          # creates the variables (train.py:78)
    mod.generator(B, init=True, **opts)
    perturb()
    n0 = len(S.DRAWS)
    f = np.asarray(mod.discriminator(x, **opts))
    img = np.asarray(mod.generator(B, **opts))
    noise = S.DRAWS[n0:]
    f_img = np.asarray(mod.discriminator(base.T(img), **opts))   # critic on generated images
    out = {"x": np.asarray(x), "features": f, "image": img, "features_of_image": f_img,
           "var_names": np.array(sorted(S.VARS.keys())),
           "var_shapes": np.array([",".join(map(str, S.VARS[k].shape)) for k in sorted(S.VARS.keys())])}
    for i, u in enumerate(noise):
        out[f"noise{i}"] = u
    return out


def run_critic64(mod):
    """The reference's DCGAN critic is size-agnostic (models/dcgan.py:7-22): run it, unmodified, on one 64x64
    image (BASELINE config 5 shape; D = 8*8*2048 = 131072).  Stored in fp32 to keep the fixture small."""
    base.set_dtype(np.float64)
    S.reset(seed=zlib.crc32(b"dcgan64") & 0xffff)
    rs = np.random.RandomState(zlib.crc32(b"xdcgan64") & 0xffff)
    x = base.T(fp32_uniform(rs, (1, 64, 64, 3)))
    mod.discriminator(x, init=True)
    perturb()
    f = np.asarray(mod.discriminator(x))
    names = sorted(k for k in S.VARS if k.startswith("discriminator/"))
    return {"x": np.asarray(x, np.float32), "features": f.astype(np.float32),
            "var_names": np.array(names),
            "var_shapes": np.array([",".join(map(str, S.VARS[k].shape)) for k in names])}


def run_data_init(mod, tag, B, opts):
    """The reference's INTENDED data-dependent initialisation (utils/nn.py:133-162; built by train.py:52-54 but never
    fetched there): the init=True pass with its g / b assigns executed (eagerly, layer by layer, so that
    every layer sees the normalised output of the previous one).  Stored: input, noise, the g and b of every layer
    after the pass and the pass's outputs (fp32).  V follows the name-seeded recipe."""
    base.set_dtype(np.float64)
    S.reset(seed=zlib.crc32(("init" + tag).encode()) & 0xffff)
    rs = np.random.RandomState(zlib.crc32(("xinit" + tag).encode()) & 0xffff)
    x = base.T(fp32_uniform(rs, (B, 32, 32, 3)))
    S.EAGER_ASSIGN[0] = True
    n0 = len(S.DRAWS)
    f = np.asarray(mod.discriminator(x, init=True, **opts))
    img = np.asarray(mod.generator(B, init=True, **opts))
    S.EAGER_ASSIGN[0] = False
    noise = S.DRAWS[n0:]
    names = sorted(S.VARS.keys())
    out = {"x": np.asarray(x, np.float32), "features_init": f.astype(np.float32), "image_init": img.astype(np.float32),
           "var_names": np.array(names),
           "var_shapes": np.array([",".join(map(str, S.VARS[k].shape)) for k in names])}
    for k in names:
        if k.endswith("/g") or k.endswith("/b"):
            out["val:" + k] = np.asarray(S.VARS[k], np.float64)
    for i, u in enumerate(noise):
        out[f"noise{i}"] = u
    # after the pass a plain forward reproduces the init-pass output (same g, b)
    f2 = np.asarray(mod.discriminator(x, **opts))
    assert np.allclose(f2, f, rtol=1e-10, atol=1e-12)
    return out


def run_optimisers(nn):
    """Three applications of each reference update rule in float32 on fixed gradients."""
    base.set_dtype(np.float32)
    out = {}
    rs = np.random.RandomState(7)
    shapes = [(5, 3), (7,), (2, 2, 3, 4)]
    p0 = [rs.standard_normal(s).astype(np.float32) for s in shapes]
    grads = [[(rs.standard_normal(s) * (0.1 + 0.05 * k)).astype(np.float32) for s in shapes] for k in range(3)]
    for i, a in enumerate(p0):
        out[f"p0_{i}"] = a
    for k in range(3):
        for i, a in enumerate(grads[k]):
            out[f"grad{k}_{i}"] = a
    cases = {"adam_m05": (nn.adam_updates, dict(lr=3e-4, mom1=0.5, mom2=0.999)),
             "adam_m0": (nn.adam_updates, dict(lr=-3e-4, mom1=0.0, mom2=0.999)),
             "adamax": (nn.adamax_updates, dict(lr=2e-3, mom1=0.5, mom2=0.999)),
             "nesterov": (nn.nesterov_updates, dict(lr=1e-2, mom1=0.9))}
    for tag, (fn, kw) in cases.items():
        S.reset()
        S.EAGER_ASSIGN[0] = True
        params = [S._new_variable(a.copy(), f"p{i}:0") for i, a in enumerate(p0)]
        for k in range(3):
            if k == 1:
                S.replay_variables()
            elif k > 1:
                S._REPLAY[0] = 0
            fn(params, [base.T(g) for g in grads[k]], **kw)
            for i, p in enumerate(params):
                out[f"{tag}_step{k + 1}_{i}"] = np.array(p, dtype=np.float32)
        out[f"{tag}_kw"] = np.array(repr(sorted(kw.items())))
    S.reset()
    return out


def main():
    nn, dcgan, densenet = S.import_reference_nets(REF)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "nets_optimisers.npz"), **run_optimisers(nn))
    np.savez_compressed(os.path.join(OUT, "nets_dcgan.npz"), **run_model(dcgan, "dcgan", 2, {}))
    np.savez_compressed(os.path.join(OUT, "nets_densenet.npz"), **run_model(densenet, "densenet", 2, {}))
    np.savez_compressed(os.path.join(OUT, "nets_dcgan_critic64.npz"), **run_critic64(dcgan))
    np.savez_compressed(os.path.join(OUT, "nets_dcgan_datainit.npz"), **run_data_init(dcgan, "dcgan", 4, {}))
    np.savez_compressed(os.path.join(OUT, "nets_densenet_datainit.npz"), **run_data_init(densenet, "densenet", 4, {}))
    np.savez_compressed(os.path.join(OUT, "nets_densenet_small_celu.npz"),
                        **run_model(densenet, "densenet_small", 3,
                                    dict(layers_per_block=3, filters_per_layer=8, nonlinearity="celu")))
    for f in sorted(os.listdir(OUT)):
        if f.startswith("nets_"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
