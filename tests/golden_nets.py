"""Rebuild the variables of the nets fixtures (tests/golden/nets_*.npz) from their recipe.

The fixtures were produced by running the reference's own models/ and utils/nn.py over a NumPy
stand-in (oracle/make_golden_nets.py); weights are not stored but derived from the variable
name:  V = fp32(0.05 * RandomState(crc32(name)).standard_normal(shape)),
       g = fp32(0.5 + RandomState(crc32(name)).random_sample(shape)),
       b = fp32(0.1 * RandomState(crc32(name)).standard_normal(shape)).
"""
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def variable(name, shape):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    shape = tuple(int(s) for s in shape)
    if name.endswith("/V"):
        return (0.05 * rs.standard_normal(shape)).astype(np.float32)
    if name.endswith("/g"):
        return (0.5 + rs.random_sample(shape)).astype(np.float32)
    if name.endswith("/b"):
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    raise ValueError(name)


def variables(fix):
    """{full variable name: fp32 ndarray} for every variable the reference created."""
    out = {}
    for n, s in zip(fix["var_names"], fix["var_shapes"]):
        out[str(n)] = variable(str(n), [int(v) for v in str(s).split(",")])
    return out


def layer_params(fix, convert):
    """{'scope/layer': {'V','g','b'}} in the layout of oracle/nets_torch.py."""
    P = {}
    for n, a in variables(fix).items():
        layer, leaf = n.rsplit("/", 1)
        P.setdefault(layer, {})[leaf] = convert(a)
    return P


def noise(fix):
    k, out = 0, []
    while f"noise{k}" in fix:
        out.append(fix[f"noise{k}"])
        k += 1
    return out
