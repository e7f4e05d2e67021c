#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Build-container only (needs /root/reference, which does not exist on the GPU box).
The reference files `utils/matching.py` and `toy_example/matching_cpu.py` are imported
*unmodified, by path* over the NumPy stand-in `oracle/tf_standin.py` (TensorFlow 1.x is
not installable here), run on seeded inputs, and inputs + outputs
are stored as small .npz fixtures.  Only data is stored -- no reference source text.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

The fixtures pin (a) oracle/matching_np.py, (b) oracle/sinkhorn_c.c and (c) the HIP path.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import tf_standin  # noqa: E402

REF = os.environ.get("OTGAN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def _ref_modules():
    m = tf_standin.import_reference(os.path.join(REF, "utils", "matching.py"), "ref_matching")
    t = tf_standin.import_reference(os.path.join(REF, "toy_example", "matching_cpu.py"),
                                    "ref_matching_cpu")
    return m, t


def _shards(x, dt):
    return [tf_standin.T(s, dt) for s in x]


def _stack(lst):
    return np.stack([np.asarray(x) for x in lst], 0)


def run_lists(ref, fa, fb, lam, iters, dt):
    """fa, fb: [S,B,D] float64 arrays. Returns dict of reference outputs in dtype dt."""
    tf_standin.set_dtype(dt)
    out = {}
    A, Bb = _shards(fa, dt), _shards(fb, dt)
    lam_t = dt(lam)
    for tag, fn in (("two", ref.get_matched_features),
                    ("single", ref.get_matched_features_single_batch)):
        m = fn(A, Bb, lam_t, iters)
        d = ref.calc_distance(A, Bb, m)
        for k, v in zip(("aa", "bb", "ab", "ba"), m[:4]):
            out[f"{tag}_{k}"] = _stack(v)
        out[f"{tag}_entropy"] = np.asarray(m[4])
        out[f"{tag}_distance"] = np.asarray(d)
    m = ref.get_matched_features_random(A, Bb)
    out["random_distance"] = np.asarray(ref.calc_distance(A, Bb, m))
    return out


def run_toy(toy, a, b, lam, iters, dt):
    tf_standin.set_dtype(dt)
    A, Bb = tf_standin.T(a, dt), tf_standin.T(b, dt)
    m = toy.get_matched_features(A, Bb, dt(lam), iters)
    d = toy.calc_distance(A, Bb, m)
    out = {f"toy_{k}": np.asarray(v) for k, v in zip(("aa", "bb", "ab", "ba"), m[:4])}
    out["toy_entropy"] = np.asarray(m[4])
    out["toy_distance"] = np.asarray(d)
    return out


def _norm(x):
    return x / np.sqrt(np.sum(x * x, axis=-1, keepdims=True))


def clustered(rng, S, B, D, K=8, sigma=0.1):
    """`a` and `b` around different centre sets (non-degenerate distance)."""
    ca, cb = rng.randn(K, D), rng.randn(K, D)
    ia = rng.randint(0, K, size=(S, B))
    ib = rng.randint(0, K, size=(S, B))
    fa = _norm(np.abs(ca[ia] + sigma * rng.randn(S, B, D)))
    fb = _norm(np.abs(cb[ib] + sigma * rng.randn(S, B, D)))
    return fa, fb


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    ref, toy = _ref_modules()

    # ---- case "survey": the sanity case of SURVEY.md section 8c (values recorded there)
    rng = np.random.RandomState(0)
    fa = np.stack([_norm(np.abs(rng.randn(8, 32))) for _ in range(4)], 0)
    fb = np.stack([_norm(np.abs(rng.randn(8, 32))) for _ in range(4)], 0)
    ta, tb = rng.randn(64, 16), rng.randn(64, 16)
    o64 = run_lists(ref, fa, fb, 500.0, 100, np.float64)
    o32 = run_lists(ref, fa, fb, 500.0, 100, np.float32)
    t64 = run_toy(toy, ta, tb, 50.0, 50, np.float64)
    t32 = run_toy(toy, ta, tb, 50.0, 50, np.float32)
    print("survey two:", float(o64["two_entropy"]), float(o64["two_distance"]))
    print("survey single:", float(o64["single_entropy"]), float(o64["single_distance"]))
    print("survey random:", float(o64["random_distance"]))
    print("survey toy:", float(t64["toy_entropy"]), float(t64["toy_distance"]))
    save("survey", fa=fa, fb=fb, lam=500.0, iters=100, toy_a=ta, toy_b=tb, toy_lam=50.0,
         toy_iters=50, **o64, **t64,
         **{k + "_f32": v for k, v in o32.items() if v.ndim == 0},
         **{k + "_f32": v for k, v in t32.items() if v.ndim == 0})

    # ---- clustered cases (a/b from different distributions)
    for name, seed, S, B, D, lam, iters in (
            ("clustered_s2_b16_d64", 2, 2, 16, 64, 500.0, 37),
            ("clustered_s4_b8_d48", 3, 4, 8, 48, 100.0, 10),
            ("clustered_s2_b32_d128", 4, 2, 32, 128, 500.0, 100),
            ("clustered_s2_b40_d72", 5, 2, 40, 72, 250.0, 20),   # ragged: N, D not tile multiples
    ):
        rng = np.random.RandomState(seed)
        fa, fb = clustered(rng, S, B, D)
        o64 = run_lists(ref, fa, fb, lam, iters, np.float64)
        o32 = run_lists(ref, fa, fb, lam, iters, np.float32)
        save(name, fa=fa, fb=fb, lam=lam, iters=iters, **o64,
             **{k + "_f32": v for k, v in o32.items() if v.ndim == 0})

    # ---- toy config 1 (BASELINE.json configs[0]): 2D Gaussians, batch 64, 50 iterations
    rng = np.random.RandomState(7)
    ta = rng.randn(64, 2) * 0.5 + np.array([1.0, 0.0])
    tb = rng.randn(64, 2) * 0.5 + np.array([-1.0, 0.5])
    t64 = run_toy(toy, ta, tb, 50.0, 50, np.float64)
    t32 = run_toy(toy, ta, tb, 50.0, 50, np.float32)
    save("toy_gauss2d_b64", toy_a=ta, toy_b=tb, toy_lam=50.0, toy_iters=50, **t64,
         **{k + "_f32": v for k, v in t32.items() if v.ndim == 0})


if __name__ == "__main__":
    main()
