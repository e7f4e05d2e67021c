"""Worker of tests/test_matching_engine_accuracy_gpu.py: one process per matching GEMM engine (the library reads its engine
switches once).  For every case it runs the training-mode matching entry the step calls (the injected gradients
`f_aa - f_ab`, `f_bb - f_ba` of reference train.py:111,125-126) and prints the relative L2 error of both differences
against the fp64 oracle (oracle/matching_np.py, pinned to the reference-generated fixtures).  The oracle's rows are
computed by the first worker that needs them and kept in `refdir` for the next engine.

usage: matching_engine_worker.py <refdir> <case> [<case> ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import matching_np as M  # noqa: E402

# name: (rows per mini-batch half N, feature width D, Sinkhorn sweeps, row ranges checked (None = all 2N rows), seed)
CASES = {
    "N128_D32768": (128, 32768, 100, None, 3),                         # configs[1]: the headline problem (exact-fp32 engine)
    "N256_D7296": (256, 7296, 200, None, 3),                           # configs[3] width on one GPU
    "N256_D131072": (256, 131072, 100, None, 3),                       # configs[4]: 64x64 critic width
    "N1024_D32768_rank": (1024, 32768, 100, ((0, 256), (5 * 256, 256)), 11),   # configs[2]: ranks 0 and 5 of eight
    "N1024_D7296_rank": (1024, 7296, 200, ((0, 256), (5 * 256, 256)), 11),     # configs[3]: ranks 0 and 5 of eight
}
LAM = 500.0


def features(seed, rows, D):
    rng = np.random.RandomState(seed)
    ca, cb = rng.randn(32, D), rng.randn(32, D)
    fa = M.clustered_features(rng, rows, D, ca).astype(np.float32)
    fb = M.clustered_features(rng, rows, D, cb).astype(np.float32)
    return fa, fb


def reference(refdir, name, fa, fb):
    """-> {(r0, n): (ra, rb)} fp32 copies of the fp64 oracle's differences for the checked row ranges, distance"""
    N, D, iters, ranges, _ = CASES[name]
    path = os.path.join(refdir, name + ".npz")
    if os.path.exists(path):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    f64 = lambda a: a.astype(np.float64)
    fa1, fa2, fb1, fb2 = f64(fa[:N]), f64(fa[N:]), f64(fb[:N]), f64(fb[N:])
    plans, costs, _ent = M.two_batch_plans(fa1, fa2, fb1, fb2, LAM, iters)
    out = {"distance": np.float64(M.closed_form_from(plans, costs, N))}
    for (r0, n) in (ranges or ((0, N), (N, N))):
        half, q0 = divmod(r0, N)
        aa, bb, ab, ba = M.matched_rows(plans, fa1, fa2, fb1, fb2, half, q0, q0 + n)
        out[f"ra_{r0}"] = (aa - ab).astype(np.float32)
        out[f"rb_{r0}"] = (bb - ba).astype(np.float32)
    np.savez(path, **out)
    return out


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))


def main():
    refdir, names = sys.argv[1], sys.argv[2:]
    from otgan_amd.utils import matching
    dev = torch.device("cuda:0")
    res = {}
    for name in names:
        N, D, iters, ranges, seed = CASES[name]
        t0 = time.time()
        fa, fb = features(seed, 2 * N, D)
        ref = reference(refdir, name, fa, fb)
        A, B = torch.as_tensor(fa, device=dev), torch.as_tensor(fb, device=dev)
        errs_a, errs_b, dists = [], [], []
        if ranges is None:
            ga, gb, _ent, dist = matching.matched_feature_grads(A, B, LAM, iters)
            ga, gb = ga.cpu().numpy(), gb.cpu().numpy()
            for r0 in (0, N):
                errs_a.append(rel(ga[r0:r0 + N], ref[f"ra_{r0}"]))
                errs_b.append(rel(gb[r0:r0 + N], ref[f"rb_{r0}"]))
            dists.append(float(dist))
        else:
            for (r0, n) in ranges:      # the rank-level call (the library forms the log-kernels with the engine under test)
                ga, gb, _ent, dist = matching.matched_feature_grads(A, B, LAM, iters, rows=(r0, n))
                errs_a.append(rel(ga.cpu().numpy(), ref[f"ra_{r0}"]))
                errs_b.append(rel(gb.cpu().numpy(), ref[f"rb_{r0}"]))
                dists.append(float(dist))
        dref = float(ref["distance"])
        res[name] = {"grad_gen": max(errs_a), "grad_dat": max(errs_b),
                     "distance_rel": max(abs(d - dref) / abs(dref) for d in dists), "seconds": round(time.time() - t0, 1)}
        del A, B
        torch.cuda.empty_cache()
    print("MATCH_ENGINE_JSON " + json.dumps(res))


if __name__ == "__main__":
    main()
