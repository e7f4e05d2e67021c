#!/usr/bin/env python3
"""One matching problem, called a few times (workload for the PMC passes of tools/pmc_matching.sh).
usage: tools/matching_workload.py N D L [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otgan_amd import _lib
from otgan_amd.utils import matching
N, D, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] not in ("", "0") else None
# mode (5th argument): "arrays" = the reference's operator (four matched arrays); "grad" = the training-mode entry the step
# calls (critic step: both gradients); "rank" = what one rank of 2N / rows runs: its three cost row slices + the
# training-mode rows entry on precomputed log-kernels
mode = sys.argv[5] if len(sys.argv) > 5 else "arrays"
dev = torch.device("cuda:0")
_lib.lib()
g = torch.Generator(device=dev).manual_seed(3)
def feats(shift):
    c = torch.rand(32, D, device=dev, generator=g) + shift
    f = (c[torch.randint(0, 32, (2 * N,), device=dev, generator=g)] + 0.1 * torch.randn(2 * N, D, device=dev, generator=g)).abs()
    return list(torch.chunk(torch.nn.functional.normalize(f, dim=1), 2, 0))
fa, fb = feats(0.0), feats(0.5)
fa_flat, fb_flat = torch.cat(fa), torch.cat(fb)
K6 = None
if mode == "rank":
    from otgan_amd import trainer as T
    W = 2 * N // rows
    own = lambda t, r: t[r * rows:(r + 1) * rows]
    allk = torch.stack([T.rank_log_kernel_slices(r, W, own(fa_flat, r), own(fb_flat, r), fa_flat, fb_flat, 500.0)
                        for r in range(W)], 0)
    K6 = T.assemble_log_kernels(allk, W)
    torch.cuda.synchronize()
for _ in range(4):
    if mode == "grad":
        out = matching.matched_feature_grads(fa_flat, fb_flat, 500.0, L, need_b=True, rows=None if rows is None else (0, rows))
    elif mode == "rank" and T.rank_stack_ok(rows, fa_flat):
        # round 5: what trainer._match runs -- ONE split of the gathered features (matching.FeatureStack) for the rank's cost
        # row slices and its plan application; the other ranks' slices are precomputed (allk)
        out = T.rank_matching_stack(0, W, rows, fa_flat, fb_flat, 500.0, L, True, gather=allk)
    elif mode == "rank":
        T.rank_log_kernel_slices(0, W, own(fa_flat, 0), own(fb_flat, 0), fa_flat, fb_flat, 500.0)
        out = matching.matched_feature_grads(fa_flat, fb_flat, 500.0, L, need_b=True, rows=(0, rows), log_kernels=K6)
    elif rows is None:
        out = matching.get_matched_features(fa, fb, 500.0, L)
    else:
        out = matching.get_matched_features_rows(fa, fb, 500.0, L, 0, rows)
torch.cuda.synchronize()
print("ok")
