#!/usr/bin/env python3
"""A rank of eight's matching call (N = 1024 rows, 256 own rows), timed per kernel class (dev tool, GPU box).
   usage: tools/exp/rank_ab.py [D] [L]      env switches are read by the library once per process"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from otgan_amd import _lib, trainer as T
dev = torch.device("cuda:0")
N, rows = 1024, 256
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator(device=dev).manual_seed(5)
def feats(n, shift):
    c = torch.rand(32, D, device=dev, generator=g) + shift
    f = (c[torch.randint(0, 32, (n,), device=dev, generator=g)] + 0.1 * torch.randn(n, D, device=dev, generator=g)).abs()
    return torch.nn.functional.normalize(f, dim=1)
fa, fb = feats(2 * N, 0.0), torch.nn.functional.normalize(feats(2 * N, 0.5) ** 2, dim=1)
W = 2 * N // rows
own = lambda t, r: t[r * rows:(r + 1) * rows]
allk = torch.stack([T.rank_log_kernel_slices(r, W, own(fa, r), own(fb, r), fa, fb, 500.0) for r in range(W)], 0)
assert T.rank_stack_ok(rows, fa)
for need_b in (False, True):
    for _ in range(3):
        out = T.rank_matching_stack(0, W, rows, fa, fb, 500.0, L, need_b, gather=allk)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = T.rank_matching_stack(0, W, rows, fa, fb, 500.0, L, need_b, gather=allk)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    p = _lib.prof_collect(); _lib.prof_enable(False)
    print(f"D={D} L={L} need_b={need_b}: call {us:7.1f} us | cost {p['cost_gemm']['ms'] / reps * 1e3:6.1f} sinkhorn {p['sinkhorn']['ms'] / reps * 1e3:6.1f} "
          f"apply {p['plan_apply']['ms'] / reps * 1e3:6.1f}", flush=True)
ga = out[0]
print("checksum", float(ga.double().abs().sum()))
