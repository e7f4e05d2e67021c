#!/usr/bin/env python3
"""Timing of the matching block at several problem sizes (dev tool, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otgan_amd import _lib
from otgan_amd.utils import matching
dev = torch.device("cuda:0")
_lib.lib()
for (S, B, D, L) in [(2, 128, 32768, 100), (2, 256, 32768, 100), (2, 512, 32768, 100), (2, 1024, 32768, 100),
                     (2, 1024, 7296, 200), (2, 128, 7296, 200)]:
    fa = [torch.nn.functional.normalize(torch.rand(B, D, device=dev) + 0.2 * i, dim=1) for i in range(S)]
    fb = [torch.nn.functional.normalize(torch.rand(B, D, device=dev) ** 2, dim=1) for i in range(S)]
    for rows in (None, B // 4):
        for _ in range(2):
            if rows is None: matching.get_matched_features(fa, fb, 500.0, L)
            else: matching.get_matched_features_rows(fa, fb, 500.0, L, 0, rows)
        torch.cuda.synchronize()
        _lib.prof_reset(); _lib.prof_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 3
        for _ in range(reps):
            if rows is None: matching.get_matched_features(fa, fb, 500.0, L)
            else: matching.get_matched_features_rows(fa, fb, 500.0, L, 0, rows)
        e1.record(); torch.cuda.synchronize()
        p = _lib.prof_collect(); _lib.prof_enable(False)
        tot = e0.elapsed_time(e1) / reps
        print(f"N={B*S//2:5d} D={D:6d} L={L} rows={'all' if rows is None else rows:>4}: total {tot:8.3f} ms | cost {p['cost_gemm']['ms']/reps:7.3f} "
              f"({p['cost_gemm']['flop']/max(p['cost_gemm']['ms'],1e-9)/1e9:5.1f} TF/s) sinkhorn {p['sinkhorn']['ms']/reps:7.3f} "
              f"apply {p['plan_apply']['ms']/reps:7.3f} ({p['plan_apply']['flop']/max(p['plan_apply']['ms'],1e-9)/1e9:5.1f} TF/s)", flush=True)
