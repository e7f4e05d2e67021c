#!/usr/bin/env python3
"""sinkhorn_panel_kernel phase stamps at N = 1024 (needs tools/debug/bin/libotgan_panel_timing.so via OTGAN_LIB_PATH)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from otgan_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
P, n, iters, lam = 6, 1024, 100, 500.0
rng = np.random.RandomState(1)
K = torch.as_tensor((-lam * rng.rand(P, n, n) * 0.3).astype(np.float32), device=dev)
plan = torch.empty(P, n, n, device=dev); planT = torch.empty(P, n, n, device=dev)
stats = torch.empty(P, 4, dtype=torch.float64, device=dev)
need = max(L.otgan_sinkhorn_workspace_bytes(P, n, n), 256)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
for _ in range(3):
    _lib.check(L.otgan_sinkhorn_plan_f32(K.data_ptr(), P, n, n, iters, lam, plan.data_ptr(), planT.data_ptr(), stats.data_ptr(),
                                         ws.data_ptr(), need, _lib.stream_ptr()), "sinkhorn")
torch.cuda.synchronize()
