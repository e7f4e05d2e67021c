#!/usr/bin/env python3
"""Does the row pitch of the features matter to cost128_h2_kernel?  (L2 channel camping at a power-of-two pitch: dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from otgan_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
n = m = 128
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for pad in (0, 16, 32, 64, 128, 256, 1024):
    ld = D + pad
    X = torch.nn.functional.normalize(torch.rand(n, ld, device=dev), dim=1)
    Y = torch.nn.functional.normalize(torch.rand(m, ld, device=dev), dim=1)
    K = torch.empty(n, m, device=dev)
    need = L.otgan_cost_matrix_workspace_bytes(n, m, D)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    def call():
        _lib.check(L.otgan_cost_matrix_f32(X.data_ptr(), Y.data_ptr(), n, m, D, ld, 500.0, 0, 0.0, K.data_ptr(), ws.data_ptr(), need,
                                           _lib.stream_ptr()), "cost")
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    print(f"D={D} row pitch {ld:6d} floats (+{pad:4d}): {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per call (one problem: cost + finish)", flush=True)
