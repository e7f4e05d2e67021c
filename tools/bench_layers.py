#!/usr/bin/env python3
"""Per-layer timing of the conv kernels on the DCGAN shapes (dev tool, GPU box).
usage: python tools/bench_layers.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otgan_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
LAYERS = [  # name, H, C, Cout, k, stride, up, pre
    ("D.conv0", 32, 3, 128, 5, 1, False, None),
    ("D.conv1", 32, 128, 256, 5, 2, False, "crelu"),
    ("D.conv2", 16, 256, 512, 5, 2, False, "crelu"),
    ("D.conv3", 8, 512, 1024, 5, 2, False, "crelu"),
    ("G.conv0", 4, 1024, 1024, 5, 1, True, None),
    ("G.conv1", 8, 512, 512, 5, 1, True, None),
    ("G.conv2", 16, 256, 256, 5, 1, True, None),
    ("G.conv3", 32, 128, 3, 5, 1, False, None),
]
_lib.lib()
print(f"B={B}")
for name, H, C, Cout, k, s, up, pre in LAYERS:
    mult = 2 if pre == "crelu" else 1
    x = (torch.randn(B, H, H, C, device=dev)).requires_grad_(True)
    V = (torch.randn(k, k, C * mult, Cout, device=dev) * 0.05).requires_grad_(True)
    g = torch.ones(Cout, device=dev, requires_grad=True)
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    y = ops.conv2d_op(x, V, g, b, stride=s, upsample=up, preact=ops.ACT[pre])
    dy = torch.randn_like(y)
    torch.autograd.grad(y, [x, V, g, b], dy, retain_graph=True)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    reps = 3
    for _ in range(reps):
        y = ops.conv2d_op(x, V, g, b, stride=s, upsample=up, preact=ops.ACT[pre])
        torch.autograd.grad(y, [x, V, g, b], dy)
    torch.cuda.synchronize()
    p = _lib.prof_collect(); _lib.prof_enable(False)
    row = []
    for cls in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        d = p[cls]
        row.append(f"{cls[5:]}: {d['ms']/reps:8.3f} ms {d['flop']/d['ms']/1e9:6.1f} TF/s")
    print(f"{name:8s} " + " | ".join(row), flush=True)
