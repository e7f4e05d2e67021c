#!/usr/bin/env python3
"""Per-block timing of the DenseNet growth layers (dev tool, GPU box).
usage: [OTGAN_DISABLE_DENSE16=1] python tools/bench_dense.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otgan_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
BLOCKS = [  # name, H, segs0 (list element widths), layers
    ("D.stage1", 32, (32,), 16), ("D.stage2", 16, (144,), 16), ("D.stage3", 8, (200,), 16),
    ("G.stage1", 8, (16, 16), 16), ("G.stage2", 16, (144, 16), 16), ("G.stage3", 32, (208, 16), 16),
]
_lib.lib()
print(f"B={B} dense16={'off' if os.environ.get('OTGAN_DISABLE_DENSE16') == '1' else 'on'}")
for name, H, segs0, L in BLOCKS:
    C0 = sum(segs0)
    x = torch.randn(B, H, H, C0, device=dev).requires_grad_(True)
    params = []
    for k in range(L):
        Ck = C0 + 16 * k
        params.append(((torch.randn(3, 3, 2 * Ck, 16, device=dev) * 0.05).requires_grad_(True),
                       torch.ones(16, device=dev, requires_grad=True), torch.zeros(16, device=dev, requires_grad=True)))
    flat = [t for p in params for t in p]
    def run():
        out = ops.dense_block_op(x, segs0, params, 3, ops.ACT["crelu"])
        return out
    y = run()
    dy = torch.randn_like(y)
    torch.autograd.grad(y, [x] + flat, dy)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    reps = 3
    for _ in range(reps):
        ops.bump_weights_epoch()
        y = run()
        torch.autograd.grad(y, [x] + flat, dy)
    torch.cuda.synchronize()
    p = _lib.prof_collect(); _lib.prof_enable(False)
    row = []
    for cls in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        d = p[cls]
        row.append(f"{cls[5:]}: {d['ms']/reps:8.3f} ms {d['flop']/max(d['ms'],1e-9)/1e9:6.1f} TF/s")
    print(f"{name:9s} " + " | ".join(row), flush=True)
