#!/usr/bin/env python3
"""Dev tool (GPU box): one conv2d case (N H W Cin Cout k stride up pre) vs the fp64 oracle, with error maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import nets_torch as NT
from otgan_amd import ops
dev = torch.device("cuda:0")
N, H, W, C, Cout, k, stride = map(int, sys.argv[1:8])
up = sys.argv[8] == "1"
pre = None if sys.argv[9] == "none" else sys.argv[9]
gen = torch.Generator().manual_seed(1)
mult = 2 if pre in ("crelu", "celu") else 1
x64 = torch.randn(N, H, W, C, generator=gen, dtype=torch.float64).float().double().requires_grad_(True)
V64 = (torch.randn(k, k, C * mult, Cout, generator=gen, dtype=torch.float64) * 0.05).float().double().requires_grad_(True)
g64 = (torch.rand(Cout, generator=gen, dtype=torch.float64) + 0.5).float().double().requires_grad_(True)
b64 = (torch.randn(Cout, generator=gen, dtype=torch.float64) * 0.1).float().double().requires_grad_(True)
y_ref = NT.conv2d([x64], {"V": V64, "g": g64, "b": b64}, pre, stride, up)
dy64 = torch.randn(y_ref.shape, generator=gen, dtype=torch.float64).float().double()
gref = torch.autograd.grad(y_ref, [x64, V64, g64, b64], dy64)
x, V, g, b = [t.detach().float().to(dev).requires_grad_(True) for t in (x64, V64, g64, b64)]
y = ops.conv2d_op(x, V, g, b, stride=stride, upsample=up, preact=ops.ACT[pre], segs=(C,))
rel = lambda a, r: float((a.detach().double().cpu() - r).norm() / r.norm())
print("fwd", rel(y, y_ref))
got = torch.autograd.grad(y, [x, V, g, b], dy64.float().to(dev))
for n, a, r in zip("dx dV dg db".split(), got, gref):
    print(n, rel(a, r))
d = (got[0].double().cpu() - gref[0]).abs().amax(3)
bad = (d > 1e-4 * gref[0].abs().max()).nonzero()
print("bad dx pixels:", len(bad), bad[:40].tolist())
