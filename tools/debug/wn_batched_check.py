"""dev check: batched weight norm (one launch per dense block) against the per-layer kernels and fp64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
C0, L = 200, 16
Vs = [torch.randn(9 * 2 * (C0 + 16 * k), 16, device=dev) * 0.05 for k in range(L)]
gs = [torch.rand(16, device=dev) + 0.5 for _ in range(L)]
dws = [torch.randn_like(v) for v in Vs]
out = ops.weightnorm_fwd_block(Vs, gs)
for k in range(L):
    w, wT, inv = ops.weightnorm_fwd(Vs[k], gs[k])
    V64 = Vs[k].double()
    w64 = gs[k].double() * V64 / V64.pow(2).sum(0).clamp_min(1e-12).sqrt()
    e = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(k, "fwd", e(out[k][0], w64), e(w, w64), e(out[k][1], w64.t()), float((out[k][2] - inv).abs().max()))
parts = [[(dws[k].data_ptr(), None, dws[k].shape[0] // 9, 16)] for k in range(L)]
dVs, dgs = ops.weightnorm_bwd_block(Vs, gs, [o[2] for o in out], parts)
for k in range(L):
    dV, dg = ops.weightnorm_bwd(Vs[k], gs[k], out[k][2], dws[k])
    V64 = Vs[k].double().requires_grad_(True); g64 = gs[k].double().requires_grad_(True)
    w64 = g64 * V64 / V64.pow(2).sum(0).clamp_min(1e-12).sqrt()
    rV, rg = torch.autograd.grad(w64, [V64, g64], dws[k].double())
    e = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(k, "bwd", e(dVs[k], rV), e(dV, rV), e(dgs[k], rg), e(dg, rg))
