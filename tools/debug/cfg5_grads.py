#!/usr/bin/env python3
"""Dev tool (GPU box): per-tensor errors of the 64x64 DCGAN critic / generator vs the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import nets_torch as NT
from otgan_amd.models import dcgan
dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

def oparams(t):
    P = {}
    for name, v in t.named_variables().items():
        layer, leaf = name.rsplit("/", 1)
        P.setdefault(layer, {})[leaf] = v.detach().double().cpu().requires_grad_(True)
    return P

dcgan.discriminator.reset(seed=21)
gen = torch.Generator().manual_seed(5)
x = torch.rand(B, size, size, 3, generator=gen) * 2 - 1
xg = x.to(dev).requires_grad_(True)
f = dcgan.discriminator(xg, nonlinearity="crelu")
P = oparams(dcgan.discriminator)
x64 = x.double().requires_grad_(True)
f_ref = NT.dcgan_discriminator(x64, P)
print("critic fwd", rel(f, f_ref))
gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
params = dcgan.discriminator.trainable_variables()
got = torch.autograd.grad(f, [xg] + params, gy.to(dev))
names = list(dcgan.discriminator.named_variables())
leaves = [x64] + [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
ref = torch.autograd.grad(f_ref, leaves, gy.double())
for n, a, r in zip(["dx"] + names, got, ref):
    print(f"  {n:34s} {rel(a, r):.3e}")
d = (got[0].double().cpu() - ref[0]).abs()
print("dx err map (max over channel) rows with err > 1e-4 * max:", (d.amax(3) > 1e-4 * ref[0].abs().max()).nonzero()[:20].tolist())
print("dx max abs err", float(d.max()), "ref max", float(ref[0].abs().max()))
