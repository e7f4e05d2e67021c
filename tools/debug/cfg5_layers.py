#!/usr/bin/env python3
"""Dev tool (GPU box): gradient w.r.t. every intermediate activation of the DCGAN critic vs the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import nets_torch as NT
from otgan_amd.models import dcgan
from otgan_amd.utils import nn
dev = torch.device("cuda:0")
size = int(sys.argv[1]); B = int(sys.argv[2])
rel = lambda a, r: float((a.detach().double().cpu() - r.detach()).norm() / r.detach().norm())
dcgan.discriminator.reset(seed=21)
gen = torch.Generator().manual_seed(5)
x = torch.rand(B, size, size, 3, generator=gen) * 2 - 1
acts = []
def spec(x, **kw):
    with nn.arg_scope([nn.conv2d, nn.dense], counters={}, init=False, weight_norm=True, ema=None):
        for filters, s, act in dcgan._CRITIC:
            x = nn.conv2d(x, filters, filter_size=[5, 5], stride=[s, s], pre_activation="crelu" if act else None)
            x.retain_grad(); acts.append(x)
        return nn.feature_head(x)
t = nn.make_template('discriminator', spec)
t.store = dcgan.discriminator.store
xg = x.to(dev).requires_grad_(True)
dcgan.discriminator(xg)          # create variables
acts.clear()
f = t(xg)
P = {}
for name, v in dcgan.discriminator.named_variables().items():
    layer, leaf = name.rsplit("/", 1)
    P.setdefault(layer, {})[leaf] = v.detach().double().cpu()
x64 = x.double().requires_grad_(True)
ra = []
h = x64
for i, (filters, s, act) in enumerate(dcgan._CRITIC):
    h = NT.conv2d(h, P[f"discriminator/conv2d_{i}"], "crelu" if act else None, s); h.retain_grad(); ra.append(h)
f_ref = NT.feature_head(h)
gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
f.backward(gy.to(dev)); f_ref.backward(gy.double())
for i, (a, r) in enumerate(zip(acts, ra)):
    d = (a.grad.double().cpu() - r.grad).abs().amax(3)
    bad = (d > 1e-4 * r.grad.abs().max()).nonzero()
    print(f"act{i} {tuple(a.shape)} fwd {rel(a, r):.2e} grad {rel(a.grad, r.grad):.2e} bad pixels {len(bad)} {bad[:12].tolist()}")
print("dx", rel(xg.grad, x64.grad))

# ---- re-run conv3's dgrad standalone on the in-net tensors, with a NaN-poisoned workspace
from otgan_amd import ops
import ctypes
x2 = acts[2].detach()
dy3 = acts[3].grad.detach()
V, g, b = [dcgan.discriminator.store.vars[f"discriminator/conv2d_3/{k}"] for k in "Vgb"]
V2d = V.detach().contiguous().view(-1, 1024)
w, wT, inv = ops.weightnorm_fwd(V2d, g.detach())
desc = ops.make_desc(x2, 512, False, 5, 5, 2, 1024, 1024, 0, 1)
ref = ra[2].grad
for poison in (False, True, True):
    for key, buf in ops._ws.items():
        if poison: buf.view(torch.float32)[:].fill_(float("nan"))
    dx = torch.empty_like(x2)
    filt = ops.prepare_filters(desc, 1, w) if poison else None
    ops.conv_dgrad_raw(desc, dy3, w, x2, None, dx, 512, False, filt)
    bad = torch.isnan(dx).any(3).nonzero()
    d = (dx.double().cpu() - ref).abs().amax(3)
    print("poison", poison, "prep", filt is not None, "nan pixels", len(bad), bad[:8].tolist(), "rel", rel(torch.nan_to_num(dx), ref),
          "bad", (d > 1e-4 * ref.abs().max()).nonzero()[:8].tolist())
e = (dx.double().cpu() - ref)[1, 4, 5]
print("pixel err: n bad channels", int((e.abs() > 1e-6).sum()), "max", float(e.abs().max()), "ref max", float(ref[1,4,5].abs().max()))
idx = e.abs().topk(8).indices
print("top channels", idx.tolist(), e[idx].tolist(), ref[1,4,5][idx].tolist(), "x", x2[1,4,5][idx].tolist())
print("dy3 stats: zeros", int((dy3 == 0).sum()), "absmax", float(dy3.abs().max()), "x2 zeros", int((x2 == 0).sum()))
# neighbourhood of dy that feeds tile 21 / pixel: dy rows 2..3? print magnitude
print("dy3[1] per-pixel absmax:\n", dy3[1].abs().amax(2))
