"""dev: where does the critic-step gradient error of the well-conditioned step come from?  Features, injected gradient
(matching on the SAME features), backward (same upstream gradient) -- each against fp64."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import matching_np as M  # noqa: E402
from oracle.train_step_cpu import CpuOTGAN  # noqa: E402
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402
from otgan_amd.utils import matching  # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


dev = torch.device("cuda:0")
lam, iters = 20.0, 10
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters, nr_gen_per_disc=1,
                    seed=5, nonlinearity="elu")
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(12)
x = torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1
noise = torch.rand(m.nb, 100, generator=gen) * 2 - 1
o = CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False)
named = {}
named.update(m.discriminator.named_variables())
named.update(m.generator.named_variables())
o.load(named)
with torch.no_grad():
    xg = m.generator(batch_size=m.nb, noise=noise.to(dev), device=dev, nonlinearity="elu")
    xg_o = o.gen(noise.double())
print("x_gen", rel(xg, xg_o))
xin = torch.cat([x.to(dev), xg], 0)
f = m.discriminator(xin, nonlinearity="elu")
f_o = o.disc(torch.cat([x.double(), xg.double().cpu()], 0))          # oracle critic on the HIP generator's images
print("features (same images)", rel(f, f_o))
nb = m.nb
# matching on the SAME (HIP) features: HIP grad entry vs fp64 oracle
ga, gb, ent, dist = matching.matched_feature_grads(f[nb:].detach(), f[:nb].detach(), lam, iters)
fo64 = f.detach().double().cpu()
g_gen_o, g_dat_o, dist_o, ent_o = o.match(fo64[nb:], fo64[:nb], 2, lam, iters)
print("injected grads on the same features: gen", rel(ga, g_gen_o), "dat", rel(gb, g_dat_o), "dist", abs(float(dist) - dist_o) / abs(dist_o))
# and from the four matched arrays (the round-2 route)
out = matching.get_matched_features(list(torch.chunk(f[nb:].detach(), 2)), list(torch.chunk(f[:nb].detach(), 2)), lam, iters)
ga2 = torch.cat(out[0]) - torch.cat(out[2])
print("  same through f_aa - f_ab of the inference entry", rel(ga2, g_gen_o))
# backward with the SAME upstream gradient (the oracle's, from HIP features)
up = torch.cat([g_dat_o, g_gen_o], 0)
gr = torch.autograd.grad(f, m.disc_params, up.float().to(dev))
gr_o = torch.autograd.grad(f_o, o.params("discriminator"), up)
names = list(m.discriminator.named_variables())
for n, a, b in zip(names, gr, gr_o):
    print(f"  backward only {n:30s} {rel(a, b):.2e}")
