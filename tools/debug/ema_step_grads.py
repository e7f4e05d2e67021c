import os, sys, torch, pytest
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_train_step_gpu as T
from otgan_amd.trainer import OTGAN, default_args
dev = torch.device("cuda:0")
lam, iters = 20.0, 10
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters, nr_gen_per_disc=1, seed=8, nonlinearity="elu", train_disc_against_ema=True, learning_rate_gen=0.05)
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(13)
x = (torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev)
u = (torch.rand(m.nb, 100, generator=gen) * 2 - 1).to(dev)
m.step(x, noise=u); m.step(x, noise=u)
names_g = list(m.generator.named_variables())
shadow = {n: m.ema.average(p) for n, p in zip(names_g, m.gen_params)}
r = m.step(x, noise=u, apply_updates=False)
o = T.CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False)
o.load(T._named(m))
gr, dist, ent = o.grads("disc", x.double().cpu(), u.double().cpu(), 2, lam, iters, ema_P=o.ema_params(shadow))
names = list(m.discriminator.named_variables())
for n, a, b in zip(names, r["grads"], gr):
    print(f"{n:40s} rel {T._rel(a, b):.2e}  |ref| {float(b.norm()):.3e}")
