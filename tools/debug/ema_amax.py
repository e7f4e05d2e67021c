import os, sys, torch
sys.path.insert(0, "/root/repo")
from otgan_amd.trainer import OTGAN, default_args
dev = torch.device("cuda:0")
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=20.0, nr_sinkhorn_iter=10, nr_gen_per_disc=1, seed=8, nonlinearity="elu", train_disc_against_ema=True, learning_rate_gen=0.05)
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(13)
x = (torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev)
u = (torch.rand(m.nb, 100, generator=gen) * 2 - 1).to(dev)
m.step(x, noise=u); m.step(x, noise=u)
torch.cuda.synchronize()
print("=== step 3", file=sys.stderr, flush=True)
os.environ["OTGAN_AMAX_GO"] = "1"
r = m.step(x, noise=u, apply_updates=False)
torch.cuda.synchronize()
