"""dev: generator output at 64 x 64 (test parameters, seed 7) against the fp64 oracle generator"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_train_step_gpu as T
from otgan_amd.trainer import OTGAN, default_args
dev = torch.device('cuda:0')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=20.0, nr_sinkhorn_iter=10, nr_gen_per_disc=1, seed=seed, nonlinearity="elu", image_size=64)
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(7 + seed)
x = torch.rand(m.nb, 64, 64, 3, generator=gen) * 2 - 1
noise = T._noise("dcgan", m.nb, gen)
with torch.no_grad():
    xg = m.generator(batch_size=m.nb, noise=noise.to(dev), **m.model_opts)
o = T.CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False, image_size=64)
o.load(T._named(m))
with torch.no_grad():
    xr = o.gen(noise.double())
d = (xg.double().cpu() - xr).abs()
print("x_gen: max abs diff %.3e  rel L2 %.3e" % (float(d.max()), float(d.norm() / xr.norm())), "argmax", [int(i) for i in torch.unravel_index(d.argmax(), d.shape)])
bad = (d > 1e-4).nonzero()
print("n > 1e-4:", len(bad), bad[:8].tolist())
