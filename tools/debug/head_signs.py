"""dev: pre-head activations of the 64 x 64 critic step (test parameters) on the HIP path against the fp64 oracle: how many
CReLU signs differ, and how large the activations are where they do"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_train_step_gpu as T
from otgan_amd.trainer import OTGAN, default_args
from otgan_amd.utils import nn as hip_nn
from oracle import nets_torch as NTO
dev = torch.device('cuda:0')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=20.0, nr_sinkhorn_iter=10, nr_gen_per_disc=1, seed=seed, nonlinearity="elu", image_size=64)
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(7 + seed)
x = torch.rand(m.nb, 64, 64, 3, generator=gen) * 2 - 1
noise = T._noise("dcgan", m.nb, gen)
zs, xs = [], []
real_head, real_ohead = hip_nn.feature_head, NTO.feature_head
hip_nn.feature_head = lambda z: (zs.append(z.detach().double().cpu()), real_head(z))[1]
NTO.feature_head = lambda z: (xs.append(z.detach()), real_ohead(z))[1]
r = m.step(x.to(dev), noise=noise.to(dev), apply_updates=False)
o = T.CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False, image_size=64)
o.load(T._named(m))
gr, dist, ent = o.grads("disc", x.double(), noise.double(), 2, 20.0, 10)
z, xr = zs[0], xs[0]
d = (z - xr).abs()
print("pre-head activations: shape", tuple(z.shape), "max abs diff %.3e  rel L2 %.3e  max |x| %.3e" % (float(d.max()), float(d.norm() / xr.norm()), float(xr.abs().max())))
mis = ((z > 0) != (xr > 0)).nonzero()
print("sign mismatches:", len(mis))
for i in mis[:10].tolist():
    print("   ", i, "hip %.3e  fp64 %.3e" % (float(z[tuple(i)]), float(xr[tuple(i)])))
