"""dev check: DenseNet critic parity numbers (features, dx, worst parameter gradients) of
tests/test_models_gpu.py::test_densenet_critic_parity over several parameter seeds, under the current environment."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_models_gpu import _oracle_params, _rel
from oracle import nets_torch as NT
from otgan_amd.models import densenet
dev = torch.device("cuda:0")
L = 16
for seed in (5, 6, 7):
    densenet.discriminator.reset(seed=seed)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(2, 32, 32, 3, generator=gen) * 2 - 1
    xg = x.to(dev).requires_grad_(True)
    f = densenet.discriminator(xg, nonlinearity="crelu", layers_per_block=L)
    P = _oracle_params(densenet.discriminator)
    x64 = x.double().requires_grad_(True)
    f_ref = NT.densenet_discriminator(x64, P, "crelu", L)
    gy = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float()
    params = densenet.discriminator.trainable_variables()
    got = torch.autograd.grad(f, [xg] + params, gy.to(dev))
    names = list(densenet.discriminator.named_variables())
    leaves = [x64] + [P[n.rsplit("/", 1)[0]][n.rsplit("/", 1)[1]] for n in names]
    ref = torch.autograd.grad(f_ref, leaves, gy.double())
    errs = sorted(((_rel(a, r), n) for n, a, r in zip(["dx"] + names, got, ref)), reverse=True)
    print("seed", seed, "f", f"{_rel(f, f_ref):.2e}", "worst", [(n, f"{e:.2e}") for e, n in errs[:3]], flush=True)
