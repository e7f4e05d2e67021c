"""dev check: the densenet critic-step gradient comparison of tests/test_train_step_gpu.py over several seeds;
prints the worst tensor per seed (name, HIP error vs fp64 oracle, fp32-oracle error)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.train_step_cpu import CpuOTGAN
from otgan_amd.trainer import OTGAN, default_args
from test_train_step_gpu import _noise, _rel
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "disc"
for seed in (7, 8, 9, 10, 11):
    args = default_args(model="densenet", batch_size=3, nr_gpu=2, sinkhorn_lambda=100.0, nr_sinkhorn_iter=20, nr_gen_per_disc=1, seed=seed)
    m = OTGAN(args, dev)
    if kind == "gen":
        m.step_counter = 1
    gen = torch.Generator().manual_seed(seed + 4)
    x = torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1
    noise = _noise("densenet", m.nb, gen)
    r = m.step(x.to(dev), noise=[t.to(dev) for t in noise], apply_updates=False)
    o = CpuOTGAN("densenet", "crelu", dtype=torch.float64, use_c_matching=False)
    named = {}
    named.update(m.discriminator.named_variables()); named.update(m.generator.named_variables())
    o.load(named)
    gr, dist, ent = o.grads(kind, x.double(), [t.double() for t in noise], 2, 100.0, 20)
    names = list((m.generator if kind == "gen" else m.discriminator).named_variables())
    errs = sorted(((_rel(a, b), n) for n, a, b in zip(names, r["grads"], gr)), reverse=True)
    print("seed", seed, "worst", [(n, f"{e:.2e}") for e, n in errs[:3]], "median", f"{errs[len(errs)//2][0]:.2e}", flush=True)
