"""Worker of tests/test_engine_accuracy_gpu.py: the well-conditioned whole step (ELU, lambda = 20, 10 sweeps) of the DCGAN
trainer under whatever GEMM-engine environment the parent set, compared tensor by tensor with the fp64 oracle step
(oracle/train_step_cpu.py) on identical parameters, data and latents.  Prints one JSON line:
{"disc": {name: rel. L2 error}, "gen": {...}, "dist": {...}}."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.train_step_cpu import CpuOTGAN  # noqa: E402
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    lam, iters = 20.0, 10
    out = {"dist": {}}
    for kind in ("disc", "gen"):
        args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters,
                            nr_gen_per_disc=1, seed=seed, nonlinearity="elu", image_size=size)
        m = OTGAN(args, dev)
        if kind == "gen":
            m.step_counter = 1
        gen = torch.Generator().manual_seed(12)
        x = torch.rand(m.nb, size, size, 3, generator=gen) * 2 - 1
        noise = torch.rand(m.nb, 100, generator=gen) * 2 - 1
        r = m.step(x.to(dev), noise=noise.to(dev), apply_updates=False)
        o = CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False, image_size=size)
        named = {}
        named.update(m.discriminator.named_variables())
        named.update(m.generator.named_variables())
        o.load(named)
        gr, dist, ent = o.grads(kind, x.double(), noise.double(), 2, lam, iters)
        names = list((m.generator if kind == "gen" else m.discriminator).named_variables())
        out[kind] = {n: rel(a, b) for n, a, b in zip(names, r["grads"], gr)}
        out["dist"][kind] = abs(float(r["distance"]) - dist) / abs(dist)
        m.close()
    print("ENGINE_JSON " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
