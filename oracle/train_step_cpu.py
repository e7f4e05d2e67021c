"""CPU restatement of one OT-GAN training step (reference train.py:70-149,207-226) on the
oracles: PyTorch-CPU nets (oracle/nets_torch.py) + the matching oracle.

TEST INFRASTRUCTURE ONLY: used by tests/ (step-level parity) and by the `cpu_baseline` leg
of bench.py (timed on the GPU box's host cores).  Never imported by the product path."""
import time

import numpy as np
import torch

from . import matching_np as M
from . import nets_torch as NT


class CpuOTGAN:
    def __init__(self, model="dcgan", nonlinearity="crelu", seed=1, dtype=torch.float32, use_c_matching=True,
                 image_size=32):
        g = torch.Generator().manual_seed(seed)
        self.model, self.nl, self.dtype = model, nonlinearity, dtype
        if model == "dcgan":
            self.P = NT.init_params(NT.dcgan_disc_shapes(nonlinearity), "discriminator", g, dtype)
            self.P.update(NT.init_params(NT.dcgan_gen_shapes(image_size), "generator", g, dtype))
        else:
            self.P = NT.init_params(NT.densenet_disc_shapes(nonlinearity), "discriminator", g, dtype)
            self.P.update(NT.init_params(NT.densenet_gen_shapes(nonlinearity), "generator", g, dtype))
        for lay in self.P.values():
            for t in lay.values():
                t.requires_grad_(True)
        self.use_c = use_c_matching and dtype == torch.float32

    def load(self, named):
        """named: {'discriminator/conv2d_0/V': tensor, ...}"""
        with torch.no_grad():
            for k, v in named.items():
                lay, leaf = k.rsplit("/", 1)
                self.P[lay][leaf].copy_(v.detach().to(self.dtype).cpu())

    def params(self, scope):
        return [t for k, lay in self.P.items() if k.startswith(scope) for t in (lay["V"], lay["g"], lay["b"])]

    def disc(self, x):
        return (NT.dcgan_discriminator(x, self.P, self.nl) if self.model == "dcgan"
                else NT.densenet_discriminator(x, self.P, self.nl))

    def gen(self, noise, P=None):
        P = self.P if P is None else P
        return (NT.dcgan_generator(noise, P) if self.model == "dcgan"
                else NT.densenet_generator(noise, P, self.nl))

    def ema_params(self, named_shadow):
        """Generator parameter dict built from EMA shadows {'generator/conv2d_0/V': tensor, ...}
        (train.py:75: `generator(ema=ema)` reads every variable through ema.average)."""
        P = {}
        for k, v in named_shadow.items():
            lay, leaf = k.rsplit("/", 1)
            P.setdefault(lay, {})[leaf] = v.detach().to(self.dtype).cpu()
        return P

    def match(self, f_gen, f_dat, S, lam, iters):
        fa, fb = f_gen.detach().numpy(), f_dat.detach().numpy()
        if self.use_c:
            from . import sinkhorn_c as C
            aa, bb, ab, ba, ent, dist = C.two_batch(fa, fb, lam, iters)
        else:
            out = M.get_matched_features(list(np.split(fa, S)), list(np.split(fb, S)), lam, iters, dtype=fa.dtype.type)
            dist = M.calc_distance(list(np.split(fa, S)), list(np.split(fb, S)), out, dtype=fa.dtype.type)
            aa, bb, ab, ba = [np.concatenate(o, 0) for o in out[:4]]
            ent = out[4]
        t = lambda z: torch.as_tensor(z, dtype=self.dtype)
        return t(aa) - t(ab), t(bb) - t(ba), float(dist), float(ent)

    def grads(self, kind, x_data, noise, S, lam, iters, ema_P=None):
        """Gradient lists exactly as train.py:108-130 injects them (summed over shards).
        `ema_P` (critic step only): --train_disc_against_ema, the generated branch comes from the
        EMA generator (train.py:102-103,119-123)."""
        if kind == "gen":
            x_gen = self.gen(noise)
            with torch.no_grad():
                f_dat = self.disc(x_data)
            f_gen = self.disc(x_gen)
            g_gen, _, dist, ent = self.match(f_gen, f_dat, S, lam, iters)
            gr = torch.autograd.grad(f_gen, self.params("generator"), g_gen)
        else:
            with torch.no_grad():
                x_gen = self.gen(noise, ema_P)
            f_all = self.disc(torch.cat([x_data, x_gen], 0))
            nb = x_data.shape[0]
            g_gen, g_dat, dist, ent = self.match(f_all[nb:], f_all[:nb], S, lam, iters)
            gr = torch.autograd.grad(f_all, self.params("discriminator"), torch.cat([g_dat, g_gen], 0))
        return gr, dist, ent


def time_cpu_steps(batch_per_shard=8, shards=2, lam=500.0, iters=100, model="dcgan", repeats=1):
    """Wall time of one generator step and one critic step on the host CPU (fp32, all
    cores torch gives us).  Returns images/sec at the reference's 5:1 step mix."""
    torch.manual_seed(0)
    m = CpuOTGAN(model)
    nb = batch_per_shard * shards
    x = torch.rand(nb, 32, 32, 3) * 2 - 1
    if model == "dcgan":
        noise = torch.rand(nb, 100) * 2 - 1
    else:
        noise = [torch.rand(nb, 100) * 2 - 1, torch.rand(nb, 8, 8, 16) * 2 - 1,
                 torch.rand(nb, 16, 16, 16) * 2 - 1, torch.rand(nb, 32, 32, 16) * 2 - 1]
    best = {}
    for kind in ("gen", "disc"):
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            m.grads(kind, x, noise, shards, lam, iters)
            ts.append(time.perf_counter() - t0)
        best[kind] = min(ts)
    ips = 6.0 * nb / (5.0 * best["gen"] + best["disc"])
    return ips, best, nb
