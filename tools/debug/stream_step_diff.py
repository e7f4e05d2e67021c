"""dev tool: one trainer state, generator / critic forward + critic-step gradients, dumped to an .npz so that two
runs under different OTGAN_X3_STREAM settings can be diffed tensor by tensor."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd.trainer import OTGAN, default_args
dev = torch.device("cuda:0")
args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=20.0, nr_sinkhorn_iter=10, nr_gen_per_disc=1,
                    seed=8, nonlinearity="elu", train_disc_against_ema=True, learning_rate_gen=0.05)
m = OTGAN(args, dev)
gen = torch.Generator().manual_seed(13)
x = (torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev)
u = (torch.rand(m.nb, 100, generator=gen) * 2 - 1).to(dev)
out = {}
r = m.step(x, noise=u, apply_updates=False)
out["dist0"] = np.array([float(r["distance"]), float(r["entropy"])])
for n, g in zip(m.discriminator.named_variables(), r["grads"]):
    out["g0." + n] = g.detach().cpu().numpy()
m.step(x, noise=u); m.step(x, noise=u)
# the same weights, the two GEMM variants (OTGAN_X3_STREAM is read per launch)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["OTGAN_X3_STREAM"] = mode
    r = m.step(x, noise=u, apply_updates=False)
    if r["kind"] != "disc":
        r = m.step(x, noise=u, apply_updates=False)
    assert r["kind"] == "disc"
    res.setdefault(mode, []).append([g.detach().double().cpu() for g in r["grads"]])
names = list(m.discriminator.named_variables())
rel = lambda a, b: float((a - b).norm() / b.norm())
for i, n in enumerate(names):
    print(f"{n:36s} stream vs one-tile {rel(res['1'][0][i], res['0'][0][i]):.2e}   repeat one-tile {rel(res['0'][1][i], res['0'][0][i]):.2e}"
          f"   repeat stream {rel(res['1'][1][i], res['1'][0][i]):.2e}")
