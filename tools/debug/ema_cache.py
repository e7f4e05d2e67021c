import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_train_step_gpu as T
from otgan_amd import ops
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
o = T.CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False)
o.load(T._named(m))
gr, dist, ent = o.grads("disc", x.double().cpu(), u.double().cpu(), 2, lam, iters, ema_P=o.ema_params(shadow))
names = list(m.discriminator.named_variables())
def run(tag):
    r = m.step(x, noise=u, apply_updates=False)
    if r["kind"] != "disc":
        r = m.step(x, noise=u, apply_updates=False)
    print(tag, "dist", float(r["distance"]), dist, " worst", max((T._rel(a, b), n) for n, a, b in zip(names, r["grads"], gr)))
run("as is          ")
_m = m._match
_d = ops.conv_dgrad_raw
store = {}
def run2(tag, only):
    cur = {}
    def m2(f_gen, f_dat, pending=None):
        out = _m(f_gen, f_dat, pending)
        cur["f_gen"], cur["f_dat"] = f_gen.double().cpu(), f_dat.double().cpu()
        cur["g_gen"], cur["g_dat"] = out[0].double().cpu(), out[1].double().cpu()
        return out
    def d2(desc, dy, w, x, inv, dx, lddx, accumulate, filters=None):
        if "dy_last" not in cur:
            cur["dy_last"] = dy.double().cpu()
        _d(desc, dy, w, x, inv, dx, lddx, accumulate, filters)
    m._match, ops.conv_dgrad_raw = m2, d2
    os.environ["OTGAN_X3_STREAM_ONLY"] = only
    r = m.step(x, noise=u, apply_updates=False)
    if r["kind"] != "disc":
        cur.clear()
        os.environ["OTGAN_X3_STREAM_ONLY"] = "5555"; os.environ["OTGAN_X3_STREAM_ONLY"] = only
        r = m.step(x, noise=u, apply_updates=False)
    m._match, ops.conv_dgrad_raw = _m, _d
    store[tag] = cur
run2("stream8", "8"); run2("onetile", "999")
for k in ("f_gen", "f_dat", "g_gen", "g_dat", "dy_last"):
    a, b = store["stream8"][k], store["onetile"][k]
    print(f"{k:8s} rel diff {float((a - b).norm() / b.norm()):.2e}   norm {float(b.norm()):.3e}")
g = torch.cat([store["onetile"]["g_dat"], store["onetile"]["g_gen"]], 0); f = torch.cat([store["onetile"]["f_dat"], store["onetile"]["f_gen"]], 0)
par = (g * f).sum(1, keepdim=True) * f
print("|g| per row", g.norm(dim=1)[:4].tolist(), " |g_perp| per row", (g - par).norm(dim=1)[:4].tolist())
