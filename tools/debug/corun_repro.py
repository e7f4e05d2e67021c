"""dev: does a kernel running BESIDE the Winograd-domain GEMM on the same GPU (second stream of one process) compute
wrong values?  Stream A loops a GEMM-heavy convolution layer, stream B loops a victim layer on fixed inputs and checks
every result against the first one.
    python tools/debug/corun_repro.py [victim: rgbin|glu|head|s2] [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
_lib.lib()
victim = sys.argv[1] if len(sys.argv) > 1 else "rgbin"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator().manual_seed(1)


def params(k, cin, cout):
    return ((torch.randn(k, k, cin, cout, generator=g) * 0.05).to(dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev))


# co-runner: a strided critic layer at batch 4 (tiny M, K = 4096: long GEMM with few workgroups, like the two-rank test)
xa = torch.randn(4, 8, 8, 512, generator=g).to(dev)
Va, ga, ba = params(5, 1024, 1024)
xa2 = torch.randn(4, 16, 16, 256, generator=g).to(dev)
Va2, ga2, ba2 = params(5, 512, 512)


def corun():
    return (ops.conv2d_op(xa, Va, ga, ba, stride=2, preact=ops.ACT["crelu"]),
            ops.conv2d_op(xa2, Va2, ga2, ba2, stride=2, preact=ops.ACT["crelu"]))


if victim == "rgbin":
    xv = (torch.rand(4, 32, 32, 3, generator=g) * 2 - 1).to(dev)
    Vv, gv, bv = params(5, 3, 128)
    run_victim = lambda: ops.conv2d_op(xv, Vv, gv, bv, stride=1, preact=ops.ACT[None])
elif victim == "rgbin3":
    xv = (torch.rand(4, 32, 32, 3, generator=g) * 2 - 1).to(dev)
    Vv, gv, bv = params(3, 3, 128)
    run_victim = lambda: ops.conv2d_op(xv, Vv, gv, bv, stride=1, preact=ops.ACT[None])
elif victim in ("rgbout", "rgbin_grad", "growth"):
    if victim == "rgbout":
        xv = torch.randn(4, 32, 32, 128, generator=g).to(dev).requires_grad_(True)
        Vv, gv, bv = params(5, 128, 3)
        kw = dict(stride=1, preact=ops.ACT[None])
    elif victim == "rgbin_grad":
        xv = (torch.rand(4, 32, 32, 3, generator=g) * 2 - 1).to(dev).requires_grad_(True)
        Vv, gv, bv = params(5, 3, 128)
        kw = dict(stride=1, preact=ops.ACT[None])
    else:
        xv = torch.randn(4, 32, 32, 64, generator=g).to(dev).requires_grad_(True)
        Vv, gv, bv = params(3, 128, 16)
        kw = dict(stride=1, preact=ops.ACT["crelu"])
    Vv.requires_grad_(True)
    dyv = None

    def run_victim():
        global dyv
        with torch.enable_grad():
            y = ops.conv2d_op(xv, Vv, gv, bv, **kw)
            if dyv is None:
                dyv = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev)
            dx, dV = torch.autograd.grad(y, [xv, Vv], dyv)
        return torch.cat([y.detach().reshape(-1), dx.reshape(-1), dV.reshape(-1)])
elif victim == "up":
    # a folded 5x5 upsampling layer of the generator, forward and both gradients: the transform kernels of winograd.hip
    # (the only translation unit still built with packed fp32, csrc/Makefile) beside the other stream's GEMM
    xv = torch.randn(4, 16, 16, 256, generator=g).to(dev).requires_grad_(True)
    Vv, gv, bv = params(5, 256, 256)
    Vv.requires_grad_(True)
    dyv = torch.randn(4, 32, 32, 256, generator=g).to(dev)

    def run_victim():
        with torch.enable_grad():
            y = ops.conv2d_op(xv, Vv, gv, bv, stride=1, upsample=True, preact=ops.ACT[None])
            dx, dV = torch.autograd.grad(y, [xv, Vv], dyv)
        return torch.cat([y.detach().reshape(-1), dx.reshape(-1), dV.reshape(-1)])
elif victim == "matching":
    from otgan_amd.utils import matching
    fa = torch.nn.functional.normalize(torch.rand(128, 8192, generator=g), dim=1).to(dev)
    fb = torch.nn.functional.normalize(torch.rand(128, 8192, generator=g), dim=1).to(dev)

    def run_victim():
        ga_, gb_, ent, dist = matching.matched_feature_grads(fa, fb, 500.0, 100)
        return torch.cat([ga_.reshape(-1), gb_.reshape(-1), ent.reshape(-1).float(), dist.reshape(-1).float()])
elif victim == "matching256":
    # N = 256: the persistent multi-workgroup Sinkhorn (sinkhorn_panel_kernel) and the split-precision cost / plan GEMMs
    from otgan_amd.utils import matching
    fa = torch.nn.functional.normalize(torch.rand(512, 4096, generator=g), dim=1).to(dev)
    fb = torch.nn.functional.normalize(torch.rand(512, 4096, generator=g), dim=1).to(dev)

    def run_victim():
        ga_, gb_, ent, dist = matching.matched_feature_grads(fa, fb, 500.0, 50)
        return torch.cat([ga_.reshape(-1), gb_.reshape(-1), ent.reshape(-1).float()])
elif victim == "adam":
    # the gathered Adam step with the fused EMA (otgan_adam_step_gather_f32) from fixed state
    sizes = [128 * 1024, 5 * 5 * 256 * 256, 256, 3 * 1000 + 7]
    offs = [0]
    for n_ in sizes:
        offs.append(offs[-1] + n_)
    p0 = torch.randn(offs[-1], generator=g).to(dev)
    grads = [torch.randn(n_, generator=g).to(dev) for n_ in sizes]
    v0, m0, e0 = torch.randn(offs[-1], generator=g).to(dev), torch.rand(offs[-1], generator=g).to(dev), torch.randn(offs[-1], generator=g).to(dev)

    def run_victim():
        p, v, mg, e = p0.clone(), v0.clone(), m0.clone(), e0.clone()
        ops.adam_step_gather(p, grads, offs, v, mg, 3e-4, 0.5, 0.999, 3.0, e, 0.999)
        return torch.cat([p, v, mg, e])
elif victim == "wn":
    # weight norm forward + backward (column reductions over [K][Cout]) through a small direct convolution
    xv = torch.randn(4, 8, 8, 64, generator=g).to(dev)
    Vv = (torch.randn(3, 3, 64, 96, generator=g) * 0.05).to(dev).requires_grad_(True)
    gv = torch.rand(96, generator=g).add(0.5).to(dev).requires_grad_(True)
    bv = torch.zeros(96, device=dev).requires_grad_(True)
    dyv = torch.randn(4, 8, 8, 96, generator=g).to(dev)

    def run_victim():
        ops.bump_weights_epoch()
        with torch.enable_grad():
            y = ops.conv2d_op(xv, Vv, gv, bv, stride=1, preact=ops.ACT[None])
            dV, dg, db = torch.autograd.grad(y, [Vv, gv, bv], dyv)
        return torch.cat([y.detach().reshape(-1), dV.reshape(-1), dg.reshape(-1), db.reshape(-1)])
elif victim == "glu":
    xv = torch.randn(4, 16, 16, 512, generator=g).to(dev)
    run_victim = lambda: ops.glu(xv)
elif victim == "head":
    xv = torch.randn(4, 4, 4, 1024, generator=g).to(dev)
    run_victim = lambda: ops.feature_head(xv)
else:
    xv = torch.randn(4, 32, 32, 128, generator=g).to(dev)
    Vv, gv, bv = params(5, 256, 256)
    run_victim = lambda: ops.conv2d_op(xv, Vv, gv, bv, stride=2, preact=ops.ACT["crelu"])

with torch.no_grad():
    gref = [t.clone() for t in corun()]
    ref = run_victim().clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = gbad = 0
    for it in range(iters):
        with torch.cuda.stream(sa):
            gouts = [corun() for _ in range(3)]
        with torch.cuda.stream(sb):
            outs = [run_victim() for _ in range(12)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                d = (o != ref)
                idx = d.nonzero()
                bad += 1
                if bad <= 5:
                    print(f"iter {it}: {int(d.sum())} elements differ, index range {idx.min(0).values.tolist()} .. {idx.max(0).values.tolist()}", flush=True)
        for go in gouts:      # the GEMM layers themselves as the victim
            for t, r in zip(go, gref):
                if not torch.equal(t, r):
                    gbad += 1
                    if gbad <= 5:
                        print(f"iter {it}: GEMM-side layer output differs in {int((t != r).sum())} elements", flush=True)
    print("CORUN", victim, "mismatching results:", bad, "of", iters * 12, "; GEMM-side mismatches:", gbad, "of", iters * 6, flush=True)
