"""dev: which lanes / vector elements of conv_fewout_dgrad_kernel come out wrong beside the GEMM (corun_repro.py rgbout)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import _lib, ops
dev = torch.device("cuda:0"); _lib.lib()
g = torch.Generator().manual_seed(1)
def params(k, cin, cout):
    return ((torch.randn(k, k, cin, cout, generator=g) * 0.05).to(dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev))
xa = torch.randn(4, 8, 8, 512, generator=g).to(dev); Va, ga, ba = params(5, 1024, 1024)
xa2 = torch.randn(4, 16, 16, 256, generator=g).to(dev); Va2, ga2, ba2 = params(5, 512, 512)
def corun():
    ops.conv2d_op(xa, Va, ga, ba, stride=2, preact=ops.ACT["crelu"])
    ops.conv2d_op(xa2, Va2, ga2, ba2, stride=2, preact=ops.ACT["crelu"])
K = int(os.environ.get("KSZ", "5"))
xv = torch.randn(4, 32, 32, 128, generator=g).to(dev).requires_grad_(True)
Vv, gv, bv = params(K, 128, 3)
dyv = torch.randn(4, 32, 32, 3, generator=g).to(dev)
def victim():
    with torch.enable_grad():
        y = ops.conv2d_op(xv, Vv, gv, bv, stride=1, preact=ops.ACT[None])
        dx, = torch.autograd.grad(y, [xv], dyv)
    return dx
corun(); ref = victim().clone(); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
hq, he, hcg, hrow = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
bad = 0; relmax = 0.0
for it in range(40):
    with torch.cuda.stream(sa):
        for _ in range(3): corun()
    with torch.cuda.stream(sb):
        outs = [victim() for _ in range(12)]
    torch.cuda.synchronize()
    for o in outs:
        d = (o != ref)
        if d.any():
            bad += 1
            idx = d.nonzero()
            for n, h, w, c in idx[:2000].tolist():
                hq[(c % 64) // 4] += 1; he[c % 4] += 1; hcg[(w % 16) // 4] += 1; hrow[h % 4] += 1
            relmax = max(relmax, float(((o - ref).abs().max()) / ref.abs().max()))
print("bad launches", bad, "of 480; max |diff| / max |ref| =", relmax)
print("quad q:", sorted(hq.items())); print("element:", sorted(he.items())); print("col group (lane>>4):", sorted(hcg.items())); print("row (wave):", sorted(hrow.items()))
