"""dev: RGB-out forward on the matrix-pipe kernel against fp64 (and the VALU tile kernel with OTGAN_DISABLE_FEWOUT_MFMA=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
dev = torch.device('cuda:0')
for (N,H,W,C,k) in [(6,64,64,128,5),(4,32,32,128,5),(2,16,64,64,3),(3,64,64,128,5),(6,64,64,128,5)]:
  for seed in range(3):
    torch.manual_seed(seed)
    x = torch.randn(N,H,W,C, device=dev)
    V = torch.randn(k,k,C,3, device=dev)*0.05
    g = torch.rand(3, device=dev) + 0.5; b = torch.randn(3, device=dev)
    ys = [ops.conv2d_op(x, V, g, b, stride=1, upsample=False, preact=ops.ACT[None], segs=(C,)) for _ in range(4)]
    same = all(torch.equal(ys[0], y) for y in ys[1:])
    y = ys[0]
    Vn = V.double(); w = Vn / Vn.pow(2).sum((0,1,2), keepdim=True).sqrt() * g.double()
    ref = torch.nn.functional.conv2d(x.double().permute(0,3,1,2), w.permute(3,2,0,1), padding=k//2).permute(0,2,3,1) + b.double()
    d = (y.double()-ref).abs()
    bad = (d > 1e-4).nonzero()
    print((N,H,W,C,k), seed, os.environ.get("OTGAN_DISABLE_FEWOUT_MFMA"), "repeatable", same, "rel err %.2e" % float(d.norm()/ref.norm()), "max %.2e" % float(d.max()), "n bad", len(bad),
          "rows", sorted(set(bad[:,1].tolist()))[:12], "cols", sorted(set(bad[:,2].tolist()))[:12])
