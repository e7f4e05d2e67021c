import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from otgan_amd import ops
    dev = torch.device('cuda:0')
    outs = {}
    for (N,H,W,C,cout,k) in [(2,16,16,3,128,5),(2,16,32,3,128,5),(2,32,16,3,128,5),(2,8,16,3,64,5),(2,16,16,3,128,3),(1,8,16,3,16,5)]:
        torch.manual_seed(0)
        x = torch.randn(N,H,W,C, device=dev, requires_grad=True)
        V = torch.randn(k,k,C,cout, device=dev)*0.05
        g = torch.ones(cout, device=dev); b = torch.zeros(cout, device=dev)
        y = ops.conv2d_op(x, V, g, b, stride=1, upsample=False, preact=ops.ACT[None], segs=(C,))
        dy = torch.randn(y.shape, device=dev)
        dx, = torch.autograd.grad(y, [x], dy)
        outs[(N,H,W,C,cout,k)] = dx.cpu()
    torch.save(outs, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/a.pt"], env=env)
    env["OTGAN_DISABLE_FEWOUT_MFMA"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/b.pt"], env=env)
    a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
    for k in a:
        d = (a[k]-b[k]).abs()
        bad = (d > 1e-4).nonzero()
        print(k, "max diff", float(d.max()), "ref max", float(b[k].abs().max()), "n bad", len(bad), "rows", sorted(set(bad[:, 1].tolist())), "cols", sorted(set(bad[:, 2].tolist())), "images", sorted(set(bad[:, 0].tolist())))
