import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from otgan_amd import ops
dev = torch.device('cuda:0')
def run(N,H,W,C,k,pre,tag):
    torch.manual_seed(0)
    mult = 2 if pre in ("crelu","celu") else 1
    x = torch.randn(N,H,W,C, device=dev, requires_grad=True)
    V = (torch.randn(k,k,C*mult,3, device=dev)*0.05).requires_grad_(True)
    g = torch.ones(3, device=dev, requires_grad=True); b = torch.zeros(3, device=dev, requires_grad=True)
    y = ops.conv2d_op(x, V, g, b, stride=1, upsample=False, preact=ops.ACT[pre], segs=(C,))
    dy = torch.randn_like(y)
    for _ in range(3): torch.autograd.grad(y, [x], dy, retain_graph=True)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): dx, = torch.autograd.grad(y, [x], dy, retain_graph=True)
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    print(tag, os.environ.get("OTGAN_DISABLE_FEWOUT_DGRAD"), "%.1f us" % (dt*1e6), float(dx.abs().sum()))
run(256,32,32,128,5,None,"dcgan G.conv3 dgrad")
run(256,32,32,480,3,"crelu","densenet conv_50 dgrad")
