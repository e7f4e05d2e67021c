"""dev: few-channel weight gradients (DCGAN G.conv3 128 -> 3 and D.conv0 3 -> 128, 5x5; DenseNet conv_50 480 CReLU -> 3, 3x3)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
dev = torch.device('cuda:0')
def run(N,H,W,C,cout,k,pre,tag):
    torch.manual_seed(0)
    mult = 2 if pre in ("crelu","celu") else 1
    x = torch.randn(N,H,W,C, device=dev)
    V = (torch.randn(k,k,C*mult,cout, device=dev)*0.05).requires_grad_(True)
    g = torch.ones(cout, device=dev); b = torch.zeros(cout, device=dev)
    y = ops.conv2d_op(x, V, g, b, stride=1, upsample=False, preact=ops.ACT[pre], segs=(C,))
    dy = torch.randn_like(y)
    f = lambda: torch.autograd.grad(y, [V], dy, retain_graph=True)[0]
    for _ in range(3): o = f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): o = f()
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    print(tag, os.environ.get("OTGAN_DISABLE_OUTER_MFMA"), "%.1f us" % (dt*1e6), float(o.double().abs().sum()))
run(256,32,32,128,3,5,None,"dcgan G.conv3 wgrad")
run(256,32,32,3,128,5,None,"dcgan D.conv0 wgrad")
run(256,32,32,480,3,3,"crelu","densenet conv_50 wgrad")
