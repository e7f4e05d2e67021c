"""dev: RGB-in forward (DCGAN D.conv0 3 -> 128, 5x5; DenseNet conv2d_0 3 -> 32 is below the 128-channel kernels)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
dev = torch.device('cuda:0')
def run(N,H,W,cout,k,tag):
    torch.manual_seed(0)
    x = torch.rand(N,H,W,3, device=dev)*2-1
    V = torch.randn(k,k,3,cout, device=dev)*0.05
    g = torch.rand(cout, device=dev)+0.5; b = torch.randn(cout, device=dev)*0.1
    with torch.no_grad():
        f = lambda: ops.conv2d_op(x, V, g, b, stride=1, upsample=False, preact=ops.ACT[None], segs=(3,))
        for _ in range(3): y = f()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(20): y = f()
        torch.cuda.synchronize(); dt=(time.time()-t)/20
    w = V.double(); w = w / w.pow(2).sum((0,1,2), keepdim=True).sqrt() * g.double()
    ref = torch.nn.functional.conv2d(x.double().permute(0,3,1,2), w.permute(3,2,0,1), padding=k//2).permute(0,2,3,1) + b.double()
    err = float((y.double()-ref).norm()/ref.norm())
    print(tag, os.environ.get("OTGAN_DISABLE_RGBIN_MFMA"), "%.1f us" % (dt*1e6), "rel err vs fp64 %.2e" % err)
run(256,32,32,128,5,"dcgan D.conv0 fwd")
run(64,64,64,128,5,"64x64 D.conv0 fwd")
run(256,32,32,128,3,"3x3 3->128")
run(6,16,16,256,5,"small 3->256 16x16")
