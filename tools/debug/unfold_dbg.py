"""dev / tests: weight gradients of three 5x5 upsampling layers -> torch.save(file)  (tests/test_layers_gpu.py runs it
with OTGAN_WINO_UNFOLD_FUSED = 1 and 0 and compares)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(4)
outs = []
for (N, H, C, Co) in ((3, 4, 64, 64), (2, 8, 32, 96), (5, 16, 64, 32)):
    x = torch.randn(N, H, H, C, generator=g).to(dev)
    V = (torch.randn(5, 5, C, Co, generator=g) * 0.05).to(dev).requires_grad_(True)
    gg = torch.ones(Co, device=dev, requires_grad=True); b = torch.zeros(Co, device=dev, requires_grad=True)
    dy = torch.randn(N, 2 * H, 2 * H, Co, generator=g).to(dev)
    y = ops.conv2d_op(x, V, gg, b, stride=1, upsample=True, preact=0)
    y.backward(dy)
    outs.append(V.grad.cpu())
torch.save(outs, sys.argv[1])
