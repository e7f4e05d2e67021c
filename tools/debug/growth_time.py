"""dev tool: forward / dgrad / wgrad time of one growth layer (3x3, 16 outputs, CReLU over k earlier 16-channel
outputs, y_accumulate) at the DenseNet stage shapes, HIP events around 20 launches."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import ops
from otgan_amd._lib_layers import ConvDesc
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, H, C0) in ((512, 8, 200), (256, 8, 32), (512, 16, 144), (256, 16, 160), (512, 32, 32), (256, 32, 224)):
    Ctot = C0 + 256
    buf = torch.randn(N, H, H, Ctot, device=dev)
    G = torch.randn(N, H, H, Ctot, device=dev)
    row = []
    for k in (1, 2, 4, 7):
        desc = ConvDesc(N, H, H, 16 * k, Ctot, 0, 3, 3, 1, 16, Ctot, C0 + 16 * 8, 1, 1)
        desc.y_accumulate = 1
        cmap, inv = ops.channel_maps((16,) * k, 1, dev)
        w = torch.randn(9 * 32 * k, 16, device=dev) * 0.05
        wT = w.t().contiguous()
        dw = torch.empty_like(w)
        src, gsrc = buf[..., C0:], G[..., C0:]
        f = t(lambda: ops.conv_fwd_raw(desc, src, cmap, wT, None, buf))
        d = t(lambda: ops.conv_dgrad_raw(desc, G, w, src, inv, gsrc, Ctot, True))
        g = t(lambda: ops.conv_wgrad_raw(desc, src, cmap, G, dw))
        row.append(f"k={k}: {f:6.1f} {d:6.1f} {g:6.1f}")
    print(f"N={N} {H}x{H} C0={C0} | fwd dgrad wgrad us | " + " | ".join(row), flush=True)
