"""dev: time of the split-precision cost GEMM launch (+ finish) per problem shape; OTGAN_X3_COST_SPLITS forces the K split.
    python tools/debug/cost_split_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from otgan_amd import _lib  # noqa: E402
from otgan_amd.utils import matching  # noqa: E402

dev = torch.device("cuda:0")
_lib.lib()
g = torch.Generator(device=dev).manual_seed(3)
out = []
for name, P, n, m, D in (("rank3 256x1024 D32768", 3, 256, 1024, 32768), ("rank3 256x1024 D7296", 3, 256, 1024, 7296),
                         ("six 1024x1024 D32768", 6, 1024, 1024, 32768), ("six 256x256 D131072", 6, 256, 256, 131072),
                         ("six 512x512 D32768", 6, 512, 512, 32768)):
    X = [torch.nn.functional.normalize(torch.rand(n, D, device=dev, generator=g), dim=1) for _ in range(2)]
    Y = [torch.nn.functional.normalize(torch.rand(m, D, device=dev, generator=g), dim=1) for _ in range(2)]
    xs, ys = [X[i % 2] for i in range(P)], [Y[(i // 2) % 2] for i in range(P)]
    for _ in range(3):
        K = matching.cost_log_kernels(xs, ys, 500.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K = matching.cost_log_kernels(xs, ys, 500.0)
    e1.record()
    torch.cuda.synchronize()
    ref = -500.0 * (1.0 - X[0].double() @ Y[0].double().T)
    err = float((K[0].double() - ref).abs().max())
    out.append(f"{name}: {e0.elapsed_time(e1) * 100:.0f} us (max |K - fp64| {err:.2e})")
print(f"splits={os.environ.get('OTGAN_X3_COST_SPLITS', 'auto')}  " + " | ".join(out))
