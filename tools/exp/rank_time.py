"""A rank of eight's matching call (trainer.rank_matching_stack), timed; usage: rank_time.py [D] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from otgan_amd import trainer as T, _lib
from otgan_amd.utils import matching
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N, rows, W = 1024, 256, 8
g = torch.Generator(device=dev).manual_seed(5)
def feats(n, shift):
    c = torch.rand(32, D, device=dev, generator=g) + shift
    f = (c[torch.randint(0, 32, (n,), device=dev, generator=g)] + 0.1 * torch.randn(n, D, device=dev, generator=g)).abs()
    return torch.nn.functional.normalize(f, dim=1)
fa, fb = feats(2 * N, 0.0), torch.nn.functional.normalize(feats(2 * N, 0.5) ** 2, dim=1)
own = lambda t, r: t[r * rows:(r + 1) * rows]
allk = torch.stack([T.rank_log_kernel_slices(r, W, own(fa, r), own(fb, r), fa, fb, 500.0) for r in range(W)], 0)
for rank in (0, 5):
    for need_b in (False, True):
        f = lambda: T.rank_matching_stack(rank, W, rows, fa, fb, 500.0, L, need_b, gather=allk)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        print(f"D={D} L={L} rank {rank} need_b={need_b}: {(time.perf_counter()-t0)/10*1e6:.1f} us", flush=True)
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(5): T.rank_matching_stack(0, W, rows, fa, fb, 500.0, L, False, gather=allk)
torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
print({k: round(v["ms"] / 5 * 1e3, 1) for k, v in p.items() if v["launches"]})
