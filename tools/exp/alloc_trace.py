#!/usr/bin/env python3
"""How long until torch's caching allocator stops calling hipMalloc in a training loop?  (dev tool, GPU box)
   usage: tools/exp/alloc_trace.py [dcgan|densenet] [periods]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from otgan_amd.trainer import OTGAN, default_args
model = sys.argv[1] if len(sys.argv) > 1 else "dcgan"
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
args = default_args(batch_size=128, nr_gpu=2, sinkhorn_lambda=500.0, nr_gen_per_disc=5, seed=1, model=model,
                    nr_sinkhorn_iter=100 if model == "dcgan" else 200, matching_scope=os.environ.get("SCOPE", "global"), image_size=32)
m = OTGAN(args, dev)
x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
last = 0
for p in range(periods):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        m.step(x)
    if os.environ.get("NOSYNC") == "1" and p % 25 != 24:
        continue                      # (an un-synchronised loop: the host is only held back by trainer._throttle)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6 * 1e3
    st = torch.cuda.memory_stats(dev)
    n = st["num_device_alloc"]
    print(f"period {p:3d}: {dt:7.3f} ms/step  device allocs so far {n:5d} (+{n - last:3d})  reserved {st['reserved_bytes.all.current'] / 2**30:6.2f} GiB "
          f"active {st['active_bytes.all.current'] / 2**30:5.2f} GiB", flush=True)
    last = n
