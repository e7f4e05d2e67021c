"""Repro (round 5): a second trainer after one that replayed step graphs -> 'invalid resource handle'?"""
import os, sys, gc, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from otgan_amd.trainer import OTGAN, default_args
from otgan_amd import ops

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"


def run(model, steps):
    args = default_args(model=model, batch_size=8, nr_gpu=2, nr_sinkhorn_iter=20, nr_gen_per_disc=2, seed=1)
    m = OTGAN(args, dev)
    x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
    for _ in range(steps):
        m.step(x)
    torch.cuda.synchronize()
    print(model, "steps ok; captured", sorted(m.graphs.graphs) if m.graphs else None, flush=True)
    return m


m = run("dcgan", 12)
keep = None
if mode == "keep":
    keep = dict(m.graphs.graphs)
if mode == "reset":
    for g, _, _ in m.graphs.graphs.values():
        g.reset()
m.close()
print("closed", flush=True)
del m
gc.collect()
if mode != "nocache":
    torch.cuda.empty_cache()
print("emptied", flush=True)
try:
    z = torch.zeros(2, 32, 32, 3, device=dev)
    torch.cuda.synchronize()
    print("zeros ok", flush=True)
    z = torch.rand(4, device=dev)
    torch.cuda.synchronize()
    print("rand ok", flush=True)
    m2 = run("dcgan", 12)
    m2.close()
    print("second trainer ok", flush=True)
except Exception:
    traceback.print_exc()
