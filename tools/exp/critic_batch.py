"""Experiment (round 5): the critic forward pass on 512 images in one go against two passes of 256 (the generator step
runs the critic once on the real batch without a graph and once on the generated batch): shared Winograd-domain filter
operands are read once, and the 4x4 / 8x8 stages get twice the tiles per launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    model = sys.argv[1] if len(sys.argv) > 1 else "dcgan"
    args = default_args(model=model, batch_size=128, nr_gpu=2, nr_sinkhorn_iter=100, seed=1)
    m = OTGAN(args, dev)
    x = torch.rand(256, 32, 32, 3, device=dev) * 2 - 1
    xg = torch.rand(256, 32, 32, 3, device=dev) * 2 - 1
    xc = torch.cat([x, xg], 0)
    D = lambda t: m.discriminator(t, **m.model_opts)
    with torch.no_grad():
        t2 = timeit(lambda: (D(x), D(xg)))
        t1 = timeit(lambda: D(xc))
    print(f"{model}: critic forward 2 x 256: {t2:.3f} ms   1 x 512: {t1:.3f} ms   gain {t2 - t1:.3f} ms")


if __name__ == "__main__":
    main()
