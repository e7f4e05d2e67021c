"""dev: the first steps of a DCGAN / DenseNet training run, distance and entropy per step (compare sweep regimes:
OTGAN_SINKHORN_LINEAR=0/1).   python tools/debug/step_trace.py [model] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from otgan_amd import _lib  # noqa: E402
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else "dcgan"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
_lib.lib()
args = default_args(model=model_name, batch_size=128, nr_gpu=2, nr_sinkhorn_iter=100, sinkhorn_lambda=500.0, nr_gen_per_disc=5,
                    matching_scope="global", seed=1, image_size=32)
model = OTGAN(args, dev)
torch.manual_seed(5)            # the generator's noise comes from the global CUDA generator
g = torch.Generator().manual_seed(7)
for i in range(steps):
    x = (torch.rand(model.nb, 32, 32, 3, generator=g) * 2 - 1).to(dev)
    out = model.step(x)
    print(f"step {i:3d} {out['kind']:4s} distance {float(out['distance']):.9f} entropy {float(out['entropy']):.7f}", flush=True)
