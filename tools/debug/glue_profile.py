"""dev: which Python lines launch the torch (aten) kernels of a train step -- the glue between the library calls.
    python tools/debug/glue_profile.py [dcgan|densenet]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from otgan_amd import _lib  # noqa: E402
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else "densenet"
dev = torch.device("cuda:0")
_lib.lib()
args = default_args(model=model_name, batch_size=128, nr_gpu=2, nr_sinkhorn_iter=100, sinkhorn_lambda=500.0, nr_gen_per_disc=5,
                    matching_scope="global", seed=1, image_size=32)
model = OTGAN(args, dev)
x = torch.rand(model.nb, 32, 32, 3, device=dev) * 2 - 1
for _ in range(6):
    model.step(x)
torch.cuda.synchronize()
model.step_counter = 0
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(6):
        model.step(x)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = e.cuda_time_total
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    stack = [s for s in e.stack if ("otgan" in s or "ot-gan" in s) and "glue_profile" not in s]
    rows.append((dt / 6.0, e.count / 6.0, e.key, stack[:3] if stack else list(e.stack)[:3]))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"{model_name}: aten ops with device time: {tot:.0f} us per step, {sum(r[1] for r in rows):.0f} calls per step")
for dt, n, key, stack in rows[:45]:
    print(f"{dt:8.1f} us/step {n:6.1f} calls  {key:28s} " + " <- ".join(s.split('/')[-1][:60] for s in stack))
