"""dev: is the tiny two-rank test's loss reproducible?  Single-process evaluation of tests/test_dist_gpu.py's steps,
repeated in this process (fresh trainers) -- prints the critic / generator step distances and gradient checksums."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from otgan_amd.trainer import OTGAN, default_args  # noqa: E402
import test_dist_gpu as T  # noqa: E402

dev = torch.device("cuda:0")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    args = default_args(model="dcgan", batch_size=T.B, nr_gpu=2, sinkhorn_lambda=T.LAM, nr_sinkhorn_iter=T.ITERS,
                        nr_gen_per_disc=1, seed=5)
    m = OTGAN(args, dev)
    x, u = T._data()
    res = T._run_steps(m, x.to(dev), u.to(dev))
    print(rep, repr(res["disc_dist"]), repr(res["gen_dist"]),
          [float(g.double().sum()) for g in res["disc"][:2]], flush=True)
    m.close()
