"""dev: the two-rank step of tests/test_dist_gpu.py repeated K times (fresh worker processes each time, two processes
sharing the one GPU over gloo): are its distances / gradients reproducible run to run?"""
import os
import sys
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as T  # noqa: E402

if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = mp.get_context("spawn")
    first = None
    for k in range(K):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r0.pt")
            port = T._free_port()
            procs = [ctx.Process(target=T._worker, args=(r, 2, port, path)) for r in range(2)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(600)
            got = torch.load(path)
        sig = (got["disc_dist"], got["gen_dist"], [float(g.double().sum()) for g in got["disc"][:3]])
        print(k, repr(sig), flush=True)
        if first is None:
            first = got
        else:
            for kind in ("disc", "gen"):
                for i, (a, b) in enumerate(zip(got[kind], first[kind])):
                    if not torch.equal(a, b):
                        print("   differs from run 0:", kind, i, float((a - b).norm() / b.norm()), flush=True)
                        break
