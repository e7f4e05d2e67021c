"""dev: the two-rank step of tests/test_dist_gpu.py repeated K times with every conv2d output AND every conv backward
result of rank 0 recorded: which tensor is the first to differ between two runs?"""
import os
import sys
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as T  # noqa: E402


def worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from otgan_amd import ops, parallel
    from otgan_amd.trainer import OTGAN, default_args
    parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    rec = []
    fwd0, bwd0 = ops.Conv2dFunction.forward, ops.Conv2dFunction.backward

    def fwd(ctx, x, V, g, b, *a):
        y = fwd0(ctx, x, V, g, b, *a)
        rec.append((f"fwd {tuple(x.shape)}->{tuple(y.shape)}", y.detach().cpu().clone()))
        return y

    def bwd(ctx, dy):
        out = bwd0(ctx, dy)
        for tag, t in zip(("dx", "dV", "dg", "db"), out[:4]):
            if t is not None:
                rec.append((f"bwd {tag} dy{tuple(dy.shape)}", t.detach().cpu().clone()))
        return out

    ops.Conv2dFunction.forward = staticmethod(fwd)
    ops.Conv2dFunction.backward = staticmethod(bwd)
    args = default_args(model="dcgan", batch_size=T.B, nr_gpu=2, sinkhorn_lambda=T.LAM, nr_sinkhorn_iter=T.ITERS,
                        nr_gen_per_disc=1, seed=5, matching_scope="global")
    m = OTGAN(args, dev)
    x, u = T._data()
    sl = slice(rank * T.B, (rank + 1) * T.B)
    rec.clear()
    T._run_steps(m, x[sl].to(dev), u[sl].to(dev))
    if rank == 0:
        torch.save(rec, path)
    parallel.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = mp.get_context("spawn")
    first = None
    for k in range(K):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r0.pt")
            port = T._free_port()
            procs = [ctx.Process(target=worker, args=(r, 2, port, path)) for r in range(2)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(600)
            got = torch.load(path)
        if first is None:
            first = got
            print("run 0:", len(got), "tensors", flush=True)
            continue
        bad = [(i, n) for i, ((n, a), (_, b)) in enumerate(zip(got, first)) if not torch.equal(a, b)]
        if not bad:
            print("run", k, "identical", flush=True)
            continue
        i, n = bad[0]
        a, b = got[i][1], first[i][1]
        d = (a != b)
        idx = d.nonzero()
        print(f"run {k}: {len(bad)} tensors differ; first #{i} {n}: {int(d.sum())} of {d.numel()} elements, rel "
              f"{float((a - b).norm() / b.norm()):.3g}; index range {idx.min(0).values.tolist()} .. {idx.max(0).values.tolist()}", flush=True)
        print("      all differing:", [j for j, _ in bad][:30], flush=True)
