"""Child process of tests/test_rccl_gpu.py (not a test module): RCCL at world size 1 with the collectives forced."""
import os, sys
sys.path.insert(0, os.environ["OTGAN_ROOT"])
import torch
import torch.distributed as dist
from otgan_amd import parallel
from otgan_amd.trainer import OTGAN, default_args

rank, world, local = parallel.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1, (dist.is_initialized(), world)
assert not parallel._skip_collectives()
dev = torch.device("cuda", 0)
torch.manual_seed(0)

# ---- plain collectives
x = torch.randn(37, 129, device=dev)
assert torch.equal(parallel.all_gather_rows(x), x)
y = x * 2                                  # produced on the compute stream right before the async gather
pend = parallel.all_gather_rows_async(y)
z = torch.randn(512, 512, device=dev) @ torch.randn(512, 512, device=dev)   # compute enqueued while it runs
got = pend.wait()
assert got.data_ptr() != y.data_ptr() and torch.equal(got, x * 2)
ts = [torch.randn(5, 3, device=dev), torch.randn(7, device=dev)]
keep = [t.clone() for t in ts]
red = parallel.allreduce_sum_(ts)
assert all(torch.equal(a, b) for a, b in zip(red, keep))

# ---- bucket hooks under torch.autograd.grad
ps = [torch.randn(64, 64, device=dev, requires_grad=True) for _ in range(6)]
gb = parallel.GradBuckets(ps, nbuckets=3)
inp = torch.randn(8, 64, device=dev)
def loss():
    h = inp
    for p in ps:
        h = torch.tanh(h @ p)
    return h.square().sum()
plain = torch.autograd.grad(loss(), ps)           # not armed: hooks are inert
gb.arm()
torch.autograd.grad(loss(), ps)
views = gb.finish()
assert len(gb.ranges) == 3
assert all(torch.equal(a, b) for a, b in zip(views, plain))
gb.remove()

# ---- the whole training step: collectives forced vs not forced, same seeds -> identical gradients
def run(force):
    os.environ["OTGAN_FORCE_COLLECTIVES"] = "1" if force else "0"
    args = default_args(model="dcgan", batch_size=4, nr_gpu=2, sinkhorn_lambda=100.0, nr_sinkhorn_iter=10,
                        nr_gen_per_disc=1, seed=5)
    m = OTGAN(args, dev)
    # (round 6: the two-stream schedule runs in BOTH exchange schedules; serial all-reduces one flat buffer after the backward
    # pass, overlapped sends gradient buckets from the side stream inside it and gathers the real features under the generator)
    assert m.collectives == force and (m.gen_buckets is not None) == (force and parallel.collectives_mode() == "overlapped")
    assert m.fork_wgrad and m.fork_real_pass
    assert m.collectives_mode == (parallel.collectives_mode() if force else "none")
    g = torch.Generator().manual_seed(3)
    xd = (torch.rand(m.nb, 32, 32, 3, generator=g) * 2 - 1).to(dev)
    u = (torch.rand(m.nb, 100, generator=g) * 2 - 1).to(dev)
    out = []
    for ctr in (0, 1):
        m.step_counter = ctr
        r = m.step(xd, noise=u, apply_updates=False)
        out.append(([t.clone() for t in r["grads"]], float(r["distance"])))
    for _ in range(3):                      # and with updates applied (optimiser consumes bucket views)
        r = m.step(xd, noise=u)
    out.append(([p.detach().clone() for p in m.gen_params + m.disc_params], float(r["distance"])))
    m.close()
    return out

os.environ["OTGAN_COLLECTIVES"] = "serial"               # pinned: no collective kernel beside the step's kernels
assert parallel.collectives_mode() == "serial"
a, b = run(True), run(False)
os.environ["OTGAN_COLLECTIVES"] = "overlapped"           # pinned: buckets inside the backward pass, gather under the generator
assert parallel.collectives_mode() == "overlapped"
c = run(True)
# the guarded default: the trainer's start-up self-check (RCCL collectives co-resident with the 256 x 128 GEMM, bit-compared)
# passes on a healthy device and selects the overlapped schedule; a forced failure falls back to serial and says so
os.environ["OTGAN_COLLECTIVES"] = "auto"
parallel._MODE["mode"] = None
os.environ["OTGAN_FORCE_COLLECTIVES"] = "1"
ok, why = parallel.overlap_self_check(dev)
assert ok and "bit-identical" in why, why
d = run(True)
assert parallel.collectives_mode() == "overlapped" and "self-check passed" in parallel.collectives_mode_reason()
real_check = parallel.overlap_self_check
parallel.overlap_self_check = lambda device, rounds=3: (False, "self-check FAILED (injected by the test): collectives stay serial")
parallel._MODE["mode"] = None
e = run(True)
assert parallel.collectives_mode() == "serial" and "FAILED" in parallel.collectives_mode_reason()
parallel.overlap_self_check = real_check
parallel._MODE["mode"] = None
for other in (a, c, d, e):
    for (ga, da), (gb_, db) in zip(other, b):
        assert abs(da - db) <= 1e-12 * abs(db), (da, db)     # fp64 atomics in the distance reduction: order-dependent last bits
        assert all(torch.equal(s, t) for s, t in zip(ga, gb_))
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WS1_OK")
