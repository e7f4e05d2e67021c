"""One-process-per-GPU data parallelism for the OT-GAN step (RCCL over xGMI via
torch.distributed backend "nccl"; "gloo" on CPU for tests).

The reference is a single process with in-graph towers (`tf.device('/gpu:%d')`,
train.py:72-85) whose cross-device traffic is implicit.  Here each rank owns
`shards_per_rank` of the reference's `nr_gpu` logical shards and the two exchange steps are
explicit collectives:

  * feature all-gather  (replaces the implicit gather of utils/matching.py:16-19) -- only in
    the reference-faithful *global* matching scope;
  * gradient all-reduce(SUM)  (replaces the sum-to-gpu:0 of train.py:134-139; the reference
    sums, it does not average).

Everything in this file is backend-agnostic tensor plumbing (no HIP kernels), so it is
covered by world_size-2 gloo tests on CPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns
    (rank, world, local_rank).  Single-process runs skip initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if single_device_mode():      # every rank on cuda:0, gloo transport (RCCL refuses two ranks on one device)
        local = 0
        os.environ.setdefault("OTGAN_DIST_BACKEND", "gloo")
    force = os.environ.get("OTGAN_FORCE_COLLECTIVES") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # OTGAN_DIST_BACKEND=gloo lets several ranks share one GPU (tests of the multi-rank
            # logic on a single-GPU box; RCCL refuses two ranks on one device)
            backend = os.environ.get("OTGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class LaunchError(RuntimeError):
    """A multi-rank run was asked for that this node cannot start."""


def launched():
    """True inside a rank started by torchrun / torch.distributed.run (or by `self_launch`)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def single_device_mode():
    """OTGAN_SINGLE_DEVICE=1 (tests of the multi-rank logic on a 1-GPU box): every rank uses cuda:0 and the
    transport is gloo, because RCCL refuses two ranks on one device."""
    return os.environ.get("OTGAN_SINGLE_DEVICE", "") not in ("", "0")


def check_devices(nranks):
    """Raise LaunchError unless `nranks` one-GPU ranks can run on this node."""
    if not torch.cuda.is_available():
        raise LaunchError("no MI355X visible (torch.cuda.is_available() is False): there is no CPU fallback")
    have = torch.cuda.device_count()
    if nranks > have and not single_device_mode():
        raise LaunchError(f"{nranks} ranks requested but this node exposes {have} GPU(s); one process per GPU is the "
                          f"only layout (set OTGAN_SINGLE_DEVICE=1 to run all ranks on cuda:0 over gloo -- a logic "
                          f"test, not a measurement)")


def self_launch(script, argv, nranks):
    """The reference drives every device from ONE command (`for i in range(args.nr_gpu): with
    tf.device('/gpu:%d' % i)`, train.py:72-85).  Here a rank is a process: when `script` is started plainly
    (no WORLD_SIZE in the environment) and more than one rank is wanted, re-execute it under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node nranks` on the loopback address and return the
    children's exit code.  The caller exits with it."""
    import subprocess
    import sys
    check_devices(nranks)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    if single_device_mode():
        env.setdefault("OTGAN_DIST_BACKEND", "gloo")
    # --standalone: torchrun's own rendezvous picks a free port on the loopback address (no bind-then-close race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nranks}", script] + list(argv)
    rc = subprocess.call(cmd, env=env)
    if rc == 2:
        # argparse's exit code: a torch.distributed.run without --local-addr (before torch 2.2).  The same rendezvous spelled
        # out -- c10d store on a free loopback port (what --standalone sets up) -- without the flag
        cmd = [sys.executable, "-m", "torch.distributed.run", "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0",
               f"--rdzv-id=otgan-{os.getpid()}", "--nnodes=1", f"--nproc-per-node={nranks}", script] + list(argv)
        rc = subprocess.call(cmd, env=env)
    return rc


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def _skip_collectives():
    """World size 1 needs no exchange.  OTGAN_FORCE_COLLECTIVES=1 (tests, bench.py --force_collectives) runs the
    collective code path anyway on an initialised 1-rank group, so that the RCCL calls, their stream ordering and
    the bucket hooks are exercised on a single-GPU box."""
    if world_size() > 1:
        return False
    return not (dist.is_initialized() and os.environ.get("OTGAN_FORCE_COLLECTIVES") == "1")


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


# ---- may a collective run BESIDE the step's own kernels? ------------------------------------------------------------------
# Overlapped: the gradient buckets leave inside the backward pass and the real features' all-gather runs under the generator's
# forward pass (RCCL's stream next to the compute streams).  Serial: every collective is enqueued behind the compute that
# precedes it and the compute stream waits for it before it continues, so no collective kernel is ever co-resident with a
# kernel of this library.  DESIGN "Four hazards", item 3: kernels co-resident with the 256 x 128 GEMM were once seen computing
# wrong values (root-caused to one packed-fp32 instruction form that the library no longer contains), and RCCL's own kernels
# are not ours to recompile.  So the mode is DECIDED BY A MEASUREMENT at start-up (round 6): `resolve_collectives_mode` runs
# RCCL collectives co-resident with the 256 x 128 GEMM, compares both sides bit for bit with their serial results, and picks
# "overlapped" only when everything matches on every rank; anything else (a mismatch, an exception, gloo) is "serial".
# OTGAN_COLLECTIVES=serial|overlapped pins the mode (no self-check); "auto" / unset = the guarded default.
_MODE = {"mode": None, "why": "not resolved yet (serial until parallel.resolve_collectives_mode runs)"}


def collectives_mode():
    """'serial' or 'overlapped' (see above).  Before `resolve_collectives_mode` has run, 'auto' reads as 'serial'."""
    v = os.environ.get("OTGAN_COLLECTIVES", "auto").strip().lower()
    if v in ("serial", "overlapped"):
        return v
    return _MODE["mode"] or "serial"


def collectives_mode_reason():
    v = os.environ.get("OTGAN_COLLECTIVES", "auto").strip().lower()
    if v in ("serial", "overlapped"):
        return f"pinned by OTGAN_COLLECTIVES={v}"
    return _MODE["why"]


def resolve_collectives_mode(device=None, force=False):
    """The guarded default: decide once per process (all ranks together) whether collectives may overlap compute.
    Returns the mode.  No-op when the mode is pinned, already resolved, or no exchange will happen."""
    v = os.environ.get("OTGAN_COLLECTIVES", "auto").strip().lower()
    if v in ("serial", "overlapped"):
        return v
    if _MODE["mode"] is not None and not force:
        return _MODE["mode"]
    if _skip_collectives():
        _MODE.update(mode="serial", why="no exchange (one rank, collectives not forced)")
    elif _staged():
        _MODE.update(mode="serial", why="gloo stages through host memory and is synchronous")
    elif device is None or not torch.cuda.is_available():
        _MODE.update(mode="serial", why="no device for the co-residency self-check")
    else:
        ok, why = overlap_self_check(device)
        _MODE.update(mode="overlapped" if ok else "serial", why=why)
    return _MODE["mode"]


def overlap_self_check(device, rounds=3):
    """RCCL collectives co-resident with the 256 x 128 split-precision GEMM, both checked bit for bit.

    Per round: a SUM all-reduce of 64 MB and an all-gather of 8 MB per rank are started asynchronously (RCCL's stream) and
    eight forward passes of a G.conv1-shaped layer (input transform, 256 x 128-tile GEMM, output transform: the kernels of a
    training step) are enqueued on the compute stream while they run.  The layer's outputs must equal the output computed
    with nothing else on the device; the collectives' results must equal their closed forms (rank r contributes
    (r + 1) x small integers: the sum and every gathered row are exact in fp32).  The verdict is the minimum over ranks.
    -> (ok, reason)."""
    w, r = world_size(), get_rank()
    ok, err = True, None
    try:
        from . import ops
        gen = torch.Generator(device=device).manual_seed(4321)
        B, H, C, Cout = 128, 8, 512, 512
        x = torch.randn(B, H, H, C, device=device, generator=gen)
        V = torch.randn(5, 5, C, Cout, device=device, generator=gen) * 0.05
        g = torch.ones(Cout, device=device)
        b = torch.zeros(Cout, device=device)
        conv = lambda: ops.conv2d_op(x, V, g, b, stride=1, upsample=True, preact=0)
        n_red, n_gat = 1 << 24, 1 << 21
        base = (torch.arange(n_red, device=device) % 1021).float()
        with torch.no_grad():
            y_ref = conv().clone()
            torch.cuda.synchronize(device)
            for _ in range(rounds):
                buf = base * float(r + 1)
                mine = buf[:n_gat].clone()
                out = torch.empty(w * n_gat, device=device)
                w1 = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
                ys = [conv() for _ in range(4)]
                w2 = dist.all_gather_into_tensor(out, mine, async_op=True)
                ys += [conv() for _ in range(4)]
                w1.wait()
                w2.wait()
                ok = ok and all(torch.equal(y, y_ref) for y in ys)
                ok = ok and torch.equal(buf, base * float(w * (w + 1) // 2))
                want = base[:n_gat][None, :] * torch.arange(1, w + 1, device=device, dtype=torch.float32)[:, None]
                ok = ok and torch.equal(out.view(w, n_gat), want)
    except Exception as e:      # noqa: BLE001 -- never lose a run over the optimisation
        ok, err = False, f"{type(e).__name__}: {e}"
    # every rank reaches the vote, whatever happened above (a rank that raised must not leave the others waiting in it)
    try:
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize(device)
        agreed = float(flag) == 1.0
    except Exception as e:      # noqa: BLE001
        agreed, err = False, err or f"{type(e).__name__}: {e}"
    if agreed:
        return True, (f"self-check passed on {w} rank(s): {rounds} x (64 MB all-reduce + all-gather) co-resident with 8 GEMM "
                      "layers, all results bit-identical to the serial ones")
    if err:
        return False, f"self-check raised {err}: collectives stay serial"
    return False, "self-check FAILED (" + ("this rank" if not ok else "another rank") + " saw a mismatch): collectives stay serial"


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def _staged():
    """gloo has no CUDA all-gather: stage through host memory (tests only; RCCL is direct)."""
    return dist.get_backend() == "gloo"


def all_gather_rows(x):
    """[n, D] per rank -> [world*n, D] in rank order (one collective)."""
    w = world_size()
    if _skip_collectives():
        return x
    x = x.contiguous()
    if _staged() and x.is_cuda:
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(w)]
        dist.all_gather(parts, x.cpu())
        return torch.cat(parts, 0).to(x.device)
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out


class PendingGather:
    """An all-gather in flight (RCCL runs it on its own stream, so it overlaps the compute that
    is enqueued after the call); `.wait()` returns the gathered [world*n, D] tensor."""

    def __init__(self, out, work):
        self.out, self.work = out, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.out.is_cuda:      # (the gather may have been started from another stream's context: used HERE from now on)
                self.out.record_stream(torch.cuda.current_stream())
        return self.out


def all_gather_rows_async(x):
    w = world_size()
    if _skip_collectives():
        return PendingGather(x, None)
    x = x.contiguous()
    if _staged() and x.is_cuda:
        return PendingGather(all_gather_rows(x), None)
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    work = dist.all_gather_into_tensor(out, x, async_op=True)
    if collectives_mode() == "serial":
        work.wait()      # the compute stream waits here: nothing enqueued after this call runs beside the gather
        work = None
    return PendingGather(out, work)


def gather_feature_shards(f_local, shards_per_rank):
    """Local features [shards_per_rank*B, D] -> the reference's global shard list of
    `world*shards_per_rank` tensors [B, D]; shard s lives on rank s // shards_per_rank, so the
    first half of the ranks forms mini-batch 1 and the second half mini-batch 2
    (utils/matching.py:16-19)."""
    allf = all_gather_rows(f_local)
    S = world_size() * shards_per_rank
    return list(torch.chunk(allf, S, 0))


def local_rows(flat_global, rows_per_rank):
    """This rank's rows of a globally ordered [world*rows_per_rank, D] tensor."""
    r = get_rank()
    return flat_global[r * rows_per_rank:(r + 1) * rows_per_rank]


def allreduce_sum_(tensors):
    """SUM all-reduce of a list of tensors through ONE flat bucket; returns the reduced tensors (a single large
    collective suits xGMI's point-to-point links better than many small ones)."""
    if _skip_collectives() or not tensors:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if _staged() and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    out, off = [], 0
    for t in tensors:       # hand back views of the reduced bucket (no per-tensor copies)
        n = t.numel()
        out.append(flat[off:off + n].view_as(t))
        off += n
    return out


class GradBuckets:
    """Gradient SUM all-reduce (train.py:134-139) overlapped with the backward pass.

    The variables are laid out, in REVERSE creation order -- the order in which backward produces
    their gradients -- in one flat buffer cut into `nbuckets` contiguous slices of similar size.
    A tensor hook copies each gradient into its slice as soon as autograd has it; when the last
    gradient of a slice has arrived its all-reduce is launched asynchronously (RCCL runs it on its
    own stream after the producing kernels, so it overlaps the rest of the backward pass: over a
    single xGMI link -- two GPUs -- the 140 MB of a DCGAN network take longer than a quarter of
    the backward).  `finish()` waits for the collectives and returns the reduced gradients as
    views of the flat buffer, in the order of `params`.  Hooks only act between `arm()` and
    `finish()`, so variables that are evaluated without being differentiated are unaffected."""

    _live = {}      # id(first param) -> the instance whose hooks are installed on that parameter set

    def __init__(self, params, nbuckets=4):
        self.params = list(params)
        n = len(self.params)
        # a second trainer over the same (module-level, shared) variables replaces the first one's
        # hooks instead of stacking a second set on top
        old = GradBuckets._live.get(id(self.params[0]))
        if old is not None:
            old.remove()
        GradBuckets._live[id(self.params[0])] = self
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=self.params[0].dtype, device=self.params[0].device)
        self.views = [None] * n
        self.bucket_of = [0] * n
        self.ranges = []                  # [lo, hi) of each bucket in the flat buffer
        self.count = []                   # gradients per bucket
        target = max(1, -(-total // max(1, nbuckets)))
        off, lo, cnt = 0, 0, 0
        for i in reversed(range(n)):
            p = self.params[i]
            self.views[i] = self.flat[off:off + p.numel()].view(p.shape)
            self.bucket_of[i] = len(self.ranges)
            off += p.numel()
            cnt += 1
            if off - lo >= target or i == 0:
                self.ranges.append((lo, off))
                self.count.append(cnt)
                lo, cnt = off, 0
        self.armed = False
        self.left, self.works, self.deferred = [], [], []
        self.handles = [p.register_hook(lambda g, i=i: self._on_grad(i, g)) for i, p in enumerate(self.params)]

    def remove(self):
        """Uninstall the tensor hooks (the instance is dead afterwards)."""
        for h in self.handles:
            h.remove()
        self.handles = []
        self.armed = False
        if GradBuckets._live.get(id(self.params[0])) is self:
            del GradBuckets._live[id(self.params[0])]

    def arm(self):
        self.armed = True
        self.left = list(self.count)
        self.works = []
        self.deferred = []

    def _on_grad(self, i, g):
        if not self.armed:
            return None
        # With the layers' weight-gradient chains on a second stream (ops.SIDE_STREAM) a gradient handed to this hook may
        # still be in flight THERE while the hook runs in the main stream's context.  The copy into the bucket and the
        # bucket's all-reduce are therefore issued on the side stream, which first waits for the main stream's position
        # (a gradient produced on the main stream -- a dense layer's, a bias' -- is covered by that; the weight-gradient
        # chain of the NEXT layer depends on the main stream's input-gradient kernels up to here anyway, so the wait costs
        # nothing).  The main stream is never made to wait for the side stream inside the backward pass.
        side = None
        if g.is_cuda:
            from . import ops
            side = ops.SIDE_STREAM
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            g.record_stream(side)
            ctx = torch.cuda.stream(side)
        else:
            ctx = _NO_CTX
        with ctx:
            self.views[i].copy_(g)
            b = self.bucket_of[i]
            self.left[b] -= 1
            if self.left[b] == 0:
                lo, hi = self.ranges[b]
                seg = self.flat[lo:hi]
                if _staged() and seg.is_cuda:      # gloo on device tensors (tests): through host memory
                    host = seg.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    seg.copy_(host)
                elif collectives_mode() == "serial":
                    # no collective kernel beside the backward pass's kernels (collectives_mode): all buckets go out after
                    # the last gradient, in finish()
                    self.deferred.append(seg)
                else:
                    # (the process group orders the collective behind the CURRENT stream: the side stream when there is one)
                    self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        return None

    def finish(self):
        if any(self.left):
            self.armed = False
            raise RuntimeError("GradBuckets.finish(): some variables received no gradient in this backward pass")
        if self.deferred:
            # the process group orders a collective behind everything already enqueued on the current stream (the whole
            # backward pass by now) and wait() below makes the stream wait for it: no host synchronisation needed
            for seg in self.deferred:
                self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
            self.deferred = []
        for w in self.works:
            w.wait()
        self.works = []
        self.armed = False
        return list(self.views)


def broadcast_(t, src=0):
    """In-place broadcast of rank `src`'s tensor (replica synchronisation after a data-dependent pass)."""
    if world_size() == 1:
        return t
    if _staged() and t.is_cuda:
        host = t.detach().cpu()
        dist.broadcast(host, src=src)
        t.detach().copy_(host)
    else:
        dist.broadcast(t.detach(), src=src)
    return t


def barrier():
    if dist.is_initialized():
        dist.barrier()
