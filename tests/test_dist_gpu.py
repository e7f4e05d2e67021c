"""GPU: two ranks (two processes sharing cuda:0, gloo transport) run one critic and one
generator step with the reference-faithful GLOBAL matching; the all-reduced gradients must equal
those of a single process holding both shards (the reference sums tower gradients,
train.py:134-139, and forms mini-batch 1 / 2 from the first / second half of the shards,
utils/matching.py:16-19).  RCCL replaces gloo on a multi-GPU node; the trainer code path
(feature all-gather -> row-range matching -> gradient SUM all-reduce) is the same."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

LAM, ITERS, B = 100.0, 20, 4
SEED = int(os.environ.get("OTGAN_TEST_DIST_SEED", "5"))        # tools/exp/dist_tolerance.sh sweeps it (measurement, not CI)


def _report(tag, got, ref, names=None):
    """Per-tensor relative error of the all-reduced gradients against the single process, printed (pytest -s): the bound below
    is set from these numbers, not from a guess."""
    errs = []
    for i, (a, b) in enumerate(zip(got, ref)):
        errs.append(float((a - b).norm() / b.norm().clamp_min(1e-30)))
    srt = sorted(errs)
    print(f"[dist-tolerance] seed={SEED} {tag}: tensors={len(errs)} max={srt[-1]:.3e} median={srt[len(srt) // 2]:.3e} "
          f"min={srt[0]:.3e} bitwise_equal={sum(e == 0.0 for e in errs)}", flush=True)
    return errs


# Bound on |multi-rank - single process| / |single process| per gradient tensor.  Both sides are this library in fp32; they differ
# in batch composition per process (operand scales of the two-piece split follow the rank's own amax, weight-gradient K
# splits follow the rank's tile count) and in the matching route (row-sharded global path against the local call).
TOL = 1.5e-4
TOL_FLIPPED = 1e-2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(123)
    x = torch.rand(2 * B, 32, 32, 3, generator=g) * 2 - 1
    u = torch.rand(2 * B, 100, generator=g) * 2 - 1
    return x, u


def _run_steps(model, x, u):
    from otgan_amd import ops
    out = {}
    for kind, ctr in (("disc", 0), ("gen", 1)):
        model.step_counter = ctr
        ops.SIGN_TRACE = []
        try:
            r = model.step(x, noise=u, apply_updates=False)
        finally:
            out[kind + "_signs"], ops.SIGN_TRACE = ops.SIGN_TRACE, None
        assert r["kind"] == kind
        out[kind] = [t.detach().cpu() for t in r["grads"]]
        out[kind + "_dist"] = float(r["distance"])
    return out


def _flips(rank_signs, single_signs, world, blocks):
    """ReLU-type units (CReLU layer inputs, the feature head) whose input has a different sign in the ranks' passes than in the
    single process's pass over the same samples.  rank_signs[r]: rank r's trace (one bool tensor per traced call);
    single_signs: the single process's trace.  blocks: sample blocks per traced call -- 2 in a critic step (one critic pass
    over cat([x_data, x_gen]), train.py:83-85), 1 in a generator step (a pass per batch); a rank's samples are contiguous
    inside each block."""
    n = 0
    for r, trace in enumerate(rank_signs):
        assert len(trace) == len(single_signs) and len(trace) > 0
        for a, b in zip(trace, single_signs):
            assert b.shape[0] == world * a.shape[0] and a.shape[0] % blocks == 0
            per = a.shape[0] // blocks
            for j in range(blocks):
                ref = b[j * world * per + r * per: j * world * per + (r + 1) * per]
                n += int((a[j * per:(j + 1) * per] != ref).sum())
    return n


def _check(tag, got, ref, signs, world):
    """got / ref: _run_steps() of a rank / of the single process; signs: every rank's traces {kind: [trace per rank]}."""
    for kind, blocks in (("disc", 2), ("gen", 1)):
        assert got[kind + "_dist"] == pytest.approx(ref[kind + "_dist"], rel=1e-4, abs=1e-8), (tag, kind)
        flips = _flips(signs[kind], ref[kind + "_signs"], world, blocks)
        errs = _report(f"{tag} {kind} sign_flips={flips}", got[kind], ref[kind])
        # No unit changed sides: the two runs differ by fp32 rounding only (operand scales follow each process's own amax, the
        # weight gradients' K splits its tile count, the matching route its row sharding).  Measured over seeds 5 - 10 x twelve
        # cases (profiles/r06_dist_tolerance_sweep.txt): 66 cases without a flipped unit, at most 4.8e-5 (median tensor 2e-6 .. 8e-6)
        # -> TOL = 3 x that.  The six cases with ONE flipped unit each: 8.3e-5 .. 3.3e-3 -> TOL_FLIPPED = 3 x that.
        assert max(errs) < (TOL if flips == 0 else TOL_FLIPPED), (tag, kind, flips, max(errs))


def _worker(rank, world, port, path, single_batch=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from otgan_amd import parallel
    from otgan_amd.trainer import OTGAN, default_args
    parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    args = default_args(model="dcgan", batch_size=B, nr_gpu=2, sinkhorn_lambda=LAM, nr_sinkhorn_iter=ITERS,
                        nr_gen_per_disc=1, seed=SEED, matching_scope="global", single_batch=single_batch)
    m = OTGAN(args, dev)
    assert m.shards == 1 and m.scope == "global"
    x, u = _data()
    sl = slice(rank * B, (rank + 1) * B)
    res = _run_steps(m, x[sl].to(dev), u[sl].to(dev))
    torch.save(res, path + str(rank))
    parallel.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("single_batch", [False, True], ids=["two_batch", "single_batch"])
def test_two_ranks_equal_single_process(single_batch):
    """single_batch (round 4): the global scope shards the three cost GEMMs over the ranks like the reference's towers
    (utils/matching.py:99-104) and applies the plans to the rank's rows only (otgan_matching_single_batch_rows_grad_f32)."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "r0.pt")
        port = _free_port()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_worker, args=(r, 2, port, path, single_batch)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        got = [torch.load(path + str(r)) for r in range(2)]
    from otgan_amd.trainer import OTGAN, default_args
    dev = torch.device("cuda:0")
    args = default_args(model="dcgan", batch_size=B, nr_gpu=2, sinkhorn_lambda=LAM, nr_sinkhorn_iter=ITERS,
                        nr_gen_per_disc=1, seed=SEED, single_batch=single_batch)
    m = OTGAN(args, dev)          # world 1: both shards local
    x, u = _data()
    ref = _run_steps(m, x.to(dev), u.to(dev))
    # (distance: two evaluations of the same loss -- the sharded path uses the closed form over row slices of separately
    # computed cost blocks; the loss is a cancellation of O(1) terms, so fp32 rounding of the features shows at ~1e-5)
    signs = {kind: [g[kind + "_signs"] for g in got] for kind in ("disc", "gen")}
    for kind in ("disc", "gen"):
        for a, a1 in zip(got[0][kind], got[1][kind]):
            assert torch.equal(a, a1)               # every rank holds the same all-reduced sum
    _check(f"2 ranks single_batch={single_batch}", got[0], ref, signs, 2)


def _ddi_worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from otgan_amd import parallel
    from otgan_amd.trainer import OTGAN, default_args
    parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(11 + rank)            # train.py seeds seed + rank: the generator's init latent differs per rank
    args = default_args(model="dcgan", batch_size=B, nr_gpu=2, sinkhorn_lambda=LAM, nr_sinkhorn_iter=ITERS,
                        nr_gen_per_disc=1, seed=SEED, matching_scope="global", data_dependent_init=True)
    x, _ = _data()
    m = OTGAN(args, dev, init_batch=x[:B])
    sd = {k: v.detach().cpu() for t in (m.discriminator, m.generator) for k, v in t.named_variables().items()}
    sd["__ema__"] = [m.ema.average(p).detach().cpu() for p in m.gen_params]
    torch.save(sd, path + str(rank))
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_data_dependent_init_gives_identical_replicas():
    """with --data_dependent_init every rank ran the init pass on its own latent draw and nothing
    synchronised the result.  Rank 0's initialisation is broadcast before the EMA shadows are cloned."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sd")
        port = _free_port()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_ddi_worker, args=(r, 2, port, path)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        a, b = torch.load(path + "0"), torch.load(path + "1")
    moved = 0
    for k in a:
        if k == "__ema__":
            for u, v in zip(a[k], b[k]):
                assert torch.equal(u, v)
            continue
        assert torch.equal(a[k], b[k]), k
        if k.endswith("/g") and "generator" in k:
            moved += int(not torch.all(a[k] == 1.0))
    assert moved > 0                        # the pass really ran (g left its default of 1)


# ---- eight ranks of the REAL trainer on one device -------------------------------------------------
# The reference drives all towers from one process (train.py:72-85,134-139) and puts shards [n/2, n) into mini-batch 2
# (utils/matching.py:16-19,35-39).  With eight ranks, ranks 4 - 7 own rows of the SECOND half: their cost slices are
# (b2,b1) (a2,b1) (a2,b2) (trainer.rank_log_kernel_slices, r >= world / 2) and their plan rows sit behind the first
# half's -- code that two ranks (one per half) never reach with more than one rank per half.  All four combinations of
# {two shards per rank, one shard per rank} x {two-batch, --single_batch} run inside ONE set of eight processes.
EIGHT_CASES = [(16, False), (8, False), (16, True), (8, True)]      # (--nr_gpu, --single_batch)
B8 = 2


def _data8(nr_gpu):
    g = torch.Generator().manual_seed(321 + nr_gpu)
    n = nr_gpu * B8
    return torch.rand(n, 32, 32, 3, generator=g) * 2 - 1, torch.rand(n, 100, generator=g) * 2 - 1


def _args8(nr_gpu, single_batch):
    from otgan_amd.trainer import default_args
    return default_args(model="dcgan", batch_size=B8, nr_gpu=nr_gpu, sinkhorn_lambda=LAM, nr_sinkhorn_iter=ITERS,
                        nr_gen_per_disc=1, seed=SEED, matching_scope="global", single_batch=single_batch)


def _worker8(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from otgan_amd import parallel
    from otgan_amd.trainer import OTGAN
    parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    out = {}
    for nr_gpu, single in EIGHT_CASES:
        m = OTGAN(_args8(nr_gpu, single), dev)
        assert m.shards == nr_gpu // world and m.scope == "global" and m.world == world
        x, u = _data8(nr_gpu)
        sl = slice(rank * m.nb, (rank + 1) * m.nb)
        out[(nr_gpu, single)] = _run_steps(m, x[sl].to(dev), u[sl].to(dev))
        m.close()
    if rank not in (0, world - 1):      # the all-reduced gradients of a first-half and of a second-half rank; everyone's sign traces
        out = {c: {k: v for k, v in o.items() if k.endswith("_signs")} for c, o in out.items()}
    torch.save(out, path + str(rank))
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_eight_ranks_equal_single_process():
    assert torch.cuda.is_available()
    world = 8
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "r")
        port = _free_port()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_worker8, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
            assert p.exitcode == 0
        allr = [torch.load(path + str(r)) for r in range(world)]
        got0, got7 = allr[0], allr[world - 1]
    from otgan_amd.trainer import OTGAN
    dev = torch.device("cuda:0")
    for nr_gpu, single in EIGHT_CASES:
        m = OTGAN(_args8(nr_gpu, single), dev)      # world 1: all shards local, halves = shards [0, S/2) | [S/2, S)
        x, u = _data8(nr_gpu)
        ref = _run_steps(m, x.to(dev), u.to(dev))
        m.close()
        g0, g7 = got0[(nr_gpu, single)], got7[(nr_gpu, single)]
        for kind in ("disc", "gen"):
            assert g7[kind + "_dist"] == g0[kind + "_dist"]
            for a, a7 in zip(g0[kind], g7[kind]):
                assert torch.equal(a, a7)               # every rank holds the same all-reduced sum
        signs = {kind: [o[(nr_gpu, single)][kind + "_signs"] for o in allr] for kind in ("disc", "gen")}
        _check(f"8 ranks nr_gpu={nr_gpu} single_batch={single}", g0, ref, signs, world)
