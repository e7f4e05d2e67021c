"""CPU, world_size 2 over gloo: the multi-rank plumbing of the training step (feature
all-gather into the reference's shard order, gradient SUM all-reduce, rank-local row
selection), checked against the single-process oracle matching."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import matching_np as M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from otgan_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    B, D, shards = 6, 20, 1
    rng = np.random.RandomState(100)                      # same global data on every rank
    fa = np.abs(rng.randn(world * shards * B, D)); fa /= np.linalg.norm(fa, axis=1, keepdims=True)
    fb = np.abs(rng.randn(world * shards * B, D) + 0.5); fb /= np.linalg.norm(fb, axis=1, keepdims=True)
    mine = slice(rank * shards * B, (rank + 1) * shards * B)
    la, lb = torch.tensor(fa[mine]), torch.tensor(fb[mine])
    # 1. feature gather reproduces the reference's global shard list (matching.py:16-19)
    ga = parallel.gather_feature_shards(la, shards)
    gb = parallel.gather_feature_shards(lb, shards)
    assert len(ga) == world * shards
    np.testing.assert_array_equal(torch.cat(ga).numpy(), fa)
    # 2. global matching on the gathered shards == single-process oracle; local rows picked
    out_ref = M.get_matched_features([fa[i * B:(i + 1) * B] for i in range(world)],
                                     [fb[i * B:(i + 1) * B] for i in range(world)], 50.0, 10)
    got = M.get_matched_features([t.numpy() for t in ga], [t.numpy() for t in gb], 50.0, 10)
    flat = torch.tensor(np.concatenate(got[0], 0))
    loc = parallel.local_rows(flat, shards * B)
    np.testing.assert_allclose(loc.numpy(), np.concatenate(out_ref[0], 0)[mine], rtol=1e-12)
    # 3. gradient all-reduce is a SUM over ranks (train.py:134-139), through one flat bucket
    g1 = torch.full((3, 4), float(rank + 1))
    g2 = torch.arange(5, dtype=torch.float32) * (rank + 1)
    g1, g2 = parallel.allreduce_sum_([g1, g2])
    tot = sum(range(1, world + 1))
    assert torch.all(g1 == tot) and torch.equal(g2, torch.arange(5, dtype=torch.float32) * tot)
    # 4. bucketed all-reduce underneath backward == SUM over ranks of the plain gradients; variables that are
    #    evaluated but not differentiated (frozen) and un-armed passes leave the buckets alone
    torch.manual_seed(7)
    ws = [torch.randn(5, 4, requires_grad=True), torch.randn(4, requires_grad=True),
          torch.randn(4, 3, requires_grad=True), torch.randn(3, requires_grad=True)]
    x = torch.randn(6, 5) * (rank + 1)                      # rank-dependent data, same weights

    def net(inp):
        return (torch.tanh(inp @ ws[0] + ws[1]) @ ws[2] + ws[3]).pow(2).sum()

    plain = torch.autograd.grad(net(x), ws)
    ref = parallel.allreduce_sum_([g.clone() for g in plain])
    gb = parallel.GradBuckets(ws, nbuckets=3)
    assert len(gb.ranges) >= 2 and sum(gb.count) == len(ws)
    torch.autograd.grad(net(x), ws)                         # not armed: hooks must not touch the buckets
    assert float(gb.flat.abs().sum()) == 0.0
    for _ in range(2):                                      # re-armed every step
        gb.arm()
        torch.autograd.grad(net(x), ws)
        got = gb.finish()
        for a_, b_ in zip(got, ref):
            np.testing.assert_allclose(a_.numpy(), b_.numpy(), rtol=1e-6, atol=1e-7)
    gb.arm()
    ws[3].requires_grad_(False)
    torch.autograd.grad(net(x), ws[:3])
    ws[3].requires_grad_(True)
    with pytest.raises(RuntimeError):
        gb.finish()                                         # a variable without gradient is an error, not a hang
    parallel.barrier()
    out.put((rank, True))
    dist.destroy_process_group()


@pytest.mark.parametrize("serial", [False, True], ids=["overlapped", "serial_collectives"])
def test_two_rank_plumbing_gloo(serial, monkeypatch):
    # serial (parallel.collectives_mode): the gradient buckets go out after the backward pass instead of inside it
    monkeypatch.setenv("OTGAN_COLLECTIVES", "serial" if serial else "overlapped")
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, True), (1, True)]


def test_collectives_mode_is_guarded(monkeypatch):
    """Round 6: the mode is pinned by OTGAN_COLLECTIVES or decided by the start-up self-check; until that has run -- and whenever it
    cannot run (one rank without forced collectives, gloo, no device) -- 'auto' means serial."""
    from otgan_amd import parallel
    monkeypatch.delenv("OTGAN_COLLECTIVES", raising=False)
    monkeypatch.setitem(parallel._MODE, "mode", None)
    assert parallel.collectives_mode() == "serial"
    assert parallel.resolve_collectives_mode(None) == "serial" and "no exchange" in parallel.collectives_mode_reason()
    monkeypatch.setenv("OTGAN_COLLECTIVES", "overlapped")
    assert parallel.collectives_mode() == "overlapped" and parallel.resolve_collectives_mode(None) == "overlapped"
    assert "pinned" in parallel.collectives_mode_reason()
    monkeypatch.setenv("OTGAN_COLLECTIVES", "serial")
    assert parallel.collectives_mode() == "serial"
    monkeypatch.setenv("OTGAN_COLLECTIVES", "auto")
    monkeypatch.setitem(parallel._MODE, "mode", "overlapped")     # what a passed self-check leaves
    assert parallel.collectives_mode() == "overlapped"


def test_single_process_passthrough():
    from otgan_amd import parallel
    x = torch.randn(4, 3)
    assert parallel.all_gather_rows(x) is x
    assert parallel.world_size() == 1 and parallel.get_rank() == 0
    assert len(parallel.gather_feature_shards(x, 2)) == 2
    g = [torch.ones(2)]
    assert parallel.allreduce_sum_(g) is g
