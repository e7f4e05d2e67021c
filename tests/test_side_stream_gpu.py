"""GPU: the second stream of a single-process step (trainer.OTGAN: the critic's pass over the real batch under the generator's
forward pass, every layer's weight-gradient chain under the input-gradient chain in front of it, the input-gradient filters
prepared under the forward pass) changes WHEN kernels run, not what they compute: a run with OTGAN_SIDE_STREAM=0 (everything
on one stream) and the default run end with bit-identical parameters, EMA shadows, optimiser moments and losses
(reference train.py:207-226 is one `sess.run` per step either way)."""
import pytest
import torch

from test_step_graph_gpu import _run, _same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("model,ngd,kw", [("dcgan", 2, {}), ("densenet", 1, {}), ("dcgan", 1, {"train_disc_against_ema": True}),
                                          ("dcgan", 2, {"single_batch": True})],
                         ids=["dcgan", "densenet", "dcgan_ema_critic", "dcgan_single_batch"])
def test_two_streams_equal_one_stream(dev, model, ngd, kw, monkeypatch):
    steps = 3 * (ngd + 1) + 1
    monkeypatch.setenv("OTGAN_SIDE_STREAM", "0")
    one = _run(dev, model, False, steps, ngd, **kw)
    monkeypatch.delenv("OTGAN_SIDE_STREAM")
    two = _run(dev, model, False, steps, ngd, **kw)
    _same(one, two)


def test_side_stream_is_on_by_default_and_under_step_graphs(dev, monkeypatch):
    from otgan_amd.trainer import OTGAN, default_args
    monkeypatch.delenv("OTGAN_SIDE_STREAM", raising=False)
    m = OTGAN(default_args(batch_size=2, nr_gpu=2, nr_sinkhorn_iter=5), dev)
    assert m.fork_real_pass and m.fork_wgrad and m._side_stream is not None
    m.close()
    m = OTGAN(default_args(batch_size=2, nr_gpu=2, nr_sinkhorn_iter=5, step_graph=True), dev)
    assert m.fork_real_pass and m.fork_wgrad and m.graphs is not None      # round 6: the capture holds both streams' chains
    m.close()


def test_allocator_reaches_a_steady_state_with_the_host_running_ahead(dev, monkeypatch):
    """Two streams + record_stream + a host that never synchronises made torch's caching allocator call hipMalloc about four
    times per step through the first ~150 steps of a run (a tensor the other stream has used is reusable only once that
    stream's work on it has COMPLETED).  trainer.OTGAN.step keeps the host at most two steps ahead of the device: after
    two warm-up periods a window of steps without any synchronisation allocates (next to) nothing."""
    from otgan_amd.trainer import OTGAN, default_args
    monkeypatch.delenv("OTGAN_SIDE_STREAM", raising=False)
    m = OTGAN(default_args(batch_size=32, nr_gpu=2, nr_sinkhorn_iter=10, nr_gen_per_disc=2, seed=3), dev)
    assert m._side_stream is not None and m._max_ahead == 2
    x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
    for _ in range(9):
        m.step(x)
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    for _ in range(18):
        m.step(x)
    torch.cuda.synchronize()
    n1 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    m.close()
    assert n1 - n0 <= 4, (n0, n1)


def test_variable_used_twice_under_the_side_stream(dev):
    """a variable consumed by two nodes of one graph has its two weight gradients (both produced on the side stream)
    added by the autograd engine on the main stream; the second node makes the main stream wait first.  Same bits as on one
    stream, over several repetitions (the race was latent: today's models use every variable once per graph)."""
    from otgan_amd import ops
    gen = torch.Generator(device=dev).manual_seed(9)
    xs = [torch.randn((64, 16, 16, 128), generator=gen, device=dev).requires_grad_(True) for _ in range(2)]
    V = (torch.randn((5, 5, 256, 128), generator=gen, device=dev) * 0.05).requires_grad_(True)
    g = torch.ones(128, device=dev, requires_grad=True)
    b = torch.zeros(128, device=dev, requires_grad=True)
    dys = [torch.randn((64, 8, 8, 128), generator=gen, device=dev) for _ in range(2)]

    def run(side):
        ops.SIDE_STREAM = side
        try:
            ys = [ops.conv2d_op(x, V, g, b, stride=2, preact=ops.ACT["crelu"]) for x in xs]
            grads = torch.autograd.grad(ys, xs + [V, g, b], dys)
            ops.join_side_stream(grads)
        finally:
            ops.SIDE_STREAM = None
        torch.cuda.synchronize()
        return [t.clone() for t in grads]

    one = run(None)
    side = torch.cuda.Stream(device=dev)
    for _ in range(5):
        two = run(side)
        for u, v in zip(one, two):
            assert torch.equal(u, v)
