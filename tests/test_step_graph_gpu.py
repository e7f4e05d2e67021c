"""GPU: whole training steps captured as hipGraphs and replayed (trainer.GraphedSteps) are the SAME arithmetic as the eager
steps -- the reference's `sess.run` loop (train.py:207-226: a critic step when step % (nr_gen_per_disc + 1) == 0, else a
generator step).  Two runs from the same seed on the same data, one eager throughout, one that switches to graph replays after
its first period, must end with bit-identical parameters, EMA shadows, optimiser moments and losses: same launches, same
arguments, Adam's bias corrections and the latent draws from device memory."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(dev, model, graph, steps, nr_gen_per_disc, eager_at=(), **kw):
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model=model, batch_size=4, nr_gpu=2, sinkhorn_lambda=100.0, nr_sinkhorn_iter=20,
                        nr_gen_per_disc=nr_gen_per_disc, seed=3, step_graph=graph, **kw)
    m = OTGAN(args, dev)
    g = torch.Generator().manual_seed(11)
    xs = [(torch.rand(m.nb, 32, 32, 3, generator=g) * 2 - 1).to(dev) for _ in range(4)]
    torch.manual_seed(7)
    dists, kinds = [], []
    for i in range(steps):
        if i in eager_at:       # a step with an injected latent always runs eagerly (tests, data-dependent tools)
            u = (torch.rand(m.nb, 100, generator=torch.Generator().manual_seed(100 + i)) * 2 - 1).to(dev)
            r = m.step(xs[i % 4], noise=u)
        else:
            r = m.step(xs[i % 4])
        dists.append(r["distance"].clone())
        kinds.append(r["kind"])
    state = {k: v.clone() for k, v in m.state_dict(full=True).items() if torch.is_tensor(v)}
    opt = m.state_dict(full=True)["__optim__"]
    ema = m.state_dict(full=True)["__ema__"]
    captured = sorted(m.graphs.graphs) if m.graphs is not None else []
    dead = m.graphs.dead if m.graphs is not None else None
    t = (m.gen_optimizer.t, m.disc_optimizer.t, m.step_counter)
    m.close()
    return {"state": state, "opt": opt, "ema": ema, "dists": torch.stack(dists).cpu(), "kinds": kinds, "captured": captured,
            "dead": dead, "t": t}


def _same(a, b):
    assert a["kinds"] == b["kinds"] and a["t"] == b["t"]
    assert torch.equal(a["dists"], b["dists"]), (a["dists"], b["dists"])
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    for k in a["ema"]:
        assert torch.equal(a["ema"][k], b["ema"][k]), k
    for net in ("gen", "disc"):
        assert a["opt"][net]["t"] == b["opt"][net]["t"]
        for sa, sb in zip(a["opt"][net]["slots"], b["opt"][net]["slots"]):
            for k in sa:
                assert (sa[k] is None and sb[k] is None) or torch.equal(sa[k], sb[k]), (net, k)


@pytest.mark.parametrize("model,ngd", [("dcgan", 3), ("dcgan", 1), ("densenet", 2)], ids=["dcgan_3to1", "dcgan_1to1", "densenet_2to1"])
def test_replayed_steps_equal_eager_steps(dev, model, ngd):
    steps = 3 * (ngd + 1) + 2
    eager = _run(dev, model, False, steps, ngd)
    graph = _run(dev, model, True, steps, ngd)
    assert eager["captured"] == [] and graph["dead"] is None
    assert graph["captured"] == (["disc", "gen", "gen1"] if ngd > 1 else ["disc", "gen1"])
    _same(eager, graph)


def test_eager_steps_between_replays(dev):
    """A step that must run eagerly (here: an injected latent) in the middle of replayed steps -- also at the position of the
    generator step that refreshes the critic's cached operands -- leaves the run on the eager trajectory."""
    ngd, steps = 2, 16
    at = (7, 10, 12)             # phases 1 ("gen1": the refresh), 1 again, 0 (critic)
    eager = _run(dev, "dcgan", False, steps, ngd, eager_at=at)
    graph = _run(dev, "dcgan", True, steps, ngd, eager_at=at)
    assert graph["dead"] is None and graph["captured"] == ["disc", "gen", "gen1"]
    _same(eager, graph)


def test_other_optimisers_and_ema_critic(dev):
    """Adamax (no step count in the update) and the EMA-generator critic step (--train_disc_against_ema: a third set of cached
    operands, recomputed in every critic step) under replay."""
    for kw in (dict(optimizer="adamax"), dict(train_disc_against_ema=True)):
        eager = _run(dev, "dcgan", False, 9, 2, **kw)
        graph = _run(dev, "dcgan", True, 9, 2, **kw)
        assert graph["dead"] is None and graph["captured"] == ["disc", "gen", "gen1"]
        _same(eager, graph)
