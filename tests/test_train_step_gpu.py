"""GPU: one whole OT-GAN step (generator / critic forward, Sinkhorn matching, gradient
injection, gradient lists) of the HIP trainer vs the CPU oracle step (oracle/train_step_cpu.py:
PyTorch fp64 nets + NumPy fp64 matching) with identical parameters, data and latents.
Also: the parameter update itself (Adam with the reference's epsilon placement, EMA)."""
import numpy as np
import pytest
import torch

from oracle import nets_torch as NT
from oracle.train_step_cpu import CpuOTGAN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _noise(model, nb, gen):
    if model == "dcgan":
        return torch.rand(nb, 100, generator=gen) * 2 - 1
    return [torch.rand(nb, 100, generator=gen) * 2 - 1, torch.rand(nb, 8, 8, 16, generator=gen) * 2 - 1,
            torch.rand(nb, 16, 16, 16, generator=gen) * 2 - 1, torch.rand(nb, 32, 32, 16, generator=gen) * 2 - 1]


@pytest.mark.parametrize("model", ["dcgan", "densenet"])
@pytest.mark.parametrize("kind", ["disc", "gen"])
def test_step_gradients_match_oracle(dev, model, kind):
    from otgan_amd.trainer import OTGAN, default_args
    lam, iters = 100.0, 20
    args = default_args(model=model, batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters,
                        nr_gen_per_disc=1, seed=7)
    m = OTGAN(args, dev)
    if kind == "gen":
        m.step_counter = 1
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1
    noise = _noise(model, m.nb, gen)
    to_dev = lambda z: [t.to(dev) for t in z] if isinstance(z, list) else z.to(dev)
    r = m.step(x.to(dev), noise=to_dev(noise), apply_updates=False)
    assert r["kind"] == kind

    o = CpuOTGAN(model, "crelu", dtype=torch.float64, use_c_matching=False)
    named = {}
    named.update(m.discriminator.named_variables())
    named.update(m.generator.named_variables())
    o.load(named)
    to64 = lambda z: [t.double() for t in z] if isinstance(z, list) else z.double()
    gr, dist, ent = o.grads(kind, x.double(), to64(noise), 2, lam, iters)
    assert float(r["distance"]) == pytest.approx(dist, rel=2e-4, abs=1e-7)
    assert float(r["entropy"]) == pytest.approx(ent, rel=2e-4)
    # yardstick: the same step evaluated by the oracle in plain fp32.  Back-propagating through
    # ~100 CReLU layers and a lambda-amplified matching makes fp32 itself deviate from fp64 at
    # the 1e-3 level on the deepest tensors (sign flips of near-zero pre-activations); the HIP
    # path must be within 1e-2 or no worse than 3x that fp32 yardstick.  (The components are
    # pinned much tighter elsewhere: layers 2e-5, nets 5e-5..2e-4, matching 1e-4; a wiring error --
    # wrong shard order, sign, missing branch -- shows up here as an O(1) deviation.  The k-ordered
    # fp32 MFMA accumulation over K ~ 1e4 leaves ~2e-5 on the features, which the lambda-weighted
    # plan turns into up to 3e-3 on individual small gradient tensors.)
    o32 = CpuOTGAN(model, "crelu", dtype=torch.float32, use_c_matching=False)
    o32.load(named)
    to32 = lambda z: [t.float() for t in z] if isinstance(z, list) else z.float()
    gr32, _, _ = o32.grads(kind, x.float(), to32(noise), 2, lam, iters)
    # A pre-activation that is zero to rounding flips its CReLU unit between two fp32 evaluation orders; one flip
    # moves the small tensors near it by up to ~1.3e-2 and the median tensor to ~2e-3 (tools/debug/step_seeds.py:
    # over five seeds the per-layer and the batched weight norm -- same values to 1 ulp -- each land on either
    # side).  Hence a per-tensor bound of 3e-2 and a median bound, not 1e-2 everywhere.
    names = list((m.generator if kind == "gen" else m.discriminator).named_variables())
    errs = []
    for n, a, b, c in zip(names, r["grads"], gr, gr32):
        e_hip, e_32 = _rel(a, b), _rel(c, b)
        errs.append(e_hip)
        assert e_hip < max(3e-2, 3 * e_32), (n, e_hip, e_32)
    assert sorted(errs)[len(errs) // 2] < 5e-3, sorted(errs)[len(errs) // 2]


def test_updates_and_ema(dev):
    """Two full steps: parameters move by the reference's Adam (critic ascends: lr = -lr,
    train.py:143), the EMA shadows follow the generator (train.py:63-64,223)."""
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model="dcgan", batch_size=2, nr_gpu=2, sinkhorn_lambda=50.0, nr_sinkhorn_iter=5,
                        nr_gen_per_disc=1, seed=3)
    m = OTGAN(args, dev)
    d0 = [p.detach().clone() for p in m.disc_params]
    g0 = [p.detach().clone() for p in m.gen_params]
    x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
    u = torch.rand(m.nb, 100, device=dev) * 2 - 1
    r = m.step(x, noise=u, apply_updates=False)          # peek at the critic gradients
    grads = [t.clone() for t in r["grads"]]
    m.step_counter = 0
    m.step(x, noise=u)                                   # critic step
    for p, p0, gr in zip(m.disc_params, d0, grads):
        st = {"t": 1.0, "v": torch.zeros_like(p0, dtype=torch.float64).cpu(),
              "mg": torch.zeros_like(p0, dtype=torch.float64).cpu()}
        ref = NT.adam_update(p0.double().cpu(), gr.double().cpu(), st, -args.learning_rate_disc, 0.5, 0.999)
        assert _rel(p, ref) < 1e-6
    for p, p0 in zip(m.gen_params, g0):
        assert torch.equal(p.detach(), p0)               # untouched by the critic step
    m.step(x, noise=u)                                   # generator step
    moved = [not torch.equal(p.detach(), p0) for p, p0 in zip(m.gen_params, g0)]
    assert all(moved)
    for p, p0 in zip(m.gen_params, g0):
        sh = m.ema.average(p)
        ref = 0.999 * p0.double() + 0.001 * p.detach().double()
        assert _rel(sh, ref) < 1e-6


def test_random_and_single_batch_modes_run(dev):
    from otgan_amd.trainer import OTGAN, default_args
    x = None
    for kw in ({"no_sinkhorn": True}, {"single_batch": True}, {"train_disc_against_ema": True},
               {"optimizer": "adamax"}, {"optimizer": "nesterov"}):
        args = default_args(model="dcgan", batch_size=2, nr_gpu=2, nr_sinkhorn_iter=5, nr_gen_per_disc=1, **kw)
        m = OTGAN(args, dev)
        x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
        for _ in range(2):
            r = m.step(x)
            assert torch.isfinite(r["distance"]).item()


def test_image_size_64_config5(dev):
    """BASELINE config 5 shape (64x64 images, D = 131072): outside the reference (its generator is
    hard-coded to 32x32); the critic is size-agnostic and the generator takes an added
    image_size option."""
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model="dcgan", batch_size=2, nr_gpu=2, nr_sinkhorn_iter=5, nr_gen_per_disc=1,
                        image_size=64)
    m = OTGAN(args, dev)
    assert m.num_features == 131072
    x = torch.rand(m.nb, 64, 64, 3, device=dev) * 2 - 1
    for _ in range(2):
        r = m.step(x)
        assert torch.isfinite(r["distance"]).item()
    assert m.sample(2).shape == (2, 64, 64, 3)


def test_weight_cache_invalidation(dev):
    """The normalised-weight cache must not survive a parameter update."""
    from otgan_amd.models import dcgan
    from otgan_amd import ops
    dcgan.discriminator.reset(seed=9)
    x = torch.rand(2, 32, 32, 3, device=dev) * 2 - 1
    with torch.no_grad():
        f0 = dcgan.discriminator(x).clone()
        f1 = dcgan.discriminator(x)
        assert torch.equal(f0, f1)
        p = dcgan.discriminator.trainable_variables()[3]          # conv2d_1/V
        grad = torch.randn_like(p)
        ops.adam_step(p, grad, torch.zeros_like(p), torch.zeros_like(p), 1e-2, 0.5, 0.999, 1)
        f2 = dcgan.discriminator(x)
        assert not torch.equal(f0, f2)
        p.mul_(1.5)                                              # torch in-place op (version bump)
        f3 = dcgan.discriminator(x)
        # weight norm makes the output invariant to the scale of V: equal up to rounding
        assert float((f3 - f2).norm() / f2.norm()) < 1e-5


def test_checkpoint_roundtrip_and_learning_signal(dev, tmp_path):
    """(i) state_dict / load_state_dict reproduce the parameters (train.py:190-193,275-277);
    (ii) sanity of the signs of the whole loop: against a FROZEN critic, generator steps must
    reduce the mini-batch energy distance to a fixed data batch."""
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model="dcgan", batch_size=16, nr_gpu=2, sinkhorn_lambda=100.0, nr_sinkhorn_iter=30,
                        nr_gen_per_disc=10 ** 6, learning_rate_gen=1e-3, seed=2)
    m = OTGAN(args, dev)
    torch.manual_seed(0)
    # structured "data": smooth colour blobs, far from the initial generator output
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 32), torch.linspace(-1, 1, 32), indexing="ij")
    base = torch.stack([torch.sin(3 * xx), torch.cos(2 * yy), xx * yy], -1).to(dev)
    x = (base[None] * (0.5 + 0.5 * torch.rand(m.nb, 1, 1, 3, device=dev))).clamp(-1, 1).contiguous()
    u = torch.rand(m.nb, 100, device=dev) * 2 - 1
    m.step_counter = 1                      # generator steps only (critic frozen)
    d = []
    for _ in range(25):
        m.step_counter = max(m.step_counter, 1)
        d.append(float(m.step(x, noise=u)["distance"]))
    assert d[-1] < 0.7 * d[0], (d[0], d[-1])
    sd = m.state_dict()
    path = tmp_path / "ckpt"
    torch.save(sd, path)
    m2 = OTGAN(default_args(model="dcgan", batch_size=16, nr_gpu=2, seed=99), dev)
    m2.load_state_dict(torch.load(path))
    for a, b in zip(m.gen_params + m.disc_params, m2.gen_params + m2.disc_params):
        assert torch.equal(a.detach(), b.detach())
    assert m2.step_counter == m.step_counter


# ------------------------------------------------------------------ tighter step-level / mode parity (round 2)
def _named(m):
    named = {}
    named.update(m.discriminator.named_variables())
    named.update(m.generator.named_variables())
    return named


def _well_conditioned_worst(dev, model, size, kind, seed, same_head_signs=False):
    """Worst per-tensor relative L2 error of the step's gradients against the fp64 oracle step (+ distance / entropy
    checks) for one parameter / data seed.  `same_head_signs`: the oracle's feature head takes its CReLU sign pattern
    from the path under test (oracle/nets_torch.py FORCED_HEAD_SIGNS)."""
    from otgan_amd.trainer import OTGAN, default_args
    from otgan_amd.utils import nn as hip_nn
    from oracle import nets_torch as NTO
    signs = []
    real_head = hip_nn.feature_head

    def recording_head(z):
        signs.append(torch.sign(z.detach()).cpu())      # (an activation that rounds to exactly 0 has happened: seed 7 at 64 x 64)
        return real_head(z)
    lam, iters = 20.0, 10
    args = default_args(model=model, batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters,
                        nr_gen_per_disc=1, seed=seed, nonlinearity="elu", image_size=size)
    m = OTGAN(args, dev)
    if kind == "gen":
        m.step_counter = 1
    gen = torch.Generator().manual_seed(7 + seed)
    x = torch.rand(m.nb, size, size, 3, generator=gen) * 2 - 1
    noise = _noise(model, m.nb, gen)
    to_dev = lambda z: [t.to(dev) for t in z] if isinstance(z, list) else z.to(dev)
    hip_nn.feature_head = recording_head
    try:
        r = m.step(x.to(dev), noise=to_dev(noise), apply_updates=False)
    finally:
        hip_nn.feature_head = real_head
    assert r["kind"] == kind and len(signs) == (2 if kind == "gen" else 1)
    o = CpuOTGAN(model, "elu", dtype=torch.float64, use_c_matching=False, image_size=size)
    o.load(_named(m))
    to64 = lambda z: [t.double() for t in z] if isinstance(z, list) else z.double()
    n_heads = len(signs)
    NTO.FORCED_HEAD_SIGNS = signs if same_head_signs else None       # (the oracle pops them)
    del NTO.FORCED_HEAD_REPORT[:]
    try:
        gr, dist, ent = o.grads(kind, x.double(), to64(noise), 2, lam, iters)
    finally:
        NTO.FORCED_HEAD_SIGNS = None
    if same_head_signs:
        # the forced pattern may differ from the oracle's own ONLY in a handful of units that sit at rounding level: a
        # forward kernel that flipped many units, or one that is not tiny, fails here
        flipped = sum(c for c, _ in NTO.FORCED_HEAD_REPORT)
        biggest = max([v for _, v in NTO.FORCED_HEAD_REPORT] + [0.0])
        assert len(NTO.FORCED_HEAD_REPORT) == n_heads
        # (measured: 0 - 2 units per case, |x| <= 4e-6 of the sample's RMS -- the forward features are good to 2-4e-6)
        assert flipped <= 8 and biggest <= 2e-5, (flipped, biggest)
    assert float(r["distance"]) == pytest.approx(dist, rel=1e-4, abs=1e-7)
    assert float(r["entropy"]) == pytest.approx(ent, rel=1e-4)
    names = list((m.generator if kind == "gen" else m.discriminator).named_variables())
    worst = max((_rel(a, b), n) for n, a, b in zip(names, r["grads"], gr))
    m.close()
    return worst


# measured (round 3, seeds 5 / 6 / 7, worst tensor per seed, head sign pattern shared with the oracle; pytest -s prints them):
#   dcgan 32 disc 6.3e-6, 6.2e-6, 6.3e-6      gen 8.2e-6, 6.0e-6, 8.0e-6
#   densenet disc 5.8e-6, 8.4e-6, 6.3e-6      gen 4.3e-6, 7.5e-6, 3.6e-6
#   dcgan 64 disc 6.0e-6, 5.9e-6, 5.9e-6      gen 1.1e-5, 7.5e-6, 9.2e-6
# (with the oracle's own signs: dcgan 32 disc seed 6 6.2e-4, dcgan 64 disc seed 5 1.2e-3 and seed 7 4e-4 -- one unit each)
# asserted: every seed at 3 x the measured maximum (see the test)
# round 4: 3 x the measured maximum instead of + 25 % -- a compiler or summation-order change moves these by tens of
# per cent; the sign pattern the oracle takes over is now counted and bounded (<= 8 units, |x| <= 2e-5 of the sample's RMS)
_WELL_TOL = {("dcgan", 32, "disc"): 1.9e-5, ("dcgan", 32, "gen"): 2.5e-5, ("densenet", 32, "disc"): 2.5e-5,
             ("densenet", 32, "gen"): 2.3e-5, ("dcgan", 64, "disc"): 1.8e-5, ("dcgan", 64, "gen"): 3.3e-5}


@pytest.mark.parametrize("model,size", [("dcgan", 32), ("densenet", 32), ("dcgan", 64)])
@pytest.mark.parametrize("kind", ["disc", "gen"])
def test_well_conditioned_step_gradients(dev, model, size, kind):
    """Whole-step numerics in a well-conditioned setting: ELU, lambda = 20, 10 sweeps (no lambda-amplified
    cancellation).  EVERY gradient tensor of the step against the fp64 oracle step, distance and entropy to 1e-4.
    size = 64 is BASELINE configs[4]'s shape (generator stem 8x8, D = 65536 with ELU).

    Round 3: three parameter / data seeds per case, EVERY seed asserted at what the default engine (Winograd F(4x4,3x3),
    two scaled fp16 pieces) achieves + 25 % -- 1e-5-class, the level of a plain fp32 evaluation (PyTorch-CPU fp32: 5e-6
    on these gradients; tests/test_engine_accuracy_gpu.py puts the four GEMM engines side by side) -- with the oracle's
    feature head taking its CReLU sign pattern from the path under test.  Round 2 asserted 1e-3 / 3e-3 here and
    attributed it to the Winograd transform; the decomposition (tools/debug/step_error_parts.py: features 2-4e-6,
    matching on fixed features 1e-6, backward on a fixed upstream gradient 2e-6) shows what it really was: the feature
    head is a CReLU whatever --nonlinearity says (models/dcgan.py:16,19), and ONE of its 1e5 .. 8e5 pre-activations landing
    on the other side of zero than in fp64 moves every gradient of the step by 2e-4 .. 1e-3.  That is a coin flip per
    engine, kernel and seed: at 32 x 32 about one seed in three has such a unit, at 64 x 64 (four times the units) three to
    five in six (tools/debug/step_seeds_64.py; changing the summation order of the generator's RGB-out convolution --
    to a MORE accurate kernel -- moved them to other seeds).  With the same sign pattern on both sides the comparison
    measures arithmetic and nothing else; one un-forced seed per step kind stays (DCGAN 32 x 32), bounded at the
    flipped-unit level."""
    tol = _WELL_TOL[(model, size, kind)]
    # (three seeds at 32 x 32 DCGAN, two for the cases whose fp64 CPU oracle takes 10+ s per seed: the GPU suite has a budget)
    seeds = (5, 6, 7) if (model, size) == ("dcgan", 32) else (5, 6)
    worst = [_well_conditioned_worst(dev, model, size, kind, seed, same_head_signs=True) for seed in seeds]
    print(f"\nwell-conditioned {model} {size} {kind}: worst tensor per seed " + ", ".join(f"{w[0]:.2e} ({w[1]})" for w in worst))
    assert max(w[0] for w in worst) < tol, worst
    if (model, size) == ("dcgan", 32):          # (one un-forced run per step kind: the suite's time budget)
        free = _well_conditioned_worst(dev, model, size, kind, 5)
        print(f"  un-forced head signs, seed 5: {free[0]:.2e} ({free[1]})")
        assert free[0] < 3e-3, free             # a flipped head unit: bounded, not tight


def _ema_critic_errors(dev, seed, same_head_signs=True):
    from otgan_amd.trainer import OTGAN, default_args
    from otgan_amd.utils import nn as hip_nn
    from oracle import nets_torch as NTO
    lam, iters = 20.0, 10
    args = default_args(model="dcgan", batch_size=3, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters,
                        nr_gen_per_disc=1, seed=seed, nonlinearity="elu", train_disc_against_ema=True,
                        learning_rate_gen=0.05)       # large step: shadows and weights clearly apart
    m = OTGAN(args, dev)
    gen = torch.Generator().manual_seed(5 + seed)
    x = (torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev)
    u = (torch.rand(m.nb, 100, generator=gen) * 2 - 1).to(dev)
    m.step(x, noise=u)        # critic update
    m.step(x, noise=u)        # generator update + EMA
    names_g = list(m.generator.named_variables())
    shadow = {n: m.ema.average(p) for n, p in zip(names_g, m.gen_params)}
    p0 = m.gen_params[0]
    assert _rel(shadow[names_g[0]], p0) > 1e-5                       # the EMA generator is a different network
    signs = []
    real_head = hip_nn.feature_head

    def recording_head(z):
        signs.append(torch.sign(z.detach()).cpu())
        return real_head(z)
    hip_nn.feature_head = recording_head
    try:
        r = m.step(x, noise=u, apply_updates=False)
    finally:
        hip_nn.feature_head = real_head
    assert r["kind"] == "disc" and len(signs) == 1
    o = CpuOTGAN("dcgan", "elu", dtype=torch.float64, use_c_matching=False)
    o.load(_named(m))
    NTO.FORCED_HEAD_SIGNS = list(signs) if same_head_signs else None
    del NTO.FORCED_HEAD_REPORT[:]
    try:
        gr, dist, ent = o.grads("disc", x.double().cpu(), u.double().cpu(), 2, lam, iters, ema_P=o.ema_params(shadow))
    finally:
        NTO.FORCED_HEAD_SIGNS = None
    if same_head_signs:
        flipped = sum(c for c, _ in NTO.FORCED_HEAD_REPORT)
        biggest = max([v for _, v in NTO.FORCED_HEAD_REPORT] + [0.0])
        assert flipped <= 8 and biggest <= 2e-5, (flipped, biggest)
    assert float(r["distance"]) == pytest.approx(dist, rel=1e-4, abs=1e-7)
    assert float(r["entropy"]) == pytest.approx(ent, rel=1e-4)
    gr_live, dist_live, _ = o.grads("disc", x.double().cpu(), u.double().cpu(), 2, lam, iters)
    assert abs(dist_live - dist) > 1e-3 * abs(dist)                  # NOT what the live generator would give
    e_ema = max(_rel(a, b) for a, b in zip(r["grads"], gr))
    e_live = min(_rel(a, c) for a, c in zip(r["grads"], gr_live))
    m.close()
    return e_ema, e_live


def test_ema_critic_step_matches_oracle(dev):
    """--train_disc_against_ema (train.py:102-103,119-123): on a critic step the generated branch is the EMA
    generator's samples (and its own matching).  After one critic + one generator update the shadows differ
    from the weights; the next critic step is compared with the oracle fed the same shadows.

    Round 3: three seeds.  A seed may carry a flipped CReLU unit of the feature head (models/dcgan.py:16,19: every
    gradient of that step then moves by ~1e-3 -- round 2 asserted 5e-3 for everything because of it; after the two
    updates of size 0.05 this happens more often than at initialisation: measured 6.1e-6, 2.6e-3, 1.4e-3).  Every seed is
    bounded at that level and pins the BRANCH (the gradients are those of the EMA generator's samples, at least 20 x
    closer to them than to the live generator's); the BEST seed must reach the accuracy of the well-conditioned step
    above (measured + 25 %), which is what the arithmetic of this branch delivers when no unit flips."""
    res = [_ema_critic_errors(dev, seed) for seed in (8, 9, 10)]
    print("\nEMA critic step: (worst error vs EMA oracle, closest vs live oracle) per seed:", [(f"{a:.2e}", f"{b:.2e}") for a, b in res])
    # round 4: the oracle's feature head takes the sign pattern of the path under test (counted and bounded, as in
    # test_well_conditioned_step_gradients), so EVERY seed measures arithmetic: all three at 3 x the level the un-flipped
    # seed reached in round 3 (6.1e-6), and the branch is pinned on every seed
    for e_ema, e_live in res:
        assert e_ema < 1.9e-5 and e_live > 20 * e_ema, res
    free = _ema_critic_errors(dev, 9, same_head_signs=False)       # un-forced: bounded at the flipped-unit level
    assert free[0] < 5e-3 and free[1] > 20 * free[0], free


@pytest.mark.parametrize("model,iters,D", [("dcgan", 100, 32768), ("densenet", 200, 7296)])
def test_full_batch_step_is_finite_and_its_distance_matches_the_matching_oracle(dev, model, iters, D):
    """One whole critic step and one whole generator step at the BASELINE batch (256 images = 2 shards x 128, lambda 500,
    configs[1] / configs[3] iteration counts): every gradient finite, and the step's reported distance / entropy equal the
    fp64 matching oracle evaluated on the features the step itself produced (recorded at the feature head) to 1e-4."""
    from otgan_amd.trainer import OTGAN, default_args
    from otgan_amd.utils import nn as hip_nn
    from oracle import matching_np as M
    lam = 500.0
    args = default_args(model=model, batch_size=128, nr_gpu=2, sinkhorn_lambda=lam, nr_sinkhorn_iter=iters,
                        nr_gen_per_disc=1, seed=3)
    m = OTGAN(args, dev)
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev)
    feats = []
    real_head = hip_nn.feature_head

    def recording_head(z):
        f = real_head(z)
        feats.append(f.detach())
        return f
    for kind in ("disc", "gen"):
        del feats[:]
        hip_nn.feature_head = recording_head
        try:
            r = m.step(x, apply_updates=False)
        finally:
            hip_nn.feature_head = real_head
        assert r["kind"] == kind
        for g in r["grads"]:
            assert bool(torch.isfinite(g).all())
        if kind == "disc":          # one critic pass over [data; generated]
            f_dat, f_gen = feats[0][:m.nb], feats[0][m.nb:]
        else:                       # critic on the data (no grad), then on the generated images
            f_dat, f_gen = feats[0], feats[1]
        assert f_gen.shape == (m.nb, D)
        fa, fb = f_gen.double().cpu().numpy(), f_dat.double().cpu().numpy()
        N = m.nb // 2
        plans, costs, ent_ref = M.two_batch_plans(fa[:N], fa[N:], fb[:N], fb[N:], lam, iters)
        dref = M.closed_form_from(plans, costs, N)
        assert abs(float(r["distance"]) - dref) <= 1e-4 * abs(dref) + 1e-7, (kind, float(r["distance"]), dref)
        assert float(r["entropy"]) == pytest.approx(float(ent_ref), rel=2e-4)
    m.close()


@pytest.mark.parametrize("opt", ["adamax", "nesterov"])
def test_adamax_nesterov_through_trainer(dev, opt):
    """--optimizer adamax / nesterov through the trainer: two critic updates on fixed inputs reproduce the
    oracle's update rule (nn.py:29-48,75-87) applied to the trainer's own gradients; lr = -lr for the critic
    (train.py:143); train.py:142-149 passes mom1 = 0.5 (and mom2 = 0.999 to adamax)."""
    from otgan_amd.trainer import OTGAN, default_args
    args = default_args(model="dcgan", batch_size=2, nr_gpu=2, sinkhorn_lambda=50.0, nr_sinkhorn_iter=5,
                        nr_gen_per_disc=10 ** 6, seed=4, optimizer=opt, learning_rate_disc=1e-3)
    m = OTGAN(args, dev)
    x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
    u = torch.rand(m.nb, 100, device=dev) * 2 - 1
    upd = NT.adamax_update if opt == "adamax" else NT.nesterov_update
    states = None
    for it in range(2):
        p_before = [p.detach().double().cpu() for p in m.disc_params]
        m.step_counter = 0
        grads = [g.double().cpu() for g in m.step(x, noise=u, apply_updates=False)["grads"]]
        m.step_counter = 0
        m.step(x, noise=u)
        if states is None:
            states = [{"v": torch.zeros_like(p), "mg": torch.zeros_like(p)} for p in p_before]
        for p, p0, g, st in zip(m.disc_params, p_before, grads, states):
            kw = dict(mom1=0.5, mom2=0.999) if opt == "adamax" else dict(mom1=0.5)
            ref = upd(p0, g, st, -args.learning_rate_disc, **kw)
            assert _rel(p, ref) < 1e-6, (opt, it)
