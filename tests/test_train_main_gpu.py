"""GPU: the command-line driver end to end (reference train.py:156-281 run loop) and checkpoint / resume.

  * `train.main([...])` with the reference's flags + --synthetic runs critic and generator steps, writes the
    sample sheets and a checkpoint under the reference's name `med_gan_params-<epoch>` (train.py:275-277), and
    `--load_params --model_name ...` (train.py:190-193) resumes from it;
  * the checkpoint carries what the reference's Saver omits: optimiser moments + step count and the
    EMA shadows -- a resumed step is BIT-IDENTICAL to the uninterrupted one;
  * maybe_flip (train.py:163-170) and save_tile_png (utils/plotting.py:9-13,29-74)
    get direct checks (load_cifar, data/cifar10_data.py:29-53, is checked on CPU in tests/test_cli_cpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_train_main_runs_saves_and_resumes(dev, tmp_path, capsys):
    from otgan_amd import train
    save = str(tmp_path / "run")
    common = ["--synthetic", "--synthetic_size", "48", "--nr_gpu", "2", "--batch_size", "8", "--nr_sinkhorn_iter", "10",
              "--sinkhorn_lambda", "100", "--nr_gen_per_disc", "2", "--save_dir", save, "--save_every", "1", "--seed", "3"]
    # 48 images / (2 x 8) = 3 steps per epoch; 7 steps = epochs 0, 1 complete (+1 step): critic steps at 0, 3, 6
    m = train.main(common + ["--max_steps", "7"])
    out = capsys.readouterr().out
    assert "model has a hidden representation with 32768 features" in out       # train.py:56
    assert "Iteration 0, time" in out and "train distance before gen" in out      # train.py:231
    assert m.step_counter == 7
    for f in ("sample0.png", "ema_sample0.png", "med_gan_params-1", "distances.npz"):
        assert os.path.exists(os.path.join(save, f)), f
    d = np.load(os.path.join(save, "distances.npz"))
    # epochs 0 and 1 are complete (one critic + two generator steps each); the truncated epoch 2 holds a single
    # critic step, so its generator mean is the mean of an empty list (nan), exactly like np.mean([]) in train.py:229
    assert np.isfinite(d["mean_dist_gen"][:2]).all() and np.isfinite(d["mean_dist_disc"][:2]).all()
    sd = torch.load(os.path.join(save, "med_gan_params-1"), map_location="cpu")
    assert "discriminator/conv2d_0/V" in sd and "generator/dense_0/V" in sd      # the reference's variable names
    assert sd["step_counter"] == 6 and "__optim__" in sd and "__ema__" in sd
    # resume (train.py:190-193): epoch parsed from the suffix, parameters / optimiser / EMA restored
    m2 = train.main(common + ["--max_steps", "1", "--load_params", "--model_name", "med_gan_params-1"])
    assert m2.step_counter == 7
    out2 = capsys.readouterr().out
    assert "Iteration 1, time" in out2


def test_resumed_step_is_bit_identical(dev, tmp_path):
    from otgan_amd.trainer import OTGAN, default_args
    kw = dict(model="dcgan", batch_size=4, nr_gpu=2, sinkhorn_lambda=100.0, nr_sinkhorn_iter=10, nr_gen_per_disc=1)
    m = OTGAN(default_args(seed=6, **kw), dev)
    gen = torch.Generator().manual_seed(1)
    xs = [(torch.rand(m.nb, 32, 32, 3, generator=gen) * 2 - 1).to(dev) for _ in range(5)]
    us = [(torch.rand(m.nb, 100, generator=gen) * 2 - 1).to(dev) for _ in range(5)]
    for i in range(3):
        m.step(xs[i], noise=us[i])
    path = tmp_path / "ckpt"
    torch.save(m.state_dict(), path)
    for i in range(3, 5):                        # uninterrupted: one generator, one critic step more
        m.step(xs[i], noise=us[i])
    want = {k: v.clone() for k, v in m.state_dict().items() if torch.is_tensor(v)}
    want_ema = {k: v.clone() for k, v in m.state_dict()["__ema__"].items()}

    m2 = OTGAN(default_args(seed=77, **kw), dev)  # different init: everything must come from the file
    m2.load_state_dict(torch.load(path))
    assert m2.step_counter == 3 and m2.gen_optimizer.t == m.gen_optimizer.t - 1
    for i in range(3, 5):
        m2.step(xs[i], noise=us[i])
    got = m2.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    for k, v in want_ema.items():
        assert torch.equal(got["__ema__"][k], v), k

    # a weights-only checkpoint (what the reference's Saver writes): EMA shadows restart at the loaded weights
    m3 = OTGAN(default_args(seed=78, **kw), dev)
    m3.load_state_dict(m.state_dict(full=False))
    for p in m3.gen_params:
        assert torch.equal(m3.ema.average(p), p.detach())


def test_maybe_flip_and_tile_png(dev, tmp_path):
    from otgan_amd import train
    torch.manual_seed(0)
    x = torch.rand(64, 32, 32, 3, device=dev) * 2 - 1
    y = train.maybe_flip(x)
    same = (y == x).flatten(1).all(1)
    flipped = (y == x.flip(2)).flatten(1).all(1)
    assert bool((same | flipped).all()) and 8 < int(flipped.sum()) < 56          # per-image coin (train.py:163-170)
    path = str(tmp_path / "sheet.png")
    train.save_tile_png(x, path, n=16)
    from PIL import Image
    im = np.asarray(Image.open(path))
    assert im.shape == (4 * 33 + 1, 4 * 33 + 1, 3)
    want = x[0].clamp(-1, 1).add(1).mul(127.5).byte().cpu().numpy()
    assert np.array_equal(im[1:33, 1:33], want)
