"""GPU (1 device): `python bench.py --gpus 2` started PLAINLY -- no torchrun -- must launch its own two ranks
(reference train.py:34-35,72-85: one command drives every device).  On a 1-GPU box the two ranks share cuda:0 over
gloo (OTGAN_SINGLE_DEVICE=1: a logic test of the global-scope step, not a measurement); without that override the
launch must fail loudly."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "OTGAN_DIST_BACKEND", "OTGAN_FORCE_COLLECTIVES")}
    env.update(kw)
    return env


def test_bench_gpus2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1",
                        "--no_cpu_baseline", "--no_secondary", "--no_prof"],
                       env=_env(OTGAN_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    c = d["config"]
    assert c["ranks"] == 2 and c["backend"].startswith("gloo")
    assert c["matching_scope"] == "global"
    assert c["sinkhorn_rows"] == 256                                # 2 ranks x 256 img = 4 shards x 128: N = 256
    assert c["global_batch"] == 512
    assert c["step_mix"] == {**c["step_mix"], "critic_steps": 1, "generator_steps": 5}
    import math
    assert math.isfinite(c["last_distance"]) and math.isfinite(c["last_entropy"])
    # per-rank exchange times (round 5): both ranks report matching / all-gather / all-reduce milliseconds per step
    per = c["rank_times"]["per_rank"]
    assert [t["rank"] for t in per] == [0, 1]
    for t in per:
        assert t["matching_ms"] > 0 and t["allgather_ms"] > 0 and t["allreduce_ms"] > 0
        assert t["matching_ms"] + t["allgather_ms"] + t["allreduce_ms"] < t["ms_per_step"]


def test_bench_gpus2_without_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(OTGAN_SINGLE_DEVICE=""), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "2 ranks requested but this node exposes 1 GPU" in r.stderr, r.stderr[-2000:]
    assert "Traceback" not in r.stderr


def test_train_nr_gpu_self_launches():
    """`python train.py --nr_gpu 4 --ranks 2` started plainly: two ranks x two shards, global matching scope."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--nr_gpu", "4", "--ranks", "2", "--batch_size", "8",
                            "--synthetic", "--synthetic_size", "64", "--max_steps", "2", "--nr_sinkhorn_iter", "10",
                            "--nr_gen_per_disc", "1", "--save_dir", td],
                           env=_env(OTGAN_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        assert r.stdout.count("starting training") == 1             # rank 0 prints
        assert "Iteration 0" in r.stdout
