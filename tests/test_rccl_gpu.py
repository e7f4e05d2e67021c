"""GPU: RCCL readiness on a single-GPU box.

`backend="nccl"` (= RCCL on ROCm) is initialised at world size 1 and the collective code paths of
ot-gan_amd/parallel.py are FORCED to run (OTGAN_FORCE_COLLECTIVES=1 disables the world-size-1 early-outs):
`all_gather_rows`, `all_gather_rows_async` (ncclAllGather on RCCL's stream, overlapping compute), `allreduce_sum_`
and `GradBuckets` (tensor hooks firing under torch.autograd.grad on device tensors, async ncclAllReduce per
bucket).  With one rank every collective is an identity, so results must be BIT-IDENTICAL to the non-collective
path -- what is exercised is the RCCL initialisation, the calls, their stream ordering against the HIP kernels
of the step, and the hook plumbing.  The whole trainer step and `bench.py --gpus 1` run under the override too.
(The 2/4/8-GPU curve itself can only be measured by the driver; DESIGN.md section 6 states the expected
bytes per step and link.)  Runs in a child process: the process group must not leak into the other tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = os.path.join(ROOT, "tests", "rccl_ws1_worker.py")


def _env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update(OTGAN_ROOT=ROOT, OTGAN_FORCE_COLLECTIVES="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OTGAN_DIST_BACKEND", None)
    return env


def test_nccl_world_size_1_forced_collectives():
    r = subprocess.run([sys.executable, WORKER], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_WS1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_runs_with_forced_collectives():
    """bench.py --gpus 1 with every collective of the step going through RCCL (async all-gather of the real
    features overlapping the generator forward, bucketed gradient all-reduce under the backward pass)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--no_cpu_baseline", "--no_secondary", "--batch_per_gpu", "64", "--nr_sinkhorn_iter", "20"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"].get("collectives") == "forced (RCCL, world size 1)"
