"""The Sinkhorn kernels run their sweeps in two forms (csrc/sinkhorn.hip, round 4): log-domain until the potentials have
settled, then linear scaling of E = exp(K + f + g) without exponentials, folding back into the log-domain form whenever a
scaling factor leaves its band.  The matching tests of tests/test_matching_gpu.py / test_matching_grad_gpu.py -- plans,
losses and gradients against the fp64 oracle and the golden vectors, N = 8 ... 1024 -- must hold in every regime:

  * OTGAN_SINKHORN_LINEAR=0        every sweep log-domain (the kernels of rounds 1 - 3),
  * OTGAN_SINKHORN_LIN_RANGE=0.01  band of +-1 %: nearly every linear sweep ends in a fold-back (the path that otherwise
                                   only runs when potentials move by more than 20 nats after they had settled),
  * OTGAN_SINKHORN_SETTLE=0.05     the linear form is entered late (potentials within 0.05 nats per sweep).

The switches are read once per process, hence the subprocesses (reference utils/matching.py:50-57)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"OTGAN_SINKHORN_LINEAR": "0"}, {"OTGAN_SINKHORN_LIN_RANGE": "0.01"},
                                 {"OTGAN_SINKHORN_SETTLE": "0.05"}],
                         ids=["log_only", "fold_back", "late_entry"])
def test_matching_suite_in_every_sweep_regime(env):
    e = dict(os.environ)
    e.update(env)
    # the fold-back regime runs both files; the other two the operator-level file only (the whole GPU suite stays under
    # eight minutes)
    files = ["tests/test_matching_gpu.py"] + (["tests/test_matching_grad_gpu.py"] if "OTGAN_SINKHORN_LIN_RANGE" in env else [])
    sel = [] if "OTGAN_SINKHORN_LIN_RANGE" in env else ["-k", "sinkhorn or golden or full_size or iteration or equivariance"]
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", *sel],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
