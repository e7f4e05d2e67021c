"""dev tool: relative L2 error against fp64 of three layers under the three GEMM engines (prints the table quoted in
DESIGN.md; the assertion lives in tests/test_gemm_engines_gpu.py)."""
import os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nets_torch as NT
from gemm_engines_worker import CASES
tmp = tempfile.mkdtemp()
runs = {}
for tag, env in (("fp16x2", {}), ("bf16x3", {"OTGAN_WINO_PIECES": "3"}), ("fp32", {"OTGAN_WINO_FP32": "1"})):
    path = os.path.join(tmp, tag + ".npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gemm_engines_worker.py"), path], check=True, env=dict(os.environ, **env))
    runs[tag] = dict(np.load(path))
for name, N, H, C, Cout, k, s, up, pre in CASES:
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre == "crelu" else 1
    x = torch.randn(N, H, H, C, generator=gen).double().requires_grad_(True)
    V = (torch.randn(k, k, C * mult, Cout, generator=gen) * 0.05).double().requires_grad_(True)
    y = NT.conv2d([x], {"V": V, "g": torch.ones(Cout, dtype=torch.float64), "b": torch.zeros(Cout, dtype=torch.float64)}, pre, s, up)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).double()
    dx, dV = torch.autograd.grad(y, [x, V], dy)
    for tag, ref in (("y", y), ("dx", dx), ("dV", dV)):
        ref = ref.detach().numpy()
        print(f"{name:10s} {tag:3s} " + "  ".join(f"{e} {np.linalg.norm(r[f'{name}.{tag}.0'] - ref) / np.linalg.norm(ref):.2e}" for e, r in runs.items()))
