"""Worker of tests/test_gemm_engines_gpu.py: runs three Winograd layers (forward, dgrad, wgrad) whose batched GEMMs
have few tiles per compute unit, under whatever engine switches the parent set, twice, and writes the results."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otgan_amd import _lib, ops  # noqa: E402

CASES = [  # name, N, H, C, Cout, k, stride, upsample, preact
    ("up_8x8", 64, 8, 256, 256, 5, 1, True, None),        # 256 tiles x (4 x 256) outputs: 4 GEMM tiles
    ("up_4x4", 128, 4, 512, 512, 5, 1, True, None),       # 128 tiles: one row tile, 8 column tiles
    ("s2_16x16", 64, 16, 128, 256, 5, 2, False, "crelu"),  # strided: absent (class, frequency) blocks, K runs
]


# layers on the implicit-GEMM engine (3x3 stride 2: no Winograd path), selected with OTGAN_WORKER_CASES=igemm
IGEMM_CASES = [
    ("s2_k3_144", 16, 32, 288, 144, 3, 2, False, "crelu"),   # the 128 x 160 forward tile, paired input gradient
    ("s2_k3_200", 16, 16, 200, 200, 3, 2, False, "crelu"),   # channel counts that are multiples of 4 only
    ("s2_k3_none", 8, 16, 64, 96, 3, 2, False, None),
]


def case_tensors(name, N, H, C, Cout, k, pre, out_shape=None):
    """(x0, V0, dy) of a case.  OTGAN_WORKER_DATA=step: operands shaped like the tensors of a real training step instead
    of Gaussians -- x = a * sigmoid(b) of wide Gaussians (what a GLU hands the next layer: heavy tail, most values near
    zero), dy = log-normally spread gradients around 1e-6 (the critic's last layers see 1e-7 .. 1e-5) with ONE element
    3e4 times the typical magnitude."""
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre == "crelu" else 1
    real = os.environ.get("OTGAN_WORKER_DATA") == "step"
    if real:
        x0 = (torch.randn(N, H, H, C, generator=gen) * 3) * torch.sigmoid(torch.randn(N, H, H, C, generator=gen) * 3)
    else:
        x0 = torch.randn(N, H, H, C, generator=gen)
    V0 = torch.randn(k, k, C * mult, Cout, generator=gen) * 0.05
    dy = None
    if out_shape is not None:
        g7 = torch.Generator().manual_seed(7)
        dy = torch.randn(out_shape, generator=g7)
        if real:
            dy = dy * torch.exp(2.0 * torch.randn(out_shape, generator=g7)) * 1e-6
            dy.view(-1)[dy.numel() // 3] = 3e-2
    return x0, V0, dy


def main(out):
    dev = torch.device("cuda:0")
    _lib.lib()
    res = {}
    for name, N, H, C, Cout, k, s, up, pre in (IGEMM_CASES if os.environ.get("OTGAN_WORKER_CASES") == "igemm" else CASES):
        x0, V0, _ = case_tensors(name, N, H, C, Cout, k, pre)
        for rep in range(2):
            x = x0.to(dev).requires_grad_(True)
            V = V0.to(dev).requires_grad_(True)
            g = torch.ones(Cout, device=dev, requires_grad=True)
            b = torch.zeros(Cout, device=dev, requires_grad=True)
            y = ops.conv2d_op(x, V, g, b, stride=s, upsample=up, preact=ops.ACT[pre])
            dy = case_tensors(name, N, H, C, Cout, k, pre, tuple(y.shape))[2].to(dev)
            dx, dV = torch.autograd.grad(y, [x, V], dy)
            for tag, t in (("y", y), ("dx", dx), ("dV", dV)):
                res[f"{name}.{tag}.{rep}"] = t.detach().cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
