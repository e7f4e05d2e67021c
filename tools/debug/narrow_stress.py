"""dev: soak test of the 256 x 128 tile GEMM kernel (wino_bgemm_x3n_kernel) -- the layers of tests/gemm_engines_worker.py
plus DCGAN-sized ones, evaluated REPS times in this process under OTGAN_X3_NARROW=1 and compared bit for bit with one
evaluation under OTGAN_X3_NARROW=0 (the variable is read per launch).  Run two copies at once to perturb timing:
    python tools/debug/narrow_stress.py 20 & python tools/debug/narrow_stress.py 20 & wait"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from otgan_amd import _lib, ops  # noqa: E402

CASES = [  # name, N, H, C, Cout, k, stride, upsample, preact
    ("up_8x8", 64, 8, 256, 256, 5, 1, True, None),
    ("up_4x4", 128, 4, 512, 512, 5, 1, True, None),
    ("s2_16x16", 64, 16, 128, 256, 5, 2, False, "crelu"),
    ("g_conv2", 256, 16, 256, 256, 5, 1, True, None),
    ("d_conv2", 256, 16, 256, 512, 5, 2, False, "crelu"),
    ("tiny", 4, 16, 128, 256, 5, 2, False, "crelu"),
    ("tiny_up", 4, 8, 256, 256, 5, 1, True, None),
]
# the layers of the DCGAN nets at 4 images per rank (tests/test_dist_gpu.py): OTGAN_STRESS_CASES=b4
B4_CASES = [
    ("g0_b4", 4, 4, 1024, 1024, 5, 1, True, None),
    ("g1_b4", 4, 8, 512, 512, 5, 1, True, None),
    ("g2_b4", 4, 16, 256, 256, 5, 1, True, None),
    ("d1_b4", 4, 32, 128, 256, 5, 2, False, "crelu"),
    ("d2_b4", 4, 16, 256, 512, 5, 2, False, "crelu"),
    ("d3_b4", 4, 8, 512, 1024, 5, 2, False, "crelu"),
    ("d1_b8", 8, 32, 128, 256, 5, 2, False, "crelu"),
    ("d3_b8", 8, 8, 512, 1024, 5, 2, False, "crelu"),
]
if os.environ.get("OTGAN_STRESS_CASES") == "b4":
    CASES = B4_CASES


_params = {}


def run(case, dev):
    name, N, H, C, Cout, k, s, up, pre = case
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre == "crelu" else 1
    x = torch.randn(N, H, H, C, generator=gen).to(dev).requires_grad_(True)
    if name not in _params:      # ONE parameter set per case: the weight cache keeps an entry (and its filters) per V
        _params.clear()
        ops._wcache.clear()
        _params[name] = ((torch.randn(k, k, C * mult, Cout, generator=gen) * 0.05).to(dev).requires_grad_(True),
                         torch.ones(Cout, device=dev, requires_grad=True), torch.zeros(Cout, device=dev, requires_grad=True))
    V, g, b = _params[name]
    y = ops.conv2d_op(x, V, g, b, stride=s, upsample=up, preact=ops.ACT[pre])
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(dev)
    dx, dV = torch.autograd.grad(y, [x, V], dy)
    return [t.detach().clone() for t in (y, dx, dV)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    _lib.lib()
    os.environ["OTGAN_X3_STREAM"] = "0"
    bad = 0
    for case in CASES:
        os.environ["OTGAN_X3_NARROW"] = "0"
        ref = run(case, dev)
        os.environ["OTGAN_X3_NARROW"] = "1"
        for r in range(reps):
            got = run(case, dev)
            for tag, a, b in zip(("y", "dx", "dV"), got, ref):
                if not torch.equal(a, b):
                    n = int((a != b).sum())
                    print(f"MISMATCH {case[0]}.{tag} rep {r}: {n} elements, max |diff| {float((a - b).abs().max()):.3g}", flush=True)
                    bad += 1
    print("NARROW_STRESS", "FAILED" if bad else "OK", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
