#!/usr/bin/env python3
"""N = 128 cost kernel A/B (dev tool, GPU box): the fp32-MFMA kernel against cost128_h2_kernel at several split counts.
   python tools/exp/cost128_ab.py            -> runs itself once per variant (the switches are read once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    from otgan_amd import _lib
    from otgan_amd.utils import matching
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    N, D, L = 128, 32768, 100
    c = torch.rand(32, D, device=dev, generator=g)
    def feats(shift):
        f = (c[torch.randint(0, 32, (2 * N,), device=dev, generator=g)] + shift + 0.1 * torch.randn(2 * N, D, device=dev, generator=g)).abs()
        return torch.nn.functional.normalize(f, dim=1)
    fa, fb = feats(0.0), torch.nn.functional.normalize(feats(0.5) ** 2, dim=1)
    for need_b in (False, True):
        for _ in range(3):
            out = matching.matched_feature_grads(fa, fb, 500.0, L, need_b=need_b)
        torch.cuda.synchronize()
        _lib.prof_reset(); _lib.prof_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            out = matching.matched_feature_grads(fa, fb, 500.0, L, need_b=need_b)
        e1.record(); torch.cuda.synchronize()
        p = _lib.prof_collect(); _lib.prof_enable(False)
        print(f"  need_b={need_b}: call {e0.elapsed_time(e1) / reps * 1e3:7.1f} us | cost {p['cost_gemm']['ms'] / reps * 1e3:6.1f} us "
              f"sinkhorn {p['sinkhorn']['ms'] / reps * 1e3:6.1f} apply {p['plan_apply']['ms'] / reps * 1e3:6.1f}", flush=True)
    ga = out[0] if isinstance(out, (tuple, list)) else out
    torch.save(ga.detach().cpu(), sys.argv[2])
    sys.exit(0)
variants = [("fp32 MFMA kernels (OTGAN_MATCH_FP32=1)", {"OTGAN_MATCH_FP32": "1"}), ("two fp16 pieces (default)", {})]
import torch
ref = None
for i, (name, env) in enumerate(variants):
    fn = f"/tmp/cost128_{i}.pt"
    print(name, flush=True)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", fn], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout, r.stderr[-2000:] if r.returncode else "", flush=True)
    if r.returncode == 0:
        t = torch.load(fn)
        if ref is None: ref = t
        else: print(f"  rel. L2 difference of the generated-side gradient to the fp32 kernel's: {float((t - ref).norm() / ref.norm()):.3e}", flush=True)
