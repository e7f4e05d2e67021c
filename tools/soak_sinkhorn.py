#!/usr/bin/env python3
"""Soak of the two-form Sinkhorn sweeps under a MOVING feature distribution: a synthetic DCGAN run whose
critic and generator train, with the kernels' device-side sweep counters on (otgan_sinkhorn_counters): per step the number of
log-domain sweeps, linear sweeps, entries into the linear form and fold-backs (reference utils/matching.py:50-57: exactly L
sweeps per problem, whatever the form).

    python tools/soak_sinkhorn.py --steps 2000 --batch 64 --out gpurun_out/soak_default.json
    OTGAN_SINKHORN_LINEAR=0 python tools/soak_sinkhorn.py ... --out gpurun_out/soak_log_only.json
    python tools/soak_sinkhorn.py --compare gpurun_out/soak_default.json gpurun_out/soak_log_only.json
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(a):
    import torch
    from otgan_amd import _lib
    from otgan_amd.trainer import OTGAN, default_args
    dev = torch.device("cuda:0")
    _lib.lib()
    _lib.sinkhorn_counters(True)          # before any capture: the counters' address travels in the kernels' arguments
    args = default_args(model=a.model, batch_size=a.batch // 2, nr_gpu=2, sinkhorn_lambda=500.0, nr_sinkhorn_iter=a.iters,
                        nr_gen_per_disc=5, seed=1, step_graph=a.graph)
    m = OTGAN(args, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    # a small fixed "data set" of smooth images (so that the critic has something to separate) cycled in batches
    base = torch.rand(16, 4, 4, 3, device=dev, generator=g)
    data = torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=32, mode="bilinear").permute(0, 2, 3, 1)
    data = (data[torch.randint(0, 16, (a.batch * 8,), device=dev, generator=g)] +
            0.05 * torch.randn(a.batch * 8, 32, 32, 3, device=dev, generator=g)).clamp(0, 1) * 2 - 1
    torch.manual_seed(5)
    dists, ents, per_step, nan = [], [], [], 0
    tot = None
    for i in range(a.steps):
        x = data[(i % 8) * a.batch:(i % 8 + 1) * a.batch].contiguous()
        r = m.step(x)
        c = _lib.sinkhorn_counters_read(reset=True)
        d, e = float(r["distance"]), float(r["entropy"])
        if not (math.isfinite(d) and math.isfinite(e)):
            nan += 1
        dists.append(d)
        ents.append(e)
        per_step.append([c["log_sweeps"], c["linear_sweeps"], c["entries"], c["fold_backs"], c["never_entered"]])
        tot = c if tot is None else {k: tot[k] + c[k] for k in c}
    m.close()
    _lib.sinkhorn_counters(False)
    P = max(tot["problems"], 1)
    out = {"steps": a.steps, "batch": a.batch, "iters": a.iters, "model": a.model,
           "regime": {k: v for k, v in os.environ.items() if k.startswith("OTGAN_SINKHORN")},
           "step_graph": bool(a.graph), "side_stream": os.environ.get("OTGAN_SIDE_STREAM", "1") != "0" and not a.graph, "non_finite_steps": nan, "totals": tot,
           "sweeps_per_problem": (tot["log_sweeps"] + tot["linear_sweeps"]) / P,
           "log_sweeps_per_problem": tot["log_sweeps"] / P,
           "fold_backs_per_1000_problems": 1000.0 * tot["fold_backs"] / P,
           "mean_first_entry_sweep": tot["first_entry_sweep_sum"] / max(P - tot["never_entered"], 1),
           "never_entered_frac": tot["never_entered"] / P,
           "steps_with_fold_back": sum(1 for s in per_step if s[3]),
           "max_log_sweeps_in_a_step": max(s[0] for s in per_step),
           "distance": dists, "entropy": ents}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f)
    print(json.dumps({k: v for k, v in out.items() if k not in ("distance", "entropy")}))


def compare(fa, fb):
    A, B = json.load(open(fa)), json.load(open(fb))
    n = min(len(A["distance"]), len(B["distance"]))
    rel = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(A["distance"][:n], B["distance"][:n])]
    marks = [1, 2, 3, 5, 10, 20, 50, 100, 200, 500, 1000, 2000]
    print("relative difference of the distance trajectories (same seed):")
    for k in marks:
        if k <= n:
            print(f"  after {k:5d} steps: step {rel[k - 1]:.2e}   worst so far {max(rel[:k]):.2e}")
    last = slice(max(0, n - 200), n)
    ma, mb = sum(A["distance"][last]) / (last.stop - last.start), sum(B["distance"][last]) / (last.stop - last.start)
    print(f"mean distance over the last 200 steps: {ma:.6f} vs {mb:.6f} ({abs(ma - mb) / abs(mb):.2e} relative)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--graph", action="store_true", help="replay the steps as hipGraphs (trainer.GraphedSteps; disables the side stream)")
    ap.add_argument("--out", default="gpurun_out/soak.json")
    ap.add_argument("--compare", nargs=2)
    a = ap.parse_args()
    if a.compare:
        compare(*a.compare)
    else:
        run(a)
