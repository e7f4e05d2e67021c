#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace (rocpd sqlite) by the kernel CLASSES bench.py times with HIP
events (conv_fwd / conv_dgrad / conv_wgrad / cost_gemm / sinkhorn / plan_apply), so the
`roofline.avg_ms` of bench.py can be checked against rocprof.  usage: class_stats.py <db> [out.json]"""
import json, re, sqlite3, sys


def cls_of(name):
    m = re.search(r"conv_igemm_kernel<GemmCfg<[^>]*>, (true|false), (\d), (\d)>", name)
    if m:
        return "conv_fwd" if m.group(2) == "0" else "conv_dgrad"
    if "conv_fewout" in name or "conv_rgbin" in name:          # (VALU / fp32-matrix-pipe kernels of the RGB layers)
        return "conv_few_channels(fwd|dgrad)"
    if "wino_bgemm_x3_kernel" in name or "wino_bgemm_x3n_kernel" in name or "wino_bgemm_x3_stream_kernel" in name:
        return "wino_gemm_bf16x3"
    if "wino_bgemm_kernel" in name:
        return "wino_gemm"
    if "dense16_fwd" in name:
        return "conv_fwd"
    if "dense16_dgrad" in name:
        return "conv_dgrad"
    if "dense16_wgrad" in name:
        return "conv_wgrad"
    if "conv_wgrad" in name or "conv_outer" in name:
        return "conv_wgrad"
    if "cost_partial_kernel" in name:
        return "cost_gemm"
    if "plan_apply_kernel" in name:
        return "plan_apply"
    if "sinkhorn_small_kernel" in name or "sinkhorn_panel_kernel" in name:
        return "sinkhorn"
    return None


db = sqlite3.connect(sys.argv[1])
out = {}
for name, dur in db.execute("select name, duration from kernels"):
    c = cls_of(name)
    if c:
        e = out.setdefault(c, {"launches": 0, "total_ms": 0.0})
        e["launches"] += 1
        e["total_ms"] += dur / 1e6
for e in out.values():
    e["avg_ms"] = e["total_ms"] / e["launches"]
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
