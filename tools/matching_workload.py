#!/usr/bin/env python3
"""One matching problem, called a few times (workload for the PMC passes of tools/pmc_matching.sh).
usage: tools/matching_workload.py N D L [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otgan_amd import _lib
from otgan_amd.utils import matching
N, D, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
_lib.lib()
g = torch.Generator(device=dev).manual_seed(3)
def feats(shift):
    c = torch.rand(32, D, device=dev, generator=g) + shift
    f = (c[torch.randint(0, 32, (2 * N,), device=dev, generator=g)] + 0.1 * torch.randn(2 * N, D, device=dev, generator=g)).abs()
    return list(torch.chunk(torch.nn.functional.normalize(f, dim=1), 2, 0))
fa, fb = feats(0.0), feats(0.5)
for _ in range(4):
    if rows is None:
        out = matching.get_matched_features(fa, fb, 500.0, L)
    else:
        out = matching.get_matched_features_rows(fa, fb, 500.0, L, 0, rows)
torch.cuda.synchronize()
print("ok")
