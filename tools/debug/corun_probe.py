"""dev: instrumented victims beside the REAL Winograd-domain GEMM (tools/debug/corun_probe.hip).

    python tools/debug/corun_probe.py <mode> [iterations]
      rgbin0      round-2 packed-FMA RGB-in kernel as it was          -> mismatching launches vs its solo result
      rgbin1      same + scalar twin of every FMA, in-kernel compare, register / LDS / canary checks at the end
      canary      LDS address pattern re-read in a loop (8 KiB and 14 KiB per workgroup)
      reduce      out = a + b staged through LDS (a collective's reduce kernel in shape), checked against torch
      form0..5    asm-pinned accumulate chains: 0 v_pk_fma_f32 + op_sel, x from LDS; 1 same, registers only; 2 v_pk_fma_f32
                  without op_sel; 3 v_pk_mul_f32 + v_pk_add_f32; 4 v_fma_f64; 5 scalar v_fma_f32 (control)
    Every mode also checks the GEMM layers of stream A bit for bit against their solo results (GEMM as the victim).
    OTGAN_LIB_PATH selects a differently built libotgan_hip.so (tools/debug/build_corun_variants.sh)."""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from otgan_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
_lib.lib()
P = ctypes.CDLL(os.path.join(ROOT, "tools", "debug", "bin", "libcorun_probe.so"))
vp, ci, cu, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_long
P.probe_rgbin.argtypes = [ci, vp, vp, vp, vp, cu, ci, ci, ci, vp]
P.probe_canary.argtypes = [vp, cu, ci, ci, ci, vp]
P.probe_reduce.argtypes = [vp, vp, vp, cl, ci, vp]
P.probe_forms.argtypes = [ci, vp, vp, ci, vp, ci, vp]

mode = sys.argv[1] if len(sys.argv) > 1 else "rgbin0"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator().manual_seed(1)


def params(k, cin, cout):
    return ((torch.randn(k, k, cin, cout, generator=g) * 0.05).to(dev), torch.ones(cout, device=dev), torch.zeros(cout, device=dev))


# the co-runner of corun_repro.py: strided critic layers at batch 4 (tiny M, K = 4096 / 2048: long GEMMs, few workgroups)
xa = torch.randn(4, 8, 8, 512, generator=g).to(dev)
Va, ga, ba = params(5, 1024, 1024)
xa2 = torch.randn(4, 16, 16, 256, generator=g).to(dev)
Va2, ga2, ba2 = params(5, 512, 512)


def corun():
    return (ops.conv2d_op(xa, Va, ga, ba, stride=2, preact=ops.ACT["crelu"]),
            ops.conv2d_op(xa2, Va2, ga2, ba2, stride=2, preact=ops.ACT["crelu"]))


CAP = 4096
dbg = torch.zeros(16 + 16 * CAP, dtype=torch.int32, device=dev)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


NV = 16    # victim images (grid = NV * 4 workgroups)
xv = (torch.rand(NV, 32, 32, 3, generator=g) * 2 - 1).to(dev)
wT = (torch.randn(128, 75, generator=g) * 0.05).to(dev)
ra = torch.randn(1 << 22, generator=g).to(dev)
rb = torch.randn(1 << 22, generator=g).to(dev)


def victim():
    if mode in ("rgbin0", "rgbin1"):
        y = torch.empty(NV, 32, 32, 128, device=dev)
        rc = P.probe_rgbin(int(mode[-1]), xv.data_ptr(), wT.data_ptr(), y.data_ptr(), dbg.data_ptr(), CAP, NV, 32, 32, stream_ptr())
        assert rc == 0, rc
        return y
    if mode == "canary":
        for lds in (8192, 14336):
            rc = P.probe_canary(dbg.data_ptr(), CAP, lds, 40, 1024, stream_ptr())
            assert rc == 0, rc
        return torch.zeros(1, device=dev)
    if mode == "reduce":
        out = torch.empty_like(ra)
        rc = P.probe_reduce(ra.data_ptr(), rb.data_ptr(), out.data_ptr(), ra.numel(), 1024, stream_ptr())
        assert rc == 0, rc
        return out
    if mode.startswith("form"):
        out = torch.empty(FB * 256 * 32, device=dev)
        rc = P.probe_forms(int(mode[4:]), fw.data_ptr(), fx.data_ptr(), 400, out.data_ptr(), FB, stream_ptr())
        assert rc == 0, rc
        return out.view(torch.int32).view(FB * 4, 64, 16, 2)     # [wave, lane, accumulator, half] (bit patterns: NaN-proof compare)
    raise SystemExit("unknown mode " + mode)


FB = 512
fw = (torch.randn(8192, generator=g) * 0.1).to(dev)
fx = (torch.rand(256, generator=g) - 0.5).to(dev)


def hw(v):   # HW_REG_HW_ID of gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (as far as the fields go on gfx950)
    return dict(wave=v & 15, simd=(v >> 4) & 3, cu=(v >> 8) & 15, sh=(v >> 12) & 1, se=(v >> 13) & 7)


with torch.no_grad():
    gref = [t.clone() for t in corun()]
    ref = victim().clone()
    torch.cuda.synchronize()
    if mode == "reduce":
        assert torch.equal(ref, ra + rb), "reduce kernel wrong on an idle GPU"
    solo = int(dbg[0])
    print("records on an idle GPU:", solo, flush=True)
    dbg.zero_()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = gbad = 0
    lanes, elems, accs = collections.Counter(), collections.Counter(), collections.Counter()
    for it in range(iters):
        with torch.cuda.stream(sa):
            gouts = [corun() for _ in range(3)]
        with torch.cuda.stream(sb):
            outs = [victim() for _ in range(12)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                d = (o != ref).nonzero()
                for idx in d[:512].tolist():
                    if mode.startswith("form"):
                        lanes[idx[1] >> 4] += 1
                        elems[idx[3]] += 1
                        accs[idx[2]] += 1
                        continue
                    c = idx[-1]
                    lanes[(c % 64) >> 4] += 1
                    elems[c // 64] += 1
                if bad <= 3:
                    print(f"iter {it}: {d.shape[0]} elements differ, first {d[0].tolist()} last {d[-1].tolist()}", flush=True)
        for go in gouts:
            for t, r in zip(go, gref):
                if not torch.equal(t, r):
                    gbad += 1
                    if gbad <= 3:
                        print(f"iter {it}: GEMM layer output differs in {int((t != r).sum())} elements", flush=True)
    n = int(dbg[0])
    print(f"PROBE {mode} lib={os.path.basename(os.environ.get('OTGAN_LIB_PATH', 'default'))}: victim mismatches {bad} of {iters * 12}; "
          f"GEMM-as-victim mismatches {gbad} of {iters * 6}; debug records {n}", flush=True)
    if bad:
        print("  differing elements by lane quarter:", sorted(lanes.items()), " by half (0 = low / channel co, 1 = high / co + 64):", sorted(elems.items()))
    if accs:
        print("  by accumulator index (forms 0 / 1: even = op_sel_hi:[1,0,1], odd = op_sel:[0,1,0]):", sorted(accs.items()))
    recs = dbg[16:16 + 16 * min(n, CAP)].view(-1, 16).cpu().numpy().astype("uint32")
    kinds = collections.Counter(int(r[0]) for r in recs)
    print("  record kinds (1 packed != scalar twin, 2 weight register != fresh load, 3 LDS tile / canary != source, 4 canary kernel):", dict(kinds))
    import numpy as np
    shown = collections.Counter()
    for r in recs:
        k = int(r[0])
        shown[k] += 1
        if shown[k] > 12:
            continue
        f = lambda u: float(np.uint32(u).view(np.float32))
        if k == 1:
            print(f"  [1] block {r[1]} tid {r[2]} (lane {r[2] & 63}) pixel pair {r[3]} {hw(int(r[4]))} xcc {r[5] & 15}: "
                  f"acc0.lo packed {f(r[6]):.9g} scalar {f(r[7]):.9g} | acc0.hi {f(r[8]):.9g} / {f(r[9]):.9g} | acc1.lo {f(r[10]):.9g} / {f(r[11]):.9g} | "
                  f"acc1.hi {f(r[12]):.9g} / {f(r[13]):.9g} | recomputed acc0.lo packed {f(r[14]):.9g} scalar {f(r[15]):.9g}")
        elif k == 2:
            print(f"  [2] block {r[1]} tid {r[2]} (lane {r[2] & 63}) {hw(int(r[4]))}: {r[3]} weight registers differ from memory, first index {r[6]}")
        elif k == 3:
            print(f"  [3] block {r[1]} tid {r[2]} LDS float4 {r[3]} {hw(int(r[4]))}: got {[hex(int(x)) for x in r[6:14:2]]} want {[hex(int(x)) for x in r[7:14:2]]}")
        elif k == 4:
            print(f"  [4] block {r[1]} tid {r[2]} LDS word {r[3]} {hw(int(r[4]))}: got {hex(int(r[6]))} want {hex(int(r[7]))} at re-read {r[8]}")
    if recs.shape[0]:
        lq = collections.Counter(int(r[2] & 63) >> 4 for r in recs if r[0] in (1, 2))
        print("  kind 1/2 records by lane quarter:", sorted(lq.items()))
