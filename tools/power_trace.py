#!/usr/bin/env python3
"""Socket power / gfx clock trace at >= 20 Hz under the Winograd-domain GEMM and under the training step (VERDICT r5 item 6:
"power-bound" needs a trace, not observed clocks).

    python tools/power_trace.py OUTDIR

A sampler PROCESS (amdsmi, falling back to rocm_smi's C library) writes (t, watts, gfx MHz, mem MHz, temperature) rows while
this process runs, one after the other, with a marker row between phases:

    idle            nothing
    gemm_real       one heavy layer's forward pass in a loop (G.conv1 shape: input transform + 256x128-tile GEMM + output
                    transform; the GEMM is 70 % of it) on Gaussian operands
    gemm_zero       the same launches on all-zero operands (same instruction stream, no toggling in the multipliers)
    gemm_p3         (second invocation with OTGAN_WINO_PIECES=3: the 24-bit build, six MFMAs per product)
    step            bench-shaped DCGAN training steps, two streams (default)
    step_one        the same with OTGAN_SIDE_STREAM=0

and prints per phase: mean / p95 power, mean / min clock, the board's power cap, and the loop's time per iteration.
Phases that need another process environment (pieces, side stream) are separate invocations: `--phases`.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _sampler(path, stop, hz):
    rows = []
    period = 1.0 / hz
    src = None
    cap = None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        try:
            cap = amdsmi.amdsmi_get_power_cap_info(h)
        except Exception as e:      # noqa: BLE001
            cap = {"error": str(e)}

        def read():
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            p = m.get("current_socket_power")
            if p in (None, "N/A", 65535):
                p = m.get("average_socket_power")
            return (p, m.get("current_gfxclk"), m.get("current_uclk"), m.get("temperature_hotspot"),
                    m.get("average_gfx_activity"))
        read()
        src = "amdsmi.gpu_metrics"
    except Exception as e:      # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
        try:
            import ctypes
            L = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
            L.rsmi_init(0)

            def read():
                p = ctypes.c_uint64(0)
                t = ctypes.c_int(0)
                if L.rsmi_dev_power_get(0, ctypes.byref(p), ctypes.byref(t)) != 0:
                    L.rsmi_dev_power_ave_get(0, 0, ctypes.byref(p))

                class F(ctypes.Structure):
                    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32),
                                ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]
                f = F()
                clk = None
                if L.rsmi_dev_gpu_clk_freq_get(0, 0, ctypes.byref(f)) == 0 and f.current < 33:
                    clk = f.frequency[f.current] / 1e6
                return (p.value / 1e6, clk, None, None, None)
            read()
            src = "rocm_smi (amdsmi failed: " + err + ")"
        except Exception as e2:      # noqa: BLE001
            json.dump({"source": None, "error": err + " / " + str(e2), "rows": []}, open(path, "w"))
            return
    t0 = time.time()
    while not stop.is_set():
        t = time.time()
        try:
            rows.append((t,) + tuple(read()))
        except Exception:      # noqa: BLE001
            pass
        dt = period - (time.time() - t)
        if dt > 0:
            time.sleep(dt)
    json.dump({"source": src, "cap": cap, "hz": len(rows) / max(1e-9, time.time() - t0), "rows": rows}, open(path, "w"))


def _layer_loop(zero, seconds):
    import torch
    from otgan_amd import _lib, ops
    dev = torch.device("cuda:0")
    _lib.lib()
    B, H, C, Cout, k = 256, 8, 512, 512, 5            # G.conv1: folded upsampling layer, GEMMs 1024 x 2048 x 512 per frequency
    mk = torch.zeros if zero else torch.randn
    x = mk(B, H, H, C, device=dev)
    V = (mk(k, k, C, Cout, device=dev) * 0.05)
    if zero:
        V = V + 0.0
    g = torch.ones(Cout, device=dev)
    b = torch.zeros(Cout, device=dev)
    with torch.no_grad():
        for _ in range(5):
            ops.conv2d_op(x, V, g, b, stride=1, upsample=True, preact=ops.ACT[None])
        torch.cuda.synchronize()
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            for _ in range(50):
                ops.conv2d_op(x, V, g, b, stride=1, upsample=True, preact=ops.ACT[None])
            torch.cuda.synchronize()
            n += 50
        dt = time.time() - t0
    return {"iters": n, "ms_per_iter": 1e3 * dt / n, "t0": t0, "t1": t0 + dt}


def _step_loop(seconds):
    import torch
    from otgan_amd.trainer import OTGAN, default_args
    dev = torch.device("cuda:0")
    args = default_args(model="dcgan", batch_size=128, nr_gpu=2, sinkhorn_lambda=500.0, nr_sinkhorn_iter=100, seed=1)
    m = OTGAN(args, dev)
    x = torch.rand(m.nb, 32, 32, 3, device=dev) * 2 - 1
    for _ in range(12):
        m.step(x)
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        for _ in range(12):
            m.step(x)
        torch.cuda.synchronize()
        n += 12
    dt = time.time() - t0
    m.close()
    return {"iters": n, "ms_per_iter": 1e3 * dt / n, "t0": t0, "t1": t0 + dt}


def _stats(rows, t0, t1):
    sel = [r for r in rows if t0 + 0.5 <= r[0] <= t1 - 0.1]
    if not sel:
        return {"samples": 0}
    def col(i):
        v = [float(r[i]) for r in sel if isinstance(r[i], (int, float))]
        return v
    out = {"samples": len(sel)}
    for name, i in (("power_w", 1), ("gfx_mhz", 2), ("mem_mhz", 3), ("temp_c", 4), ("gfx_activity", 5)):
        v = sorted(col(i))
        if v:
            out[name] = {"mean": round(sum(v) / len(v), 1), "min": v[0], "p95": v[int(0.95 * (len(v) - 1))], "max": v[-1]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--phases", default="idle,gemm_real,gemm_zero,step")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--hz", type=float, default=25.0)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)
    tag = a.tag or "default"
    raw = os.path.join(a.outdir, f"power_raw_{tag}.json")
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    p = ctx.Process(target=_sampler, args=(raw, stop, a.hz))
    p.start()
    time.sleep(1.0)
    phases = {}
    for ph in a.phases.split(","):
        if ph == "idle":
            t0 = time.time()
            time.sleep(3.0)
            phases[ph] = {"t0": t0, "t1": time.time()}
        elif ph == "gemm_real":
            phases[ph] = _layer_loop(False, a.seconds)
        elif ph == "gemm_zero":
            phases[ph] = _layer_loop(True, a.seconds)
        elif ph == "step":
            phases[ph] = _step_loop(a.seconds)
        time.sleep(1.0)
    stop.set()
    p.join(20)
    d = json.load(open(raw))
    summary = {"tag": tag, "env": {k: v for k, v in os.environ.items() if k.startswith("OTGAN_")}, "source": d.get("source"),
               "cap": d.get("cap"), "sample_hz": round(d.get("hz", 0.0), 1), "error": d.get("error")}
    for ph, info in phases.items():
        s = _stats(d.get("rows", []), info["t0"], info["t1"])
        s.update({k: round(v, 4) for k, v in info.items() if k in ("ms_per_iter",)})
        s["iters"] = info.get("iters")
        summary[ph] = s
    # the trace itself, decimated to what a reader needs: t relative to the first sample, W, MHz
    rows = d.get("rows", [])
    if rows:
        tz = rows[0][0]
        with open(os.path.join(a.outdir, f"power_trace_{tag}.csv"), "w") as f:
            f.write("t_s,power_w,gfx_mhz,mem_mhz,temp_c,gfx_activity,phase\n")
            for r in rows:
                ph = next((n for n, i in phases.items() if i["t0"] <= r[0] <= i["t1"]), "")
                f.write(",".join("" if v is None else str(v) for v in ((round(r[0] - tz, 3),) + tuple(r[1:]))) + f",{ph}\n")
    os.remove(raw)
    json.dump(summary, open(os.path.join(a.outdir, f"power_summary_{tag}.json"), "w"), indent=1, default=str)
    print(json.dumps(summary, default=str))


if __name__ == "__main__":
    main()
