#!/usr/bin/env python3
"""Combine the FETCH_SIZE / WRITE_SIZE / MFMA PMC passes of tools/pmc_bench.sh into a per-kernel-
class summary (JSON): launches, average HBM bytes per launch (FETCH_SIZE x2 on gfx950 for wide
streaming reads, see MI355X_MICROARCH.md section HBM; counters are in KiB), MFMA utilisation.
usage: tools/traffic_summary.py <fetch.db> <write.db> <mfma.db> <out.json>"""
import collections, json, re, sqlite3, sys


def cls_of(name):
    m = re.search(r"conv_igemm_kernel<GemmCfg<[^>]*>, (true|false), (\d), (\d)>", name)
    if m:
        return "conv_fwd" if m.group(2) == "0" else "conv_dgrad"
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
    if "conv_wgrad" in name:
        return "conv_wgrad"
    if "cost_partial_kernel" in name:
        return "cost_gemm"
    if "plan_apply_kernel" in name:
        return "plan_apply"
    if "sinkhorn_small_kernel" in name:
        return "sinkhorn"
    return None


def collect(db, counters):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    names, durs = {}, {}
    for did, kn, cn, val, dur in rows:
        per[did][cn] += val
        names[did] = kn
        durs[did] = dur
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for did, cs in per.items():
        c = cls_of(names[did])
        if not c:
            continue
        out[c]["launches"] += 1
        out[c]["dur_ns"] += durs[did]
        for k in counters:
            out[c][k] += cs.get(k, 0.0)
    return out


fetch = collect(sys.argv[1], ["FETCH_SIZE"])
write = collect(sys.argv[2], ["WRITE_SIZE"])
mfma = collect(sys.argv[3], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
res = {}
for c in sorted(set(fetch) | set(write)):
    n = fetch[c]["launches"] or 1
    rd = fetch[c]["FETCH_SIZE"] * 1024 * 2 / n       # KiB -> B, x2 gfx950 wide-read correction
    wr = write[c]["WRITE_SIZE"] * 1024 / (write[c]["launches"] or 1)
    e = {"launches": int(n), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
         "hbm_bytes_per_launch": rd + wr}
    if c in mfma and mfma[c]["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
        cyc = mfma[c]["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_util"] = mfma[c]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        e["clock_ghz"] = cyc / mfma[c]["dur_ns"] if mfma[c]["dur_ns"] else None
    res[c] = e
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))
