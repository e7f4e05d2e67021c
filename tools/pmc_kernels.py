#!/usr/bin/env python3
"""Per-kernel-NAME table from the PMC passes of tools/pmc_bench.sh: launches, average duration, HBM bytes read
(FETCH_SIZE x 2: gfx950 correction for wide streaming reads, MI355X_MICROARCH.md; counters are KiB) and written
(WRITE_SIZE) per launch, achieved GB/s over the launch duration, MFMA utilisation.  Durations come from the
kernel-trace pass (un-perturbed by counters) when given.
usage: tools/pmc_kernels.py <fetch.db> <write.db> <mfma.db> <trace.db|-> <out.json> [min_total_ms]"""
import collections, json, re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"GemmCfg<(\d), (\d), (\d), (\d), (\d+)(?:, (\d+))?>", lambda m: "Cfg" + "".join(g for g in m.groups() if g), n)
    return n[:90]


def pmc(db, names):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    per, kn, du = collections.defaultdict(lambda: collections.defaultdict(float)), {}, {}
    for did, k, c, v, d in rows:
        per[did][c] += v
        kn[did], du[did] = short(k), d
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for did, cs in per.items():
        o = out[kn[did]]
        o["n"] += 1
        o["dur"] += du[did]
        for c in names:
            o[c] += cs.get(c, 0.0)
    return out


def trace(db):
    con = sqlite3.connect(db)
    out = collections.defaultdict(lambda: [0, 0.0])
    for n, c, s in con.execute("select name, count(*), sum(duration) from kernels group by name"):
        o = out[short(n)]
        o[0] += c
        o[1] += s
    return out


f = pmc(sys.argv[1], ["FETCH_SIZE"])
w = pmc(sys.argv[2], ["WRITE_SIZE"])
m = pmc(sys.argv[3], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
t = trace(sys.argv[4]) if sys.argv[4] != "-" else None
min_ms = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
res = {}
for k in sorted(set(f) | set(w)):
    n = f[k]["n"] or w[k]["n"]
    rd = f[k]["FETCH_SIZE"] * 1024 * 2 / max(f[k]["n"], 1)
    wr = w[k]["WRITE_SIZE"] * 1024 / max(w[k]["n"], 1)
    if t is not None and k in t and t[k][0]:
        dur_ns = t[k][1] / t[k][0]
        total_ms = t[k][1] / 1e6
        launches = t[k][0]
    else:
        dur_ns = f[k]["dur"] / max(f[k]["n"], 1)
        total_ms = f[k]["dur"] / 1e6
        launches = int(n)
    if total_ms < min_ms:
        continue
    e = {"launches": int(launches), "avg_us": round(dur_ns / 1e3, 2), "total_ms": round(total_ms, 3),
         "hbm_read_MB_per_launch": round(rd / 1e6, 3), "hbm_write_MB_per_launch": round(wr / 1e6, 3),
         "hbm_GBps": round((rd + wr) / dur_ns, 1) if dur_ns else None}
    if k in m and m[k]["GRBM_GUI_ACTIVE"] > 0:
        cyc = m[k]["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_util"] = round(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
    res[k] = e
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_ms"]))
json.dump(res, open(sys.argv[5], "w"), indent=1)
for k, e in res.items():
    print(f"{k:90s} n={e['launches']:4d} avg={e['avg_us']:9.1f}us tot={e['total_ms']:8.2f}ms rd={e['hbm_read_MB_per_launch']:9.1f}MB wr={e['hbm_write_MB_per_launch']:9.1f}MB "
          f"{e['hbm_GBps'] or 0:7.0f} GB/s mfma={e.get('mfma_util', 0):.3f}")
