#!/usr/bin/env python3
"""Per-kernel-name totals over the LAST window_ms of a rocprofv3 (rocpd sqlite) kernel trace -- the timed steps of
bench.py, without model construction and warm-up.  (dev tool)
usage: tools/rocpd_window_stats.py <results.db> <window_ms> [steps in the window, for per-step figures]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[-1][2] - float(sys.argv[2]) * 1e6
rows = [r for r in rows if r[1] >= t0]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"window {float(sys.argv[2]):.1f} ms, kernel time {tot/1e6:.2f} ms, {len(rows)} launches")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:70s} n={c:5d} avg={t/c/1e3:9.1f}us  per step {t/1e6/steps:7.3f} ms  {100*t/tot:5.1f}%")
