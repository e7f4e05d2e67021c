#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 (rocpd sqlite) kernel trace: total idle time, the
distribution of gaps, and which kernel pairs the idle time sits between.  (dev tool)
usage: tools/rocpd_gaps.py <results.db> [top] [window_ms: only the last window_ms of the trace]"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = db.execute("select name, start, end from kernels order by start").fetchall()
if len(sys.argv) > 3:
    t0 = rows[-1][2] - float(sys.argv[3]) * 1e6
    rows = [r for r in rows if r[1] >= t0]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:48]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
prev_end, prev_name = rows[0][2], rows[0][0]
for n, s, e in rows[1:]:
    g = s - prev_end
    if g > 0:
        k = (short(prev_name), short(n))
        gaps[k][0] += g
        gaps[k][1] += 1
        hist[min(int(g / 1000).bit_length(), 12)] += g
    if e > prev_end:
        prev_end, prev_name = e, n
idle = sum(v[0] for v in gaps.values())
print(f"span {span/1e6:.2f} ms, kernels {busy/1e6:.2f} ms, idle between kernels {idle/1e6:.2f} ms ({100*idle/span:.1f} %)")
print("idle time by gap length (us): " + ", ".join(f"<{2**b}: {v/1e6:.2f} ms" for b, v in sorted(hist.items())))
for (a, b), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{g/1e6:7.3f} ms  n={c:4d}  avg {g/c/1e3:7.1f} us   {a}  ->  {b}")
