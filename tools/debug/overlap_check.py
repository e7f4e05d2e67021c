"""dev: in a rocprofv3 kernel trace (rocpd sqlite) of bench.py, what ran while sinkhorn_small_kernel was running?
usage: overlap_check.py <results.db>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:60]
sk = [(s, e) for n, s, e in rows if "sinkhorn_small_kernel" in n][-6:]
for s0, e0 in sk:
    inside = [(short(n), (max(s, s0) - s0) / 1e3, (min(e, e0) - s0) / 1e3) for n, s, e in rows if s < e0 and e > s0 and "sinkhorn_small" not in n]
    print("sinkhorn %.1f us:" % ((e0 - s0) / 1e3), inside[:8])
fb = [(short(n), s, e) for n, s, e in rows if "filter_bwd" in n][-6:]
for n, s, e in fb:
    prev = [short(m) for m, s2, e2 in rows if s2 < e and e2 > s and m is not n and "filter_bwd" not in m]
    print(n, "%.1f us, overlapping:" % ((e - s) / 1e3), prev[:6])
