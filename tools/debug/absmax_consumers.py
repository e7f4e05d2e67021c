#!/usr/bin/env python3
"""dev: which kernels follow the library's own `absmax_kernel` launches (= which passes still reduce a tensor because no producer
left its amax record)?   usage: tools/debug/absmax_consumers.py <results.db> [pattern]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "absmax_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
grid = [c for c in cols if "grid" in c.lower()]
q = "select name, start, queue_id" + ("".join(", " + g for g in grid)) + " from kernels order by start"
try:
    rows = db.execute(q).fetchall()
except sqlite3.OperationalError:
    rows = db.execute("select name, start, 0 from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::|wino_p2::|void ", "", n)[:70]
cnt = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev = next((rows[j] for j in range(i - 1, -1, -1) if rows[j][2] == r[2] and pat not in rows[j][0]), None)
        nxt = next((rows[j] for j in range(i + 1, len(rows)) if rows[j][2] == r[2] and pat not in rows[j][0]), None)
        cnt[(short(prev[0]) if prev else "-", tuple(r[3:]), short(nxt[0]) if nxt else "-")] += 1
for (p, g, n), c in cnt.most_common(40):
    print(f"{c:5d}  grid={g}  after [{p}]  before [{n}]")
