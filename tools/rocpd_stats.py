#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into a text/CSV table:
per kernel name: calls, total ms, average us, % of GPU kernel time.  (dev tool)
usage: tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["name,calls,total_ms,avg_us,min_us,max_us,percent"]
for n, c, s, a, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n.replace(",", ";")
    lines.append(f"\"{n[:150]}\",{c},{s/1e6:.3f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*s/tot:.2f}")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
