#!/usr/bin/env python3
"""Per (kernel, grid) breakdown of a rocprofv3 rocpd kernel trace (dev tool).
usage: tools/rocpd_shapes.py <results.db> [top_n]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration) "
                  "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
tot = sum(r[6] for r in rows)
print(f"total kernel ms {tot/1e6:.2f}")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:90]
for n, gx, gy, gz, wx, c, s, a in rows[:top]:
    print(f"{100*s/tot:5.1f}% {s/1e6:8.2f}ms n={c:5d} avg={a/1e3:8.1f}us blocks=({gx//wx},{gy},{gz}) {short(n)}")
