#!/usr/bin/env python3
"""Per-dispatch PMC table from a rocprofv3 rocpd sqlite file, filtered by kernel-name regex.
usage: tools/rocpd_pmc.py <db> [regex]   (dev tool)"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
rx = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
rows = db.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, counter_name, value, duration "
                  "from counters_collection").fetchall()
d = collections.OrderedDict()
for did, kn, gx, gy, gz, wx, cn, val, dur in rows:
    if not rx.search(kn): continue
    short = re.sub(r"\(anonymous namespace\)::", "", kn)
    short = re.sub(r"GemmCfg<(\d), (\d), (\d), (\d), (\d+)>", r"Cfg\1\2\3\4_\5", short)[:60]
    e = d.setdefault(did, {"name": short, "grid": (gx // wx, gy, gz), "dur_us": dur / 1e3})
    e[cn] = e.get(cn, 0) + val
names = sorted({k for e in d.values() for k in e if k not in ("name", "grid", "dur_us")})
print("dispatch name grid dur_us " + " ".join(names))
for did, e in d.items():
    print(did, e["name"], e["grid"], f"{e['dur_us']:.1f}", " ".join(f"{e.get(n,0):.4g}" for n in names))
