#!/bin/bash
# kernel trace of a rank of eight's matching call (tools/exp/rank_ab.py): every launch of the last call with its grid and duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rk_trace
rocprofv3 --kernel-trace -d $R/gpurun_out/rk_trace -- python $R/tools/exp/rank_ab.py 32768 100 > /dev/null 2>&1
t=$(find $R/gpurun_out/rk_trace -name "*.db" | head -1)
python - "$t" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[-60:]
t0 = rows[0][1]
for n, s, e, gx, gy, gz, wx in rows:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  grid {gx // max(wx,1):6d} x {gy} x {gz}  {n[:70]}")
PY
rm -rf $R/gpurun_out/rk_trace
