#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per GEMM launch shape of the per-layer bench (dev tool, GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/lyf -- python $R/tools/bench_layers.py 256 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/lyw -- python $R/tools/bench_layers.py 256 > /dev/null 2>&1
f=$(find $R/gpurun_out/lyf -name "*.db" | head -1); w=$(find $R/gpurun_out/lyw -name "*.db" | head -1)
python - "$f" "$w" <<'PY'
import sqlite3, sys, collections
def load(db, ctr):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, counter_name, value, duration from counters_collection").fetchall()
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])
    seen = {}
    for did, kn, gx, gy, gz, wx, cn, val, dur in rows:
        if "bgemm_x3" not in kn or cn != ctr: continue
        key = ("TL" if "true, true" in kn else "NT", gx // wx, gy, gz)
        e = per[key]
        e[0] += val
        if did not in seen:
            seen[did] = 1; e[1] += dur; e[2] += 1
    return per
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
for k in sorted(f, key=lambda k: -f[k][1]):
    n = f[k][2]
    rd = f[k][0] * 1024 * 2 / n / 1e6
    wr = w[k][0] * 1024 / max(w[k][2], 1) / 1e6 if k in w else 0
    us = f[k][1] / n / 1e3
    print(f"{k} n={n} avg={us:7.1f}us rd={rd:8.1f}MB wr={wr:8.1f}MB {(rd+wr)/us:6.2f} TB/s")
PY
rm -rf $R/gpurun_out/lyf $R/gpurun_out/lyw
