#!/bin/bash
# two-stream schedule of a step (extra arguments go to bench.py, e.g. --model densenet --nr_sinkhorn_iter 200) under the tracer: per queue busy time, union, idle gaps, and what runs alone (dev)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/so_trace
rocprofv3 --kernel-trace -d $R/gpurun_out/so_trace -- python $R/bench.py --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary "$@" > /dev/null 2>&1
t=$(find $R/gpurun_out/so_trace -name "*.db" | head -1)
python - "$t" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the last 12 steps: take the last 12/18 of the launches by time window: use Adam launches as step markers
adam = [i for i, r in enumerate(rows) if "adam" in r[0]]
marks = adam[-13:]
rows = rows[marks[0] + 1: marks[-1] + 1]
t0, t1 = rows[0][1], max(r[2] for r in rows)
print(f"window {(t1 - t0) / 1e6:.2f} ms for 12 steps = {(t1 - t0) / 12e6:.3f} ms/step, {len(rows)} launches")
byq = collections.defaultdict(float)
for n, s, e, q in rows: byq[q] += e - s
for q, v in sorted(byq.items(), key=lambda x: -x[1]): print(f"  queue {q}: busy {v / 12e6:.3f} ms/step")
ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
depth, last, hist = 0, t0, collections.defaultdict(float)
for t, d in ev:
    hist[min(depth, 3)] += t - last; last = t; depth += d
for k in sorted(hist): print(f"  {k} kernel(s) running: {hist[k] / 12e6:.3f} ms/step")
# which kernels run ALONE (depth 1) the most -> candidates to overlap
import re
alone = collections.defaultdict(float); tot = collections.defaultdict(float)
ivs = sorted((s, e, n, q) for n, s, e, q in rows)
bounds = sorted(set([s for s, e, n, q in ivs] + [e for s, e, n, q in ivs]))
import bisect
active = []
j = 0
cur = []
for a, b in zip(bounds[:-1], bounds[1:]):
    while j < len(ivs) and ivs[j][0] <= a:
        cur.append(ivs[j]); j += 1
    cur = [iv for iv in cur if iv[1] > a]
    if len(cur) == 1:
        alone[(re.sub(r"\(.*", "", cur[0][2])[:60], cur[0][3])] += b - a
for n, s, e, q in rows: tot[(re.sub(r"\(.*", "", n)[:60], q)] += e - s
print("  kernels by time running ALONE (ms/step; total ms/step; queue):")
for k, v in sorted(alone.items(), key=lambda x: -x[1])[:22]: print(f"    {v / 12e6:6.3f}  {tot[k] / 12e6:6.3f}  q{k[1]}  {k[0]}")
PY
rm -rf $R/gpurun_out/so_trace
