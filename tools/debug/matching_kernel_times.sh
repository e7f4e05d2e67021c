#!/bin/bash
# dev: per-kernel times of the matching workload (tools/matching_workload.py) under rocprofv3 --kernel-trace --stats
#   usage: tools/debug/matching_kernel_times.sh N D L rows mode   (environment passes through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/mkt_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/matching_workload.py "$@" > /dev/null 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    if "at::" in r["Name"]: continue
    print(f'  {r["Name"][:70]:70s} n={int(r["Calls"]):4d} avg={float(r["AverageNs"])/1e3:9.1f}us tot={float(r["TotalDurationNs"])/1e6:8.2f}ms')
PY
rm -rf $O
