#!/bin/bash
# dev: per-kernel times of a bench run under rocprofv3 --kernel-trace --stats, kernels matching a pattern
#   usage: tools/debug/kernel_times_bench.sh <model> <grep pattern>     (environment passes through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/ktb_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --model $1 --steps 12 --warmup 6 --no_secondary --no_cpu_baseline --no_prof > /dev/null 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" "$2" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Name"])]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'  {r["Name"][:86]:86s} n={int(r["Calls"]):4d} avg={float(r["AverageNs"])/1e3:9.1f}us tot={float(r["TotalDurationNs"])/1e6:8.2f}ms')
PY
rm -rf $O
