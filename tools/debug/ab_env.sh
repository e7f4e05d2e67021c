#!/bin/bash
# dev: bench.py once per environment setting on ONE box (box-to-box spread is +-3-5 %), baseline first and last.
#   tools/debug/ab_env.sh "OTGAN_X3_NARROW=0" "OTGAN_X3_NARROW=1" ...   (extra bench flags in BENCH_FLAGS)
cd ${GRAFT_REPO_ROOT:-.}
for setting in "$@" "$1"; do
  line=$(env $setting python bench.py --steps 36 --warmup 6 --no_cpu_baseline --no_secondary ${BENCH_FLAGS} 2>/dev/null | grep '^{' | tail -1)
  python - "$setting" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline", {})
print(f"{sys.argv[1]:40s} {d['value']:9.1f} img/s  {d['ms_per_step']:7.3f} ms  gemm frac {r.get('frac')} avg_ms {r.get('avg_ms')} launches {r.get('launches')}")
PY
done
