B="python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary"
for v in 64 128 300 64; do
  OTGAN_PLAIN3_MIN_CEFF=$v timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_ceff=$v', d['value'], d['ms_per_step'])"
done
