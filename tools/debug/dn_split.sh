B="python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary"
for v in 1 0; do
  OTGAN_DENSE_SPLIT=$v timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v', d['value'], d['ms_per_step'])"
done
