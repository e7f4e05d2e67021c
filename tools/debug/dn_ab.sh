# usage: dn_ab.sh "ENV=1" "ENV2=1" ...   runs the DenseNet bench once per argument (plus a baseline first and last)
B="python bench.py --model densenet --nr_sinkhorn_iter 200 --steps 12 --warmup 6 --no_cpu_baseline --no_prof --no_secondary"
for v in "X=0" "$@" "X=0"; do
  env $v timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done
