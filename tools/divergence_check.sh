for k in 1 2 3 5 9; do
  a=$(python bench.py --steps $k --warmup 0 --no_cpu_baseline --no_prof | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['last_distance'])")
  b=$(OTGAN_DISABLE_WINOGRAD=1 python bench.py --steps $k --warmup 0 --no_cpu_baseline --no_prof | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['last_distance'])")
  echo "steps=$k winograd=$a direct=$b"
done
