for b in 1 2 4 8 16 24 32; do for l in 1 0 1 0; do
  st=4; [ $b -le 4 ] && st=8
  VV_BENCH_OPTIONS=lanes=$l python bench.py --batch $b --steps $st --warmup 2 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b lanes=$l', d['ms_per_step'], d['value'])"
done; done
