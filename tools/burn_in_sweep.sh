for b in 0 100 200 300 390 400 600; do
  python bench.py --burn-in $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "burn-in $b K=20"
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "default K=2000"
python bench.py --burn-in 0 --steps 400 --warmup 0 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "one whole episode K=400"
