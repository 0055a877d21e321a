#!/bin/bash
# GPU box: A/B on the dense workloads.  usage: tools/ab_agh.sh OUT LIB...
out=$1; shift; : > $out
for round in 1 2; do
for lib in "$@"; do
  for a in "--map agh-map --envs 4096" "--map mixed --envs 16384"; do
    CAT_SIM_LIB=$lib timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$(basename $lib) $a" >> $out || exit 1
  done
done
done
cat $out
