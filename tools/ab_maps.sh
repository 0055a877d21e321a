#!/bin/bash
# GPU box: A/B of diagnostic builds on chosen workloads.  usage: tools/ab_maps.sh OUT "ARGS;ARGS;..." LIB...
out=$1; IFS=';' read -ra args <<< "$2"; shift 2; : > $out
for round in 1 2; do
for lib in "$@"; do
  for a in "${args[@]}"; do
    CAT_SIM_LIB=$lib timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$(basename $lib) $a" >> $out || exit 1
  done
done
done
cat $out
