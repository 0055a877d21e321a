#!/bin/bash
# GPU box: A/B of one environment switch of the shipped library on chosen workloads.  usage: tools/ab_env.sh OUT VAR "V1 V2 ..." "ARGS;ARGS;..."
out=$1; var=$2; vals=$3; IFS=';' read -ra args <<< "$4"; : > $out
for round in 1 2; do
for v in $vals; do
  for a in "${args[@]}"; do
    env $var=$v timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$var=$v $a" >> $out || exit 1
  done
done
done
cat $out
