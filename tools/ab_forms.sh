#!/bin/bash
# GPU box: both forms of the ray fan on the light maps at the mixed batch's per-map share (16384 / 5 envs).  usage: tools/ab_forms.sh OUT
out=$1; : > $out
for m in squarinth grandbyrinth lbirinth labyrinth; do
  for f in groups chunks; do
    CAT_FAN=$f timeout -k 10 200 python bench.py --map $m --envs 3277 --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$m x3277 fan=$f" >> $out || exit 1
  done
done
cat $out
