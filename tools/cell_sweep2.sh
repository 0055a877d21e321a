#!/bin/bash
# GPU box: tick_kernel time against the spatial-hash cell size, shipped library.  usage: tools/cell_sweep2.sh OUT
out=$1; : > $out
for round in 1 2; do
for c in ${CELLS:-16 12 8 6}; do
  for a in "--map labyrinth --envs 4096" "--map agh-map --envs 4096"; do
    CAT_GRID_CELL=$c timeout -k 10 300 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "cell $c $a" >> $out || exit 1
  done
done
done
cat $out
