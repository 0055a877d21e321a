#!/bin/bash
# GPU box: tick_kernel time against the workgroup size (waves = env slots per workgroup), shipped library.  usage: tools/wpb_sweep.sh OUT
out=$1; : > $out
for round in 1 2; do
for w in 16 8 4 2; do
  for a in "--map labyrinth --envs 4096" "--map agh-map --envs 4096"; do
    CAT_WAVES_PER_BLOCK=$w timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "wpb $w $a" >> $out || exit 1
  done
done
done
cat $out
