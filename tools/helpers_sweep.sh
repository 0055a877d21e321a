#!/bin/bash
# GPU box (diagnostic): 4096 envs with h helper waves in every 16-wave workgroup (16 - h env slots per workgroup).  usage: tools/helpers_sweep.sh OUT LIB
out=$1; export CAT_SIM_LIB=$2; : > $out
for round in 1 2; do
for h in 0 4 8 12; do
  CAT_HELPERS=$h timeout -k 10 300 python bench.py --map labyrinth --envs 4096 --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "helpers $h" >> $out || exit 1
done
done
cat $out
