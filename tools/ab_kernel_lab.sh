#!/bin/bash
# GPU box: like ab_kernel.sh, labyrinth 4096 only (diagnostic builds).  usage: tools/ab_kernel_lab.sh OUT.log LIB...
out=$1; shift; : > $out
for round in 1 2; do
for lib in "$@"; do
  CAT_SIM_LIB=$lib timeout -k 10 200 python bench.py --map labyrinth --envs 4096 --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$(basename $lib)" >> $out || exit 1
done
done
cat $out
