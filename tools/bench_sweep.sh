#!/bin/bash
# usage: tools/bench_sweep.sh OUT.log  -- the four bench configurations quoted in DESIGN.md (GPU box)
out=$1; : > $out
for a in "--map labyrinth --envs 4096" "--map agh-map --envs 4096" "--map labyrinth --envs 32768" "--map mixed --envs 16384"; do
  timeout -k 10 200 python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "$a" >> $out || exit 1
done
cat $out
