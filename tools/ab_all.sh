#!/bin/bash
# GPU box: A/B of kernel builds on the four BASELINE shapes; specs as in tools/ab_lab.sh.  usage: tools/ab_all.sh OUT.log SPEC...
out=$1; shift; : > $out
for round in 1 2; do
for spec in "$@"; do
  lib=${spec##*,}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec%,*}" | tr ',' ' ')
  for a in "--map labyrinth --envs 4096" "--map agh-map --envs 4096" "--map grandbyrinth --cops 3 --thieves 2 --envs 8192" "--map mixed --envs 16384" "--map labyrinth --envs 4096 --rays 90"; do
    env $envs CAT_SIM_LIB=$lib timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$spec $a" >> $out || echo "FAILED $spec $a" >> $out
  done
done
done
cat $out
