#!/bin/bash
# GPU box: A/B of kernel builds on the headline workload (+ agh-map): usage tools/ab_lab.sh OUT.log LIB1 LIB2 ...; a LIB may carry
# environment settings in front, separated by commas: "CAT_WAVES_PER_BLOCK=10,CAT_HELPERS=2,build/var/x.so"
out=$1; shift; : > $out
for round in 1 2; do
for spec in "$@"; do
  lib=${spec##*,}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec%,*}" | tr ',' ' ')
  for a in "--map labyrinth --envs 4096" "--map agh-map --envs 4096"; do
    env $envs CAT_SIM_LIB=$lib timeout -k 10 200 python bench.py $a --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "$spec $a" >> $out || echo "FAILED $spec $a" >> $out
  done
done
done
cat $out
