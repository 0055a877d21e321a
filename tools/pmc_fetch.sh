#!/bin/bash
# GPU box: HBM bytes per tick_kernel launch (FETCH_SIZE / WRITE_SIZE passes) under an environment setting.  usage: tools/pmc_fetch.sh OUT "VAR=VAL ..." [bench args]
out=$1; envs=$2; shift 2
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  env $envs rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out/$c.err || { tail -5 $out/$c.err; exit 1; }
done
python3 tools/pmc_summary.py --last 25 $out/*/p_counter_collection.csv | grep -A3 "tick_kernel"
