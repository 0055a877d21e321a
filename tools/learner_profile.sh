#!/bin/bash
# GPU box: kernel traces of the learner leg (collect only, collect + update) at the bench's three settings -> profiles/<tag>_learner_<setting>.txt
# usage: tools/learner_profile.sh r05
set -e
tag=${1:-r05}
out=gpurun_out/learner_$tag; mkdir -p $out profiles
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "h16 16 64" "h128 128 64" "r90 16 90"; do
  set -- $cfg; name=$1; H=$2; R=$3
  for mode in collect full; do
    rm -rf $out/${name}_$mode
    timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/${name}_$mode -o t -- python3 tools/learner_trace.py $mode $H $R 5 > $out/${name}_$mode.log 2>&1
    echo "$name $mode done" >> $out/progress.txt
  done
  python3 tools/learner_split.py "$name" $out/${name}_collect/t_kernel_trace.csv $out/${name}_full/t_kernel_trace.csv $H $R 5 3 > profiles/${tag}_learner_$name.txt
done
mkdir -p $out/profiles; cp profiles/${tag}_learner_* $out/profiles/
cat profiles/${tag}_learner_*.txt
