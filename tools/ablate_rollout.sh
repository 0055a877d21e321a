#!/bin/bash
# GPU box: dynamic instruction counts of rollout_kernel per phase, by difference between diagnostic builds with a phase compiled out
# (build them first: for v in full nofan nofan_nophys nofan_nophys_nofront nofan_nophys_nofront_nowb; tools/build_variant.sh <prefix>_$v -DCAT_QUICK_BUILD -D...).
# usage: tools/ablate_rollout.sh [workload] [T] [prefix]      prefix: abl (default) -- or the name a set of builds of another source was given
wl=${1:-lab}; T=${2:-64}; pre=${3:-abl}
out=gpurun_out/ablate_${pre}_$wl; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CAT_RB_WORKLOADS=$wl
for v in full nofan nofan_nophys nofan_nophys_nofront nofan_nophys_nofront_nowb; do
  for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    CAT_SIM_LIB=build/var/${pre}_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$v.$(echo $c | cut -c4-8) -o p -- python3 tools/rollout_bench.py $T > /dev/null 2> $out/$v.err || { echo "$v failed"; tail -3 $out/$v.err; }
  done
  echo "== $v" >> $out/summary.txt
  python3 tools/pmc_summary.py --last 4 --only rollout $out/$v.*/p_counter_collection.csv >> $out/summary.txt
done
cat $out/summary.txt
