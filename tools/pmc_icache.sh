#!/bin/bash
# GPU box: instruction- and scalar-cache counters of tick_kernel on the default bench workload (one counter group per pass).
# usage: tools/pmc_icache.sh OUTDIR [bench args]
out=${1:-gpurun_out/icache}; shift
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out/p$i.err || { tail -5 $out/p$i.err; exit 1; }
done
python3 tools/pmc_summary.py --last 25 $out/p*/p_counter_collection.csv | grep -A40 "tick_kernel"
