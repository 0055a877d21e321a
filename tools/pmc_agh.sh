#!/bin/bash
# GPU box: the VALU-side PMC passes for another bench workload (default: the agh-map shard).  usage: tools/pmc_agh.sh TAG [bench args]
set -e
tag=$1; shift
args=${*:-"--map agh-map --envs 4096"}
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $out/pmc$i.err
done
python3 tools/pmc_summary.py --last 25 $out/pmc*/p_counter_collection.csv > $out/summary.txt
cat $out/summary.txt
