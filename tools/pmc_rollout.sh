#!/bin/bash
# GPU box: PMC passes over the resident rollout kernel (tools/rollout_bench.py, one workload, T ticks per launch).
# usage: tools/pmc_rollout.sh TAG WORKLOAD T     -> gpurun_out/pmc_roll_TAG/summary.txt
tag=${1:-x}; wl=${2:-lab}; T=${3:-64}
out=gpurun_out/pmc_roll_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CAT_RB_WORKLOADS=$wl
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 tools/rollout_bench.py $T > /dev/null 2> $out/pmc$i.err || exit 1
  echo "pmc pass $i done" >> $out/progress.txt
done
python3 tools/pmc_summary.py --last 4 $out/pmc*/p_counter_collection.csv > $out/summary.txt
cat $out/summary.txt
