#!/bin/bash
# GPU box: rocprofv3 kernel trace of the default bench command + PMC passes (one counter group per pass, with
# --kernel-trace only; 25 launches from the reset, as in round 1 -- no burn-in: the running batch reads a third more, DESIGN.md
# section 4), summarised into profiles/<tag>_*.  usage: tools/collect_profiles.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/trace.err
python3 bench.py > $out/bench.json 2> $out/bench.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 bench.py --steps 20 --warmup 5 --burn-in 0 --no-cpu-baseline --no-extras > /dev/null 2> $out/pmc$i.err
done
mkdir -p profiles
head -4 $out/trace/t_kernel_stats.csv > profiles/${tag}_final_kernel_stats.csv
python3 tools/pmc_summary.py $out/pmc*/p_counter_collection.csv > profiles/${tag}_final_pmc_summary.txt
python3 tools/make_traffic_json.py $tag > /dev/null
grep -v amdgpu.ids $out/bench.json | tail -1 > profiles/${tag}_final_bench.json
grep -v amdgpu.ids $out/bench_under_rocprof.json | tail -1 > profiles/${tag}_final_bench_under_rocprof.json
cp profiles/${tag}_final_* $out/
cat profiles/${tag}_final_kernel_stats.csv
