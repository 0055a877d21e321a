#!/bin/bash
# GPU box: rocprofv3 kernel trace of the default bench command + PMC passes (one counter group per pass, with --kernel-trace only),
# summarised into profiles/<tag>_*.  The PMC passes run the command the driver runs (bench.py --steps 20 --warmup 5: 600 burn-in
# ticks first) and the summary averages the LAST 25 launches -- the running batch the bench line times; the HBM counters are
# collected a second time straight from the reset (--burn-in 0, the regime of rounds 1-2) for the from-reset figure.
# usage: tools/collect_profiles.sh r03
set -e
tag=${1:-r03}
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/trace.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $out/pmc$i.err
  echo "pmc pass $i done" >> $out/progress.txt
done
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/reset$i -o p -- python3 bench.py --steps 20 --warmup 5 --burn-in 0 --no-cpu-baseline --no-extras > /dev/null 2> $out/reset$i.err
done
mkdir -p profiles
head -4 $out/trace/t_kernel_stats.csv > profiles/${tag}_final_kernel_stats.csv
python3 tools/pmc_summary.py --last 25 $out/pmc*/p_counter_collection.csv > profiles/${tag}_final_pmc_summary.txt
python3 tools/pmc_summary.py $out/reset*/p_counter_collection.csv > profiles/${tag}_pmc_from_reset.txt
python3 tools/make_traffic_json.py $tag > /dev/null
python3 bench.py > $out/bench.json 2> $out/bench.err     # after the passes: the line replays the traffic figure just collected
grep -v amdgpu.ids $out/bench.json | tail -1 > profiles/${tag}_final_bench.json
grep -v amdgpu.ids $out/bench_under_rocprof.json | tail -1 > profiles/${tag}_final_bench_under_rocprof.json
cp profiles/${tag}_* $out/
cat profiles/${tag}_final_kernel_stats.csv
