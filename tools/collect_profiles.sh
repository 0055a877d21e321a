#!/bin/bash
# GPU box: per BASELINE shape, a rocprofv3 kernel trace + PMC passes (one counter group per pass, with --kernel-trace only) of the
# bench command the driver runs (--steps 20 --warmup 5 after the burn-in ticks) restricted to that shape (--shape-only: the
# one-launch-per-tick region + the resident-rollout leg of the SAME workload), summarised into profiles/<tag>_*:
#   <tag>_<shape>_kernel_stats.csv   trace: tick_kernel / rollout_kernel rows      <tag>_<shape>_pmc_summary.txt   counters
#   <tag>_traffic_<shape>.json       tick_kernel, last 25 launches                  <tag>_traffic_<shape>_rollout.json   rollout_kernel, last 4
# The headline shape (labyrinth 2v1 x4096) also keeps the names of the earlier rounds (<tag>_final_*, <tag>_traffic.json) and gets
# the HBM counters a second time straight from the reset (--burn-in 0).  rocprofv3 is given `python3 bench.py ...` directly.
# usage: tools/collect_profiles.sh r05 [shape ...]      shapes: lab agh 3v2 mixed r90 (default: all); run tools/stamp_head.sh HERE first (git head for the JSON files)
set -e
tag=${1:-r04}; shift || true
shapes=${@:-lab agh 3v2 mixed r90}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
declare -A ARGS KEY
ARGS[lab]="--map labyrinth --envs 4096";                 KEY[lab]='{"map": "labyrinth", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1}'
ARGS[agh]="--map agh-map --envs 4096";                   KEY[agh]='{"map": "agh-map", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1}'
ARGS[3v2]="--map grandbyrinth --cops 3 --thieves 2 --envs 8192"; KEY[3v2]='{"map": "grandbyrinth", "envs": 8192, "rays": 64, "cops": 3, "thieves": 2}'
ARGS[mixed]="--map mixed --envs 16384";                  KEY[mixed]='{"map": "mixed", "envs": 16384, "rays": 64, "cops": 2, "thieves": 1}'
ARGS[r90]="--map labyrinth --envs 4096 --rays 90";       KEY[r90]='{"map": "labyrinth", "envs": 4096, "rays": 90, "cops": 2, "thieves": 1}'
for sh in $shapes; do
  a="${ARGS[$sh]} --steps 20 --warmup 5 --shape-only"
  d=$out/$sh; rm -rf $d; mkdir -p $d
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python3 bench.py $a > $d/bench_under_rocprof.json 2> $d/trace.err
  echo "$sh trace done" >> $out/progress.txt
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/pmc$i -o p -- python3 bench.py $a > /dev/null 2> $d/pmc$i.err
    echo "$sh pmc pass $i done" >> $out/progress.txt
  done
  grep -E '"Name"|tick_kernel|step_kernel|rollout_kernel|reset_kernel' $d/trace/t_kernel_stats.csv > profiles/${tag}_${sh}_kernel_stats.csv
  { echo "# tick_kernel / step_kernel (whichever serves this sim's one-tick calls) / reset_kernel: the last 25 launches; rollout_kernel: the last 4 (64 ticks each)"; python3 tools/pmc_summary.py --last 25 --only tick_kernel,step_kernel,reset_kernel $d/pmc*/p_counter_collection.csv; python3 tools/pmc_summary.py --last 4 --only rollout_kernel $d/pmc*/p_counter_collection.csv; } > profiles/${tag}_${sh}_pmc_summary.txt
  grep -v amdgpu.ids $d/bench_under_rocprof.json | tail -1 > profiles/${tag}_${sh}_bench_under_rocprof.json
  bi=$(python3 -c "import json; print(json.load(open('profiles/${tag}_${sh}_bench_under_rocprof.json'))['config']['burn_in_steps'])")   # what the run used
  k1=$(python3 -c "import json; print(json.load(open('profiles/${tag}_${sh}_bench_under_rocprof.json'))['roofline']['kernel'])")   # tick_kernel or step_kernel
  python3 tools/make_traffic_json.py $tag --shape $sh --key "${KEY[$sh]}" --kernel $k1 --burn-in $bi > /dev/null
  k2=$(python3 -c "import json; e=json.load(open('profiles/${tag}_${sh}_bench_under_rocprof.json'))['extra']; print([v['kernel'] for k, v in e.items() if 'resident rollout' in k][0])")   # the resident launch of the same sim
  python3 tools/make_traffic_json.py $tag --shape $sh --key "${KEY[$sh]}" --kernel $k2 --ticks-per-launch 64 --burn-in $bi > /dev/null
  if [ $sh = lab ]; then
    for c in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/reset$i -o p -- python3 bench.py $a --burn-in 0 > /dev/null 2> $d/reset$i.err
    done
    python3 tools/pmc_summary.py --only tick_kernel,step_kernel $d/reset*/p_counter_collection.csv > profiles/${tag}_pmc_from_reset.txt
    cp profiles/${tag}_lab_kernel_stats.csv profiles/${tag}_final_kernel_stats.csv
    cp profiles/${tag}_lab_pmc_summary.txt profiles/${tag}_final_pmc_summary.txt
    python3 tools/make_traffic_json.py $tag --kernel $k1 --burn-in $bi > /dev/null
    cp profiles/${tag}_lab_bench_under_rocprof.json profiles/${tag}_final_bench_under_rocprof.json
  fi
  echo "$sh done" >> $out/progress.txt
done
mkdir -p $out/profiles; cp profiles/${tag}_* $out/profiles/     # gpurun merges gpurun_out/ back: copy these into profiles/ and commit
cat profiles/${tag}_*_kernel_stats.csv
