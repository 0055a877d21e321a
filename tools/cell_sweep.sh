#!/bin/bash
# GPU box: spatial-hash cell size (CAT_GRID_CELL) against tick time and HBM traffic per launch.  usage: tools/cell_sweep.sh OUT.log [map] CELL...
out=$1; map=$2; shift 2; : > $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cell in "$@"; do
  export CAT_GRID_CELL=$cell
  t=$(timeout -k 10 200 python bench.py --map $map --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_line.py "cell $cell")
  d=gpurun_out/cellsweep_$cell; rm -rf $d; mkdir -p $d
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/$c -o p -- python3 bench.py --map $map --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $d/$c.err
  done
  s=$(python3 tools/pmc_summary.py $d/*/p_counter_collection.csv | awk '/tick_kernel/{f=1} f&&/FETCH_SIZE/{fs=$3} f&&/WRITE_SIZE/{ws=$3} END{printf "FETCH %.0f KB WRITE %.0f KB -> %.2f MB per launch", fs, ws, (2*fs+ws)/1024}')
  echo "$t | $s" >> $out
  rm -rf $d
done
cat $out
