#!/bin/bash
# GPU box: a diagnostic build of the ray fan against the committed kernel: parity, then A/B on the dense workloads.
# usage: tools/ab_big.sh OUTDIR NAME...   (build/var/NAME.so; build/var/base.so is the reference point)
out=$1; shift; mkdir -p $out
libs=()
for v in "$@"; do
  tools/ab_parity.sh build/var/$v.so > $out/parity_$v.txt 2>&1 || { cat $out/parity_$v.txt; exit 1; }
  cat $out/parity_$v.txt
  libs+=(build/var/$v.so)
done
tools/ab_agh.sh $out/ab.txt build/var/base.so "${libs[@]}" > /dev/null || exit 1
cat $out/ab.txt
