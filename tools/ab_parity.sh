#!/bin/bash
# GPU box: the bit-exact parity files against a diagnostic build of cat_sim.hip.  usage: tools/ab_parity.sh LIB...
for lib in "$@"; do
  echo "== $lib"
  CAT_SIM_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -x -q 2>&1 | tail -3 || exit 1
done
