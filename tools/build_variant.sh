#!/bin/bash
# Diagnostic builds of cat_sim.hip for A/B runs (tools/ab_kernel.sh, tools/ab_parity.sh): build/var/NAME.so from extra -D flags, the compiler's
# resource report in build/var/NAME.log (tools/regs.sh NAME prints the register / spill figures of every kernel instantiation).
# usage: tools/build_variant.sh NAME [-DFLAG ...] [--src FILE]
set -e
name=$1; shift
src=as_cops_and_thieves_amd/csrc/cat_sim.hip
flags=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--src" ]; then src=$2; shift 2; else flags+=("$1"); shift; fi
done
mkdir -p build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude "${flags[@]}" -Rpass-analysis=kernel-resource-usage \
  -o build/var/$name.so $src 2> build/var/$name.log || { tail -30 build/var/$name.log; exit 1; }
$(dirname $0)/regs.sh $name 'step_kernel.*FixDimsILi3ELi64ELi2' | sed 's/ScratchSize : /scr /; s/SGPRs Spill: /sS /; s/VGPRs Spill: /vS /; s/VGPRs: /V /' | tr -s '\t ' ' '
