#!/bin/bash
# Diagnostic builds of cat_sim.hip for A/B runs (tools/ab_kernel.sh, tools/ab_parity.sh): build/var/NAME.so from extra -D flags.
# Prints the register / spill figures of tick_kernel<WithFan<FixDims<3,64,2>, F>> as the compiler reports them.
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
for fan in 1 0; do
  grep -A12 "Function Name: .*tick_kernelINS_7WithFanINS_7FixDimsILi3ELi64ELi2EEELi${fan}E" build/var/$name.log | grep -E "VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* *//' | tr '\n' ';'; echo " <- $name fan form $fan (1 = groups, 0 = chunks)"
done
