#!/bin/bash
# register / spill figures of every kernel instantiation in a build log of tools/build_variant.sh: tools/regs.sh NAME [pattern]
log=build/var/$1.log; pat=${2:-kernel}
grep -A14 "Function Name: .*$pat" $log | grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//; s/Function Name: _ZN12_GLOBAL__N_1[0-9]*//; s/EvPKNS_6ParamsENS_10LaunchArgsE//' | paste - - - - -
