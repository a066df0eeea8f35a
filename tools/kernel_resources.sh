#!/bin/bash
# usage: tools/kernel_resources.sh ess_amd/csrc/<file>.hip [grep-pattern]  -- registers / spills / scratch per kernel (hipcc remarks)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -x hip -c "$1" -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|SGPRs Spill" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - | grep -E "${2:-.}" | \
  sed -E 's/Function Name: _ZN[0-9a-zA-Z_]*conv_bf16_ws_k3s1_kernelILi([0-9])ELi([0-9])ELb([01])ELb([01])E.*N7essconv9ConvKArgsE/ws<MB=\1,EPI=\2,SRCBF=\3,OUT8=\4>/'
