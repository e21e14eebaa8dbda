#!/bin/bash
# Registers / scratch of every gemm_bf16x3.hip kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
cd "$(dirname "$0")/../multishiftseg_amd/csrc" || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c gemm_bf16x3.hip -o /tmp/split_regs.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/gemm_bf16x3.hip:[0-9]*:[0-9]*: //g; s/remark: //g; s/ \[-Rpass-analysis=kernel-resource-usage\]//g' | paste - - - - | sed 's/Function Name: _ZN12_GLOBAL__N_1//' | cut -c1-200
