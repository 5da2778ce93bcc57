#!/bin/bash
# usage: tools/regs.sh fa_fwd_bf16.hip  -> per-kernel VGPR / scratch / spill summary
cd /root/repo/nnop.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -c $1 -o /tmp/build/regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|LDS Size" | paste - - - - - | sed -E 's/.*Function Name: _ZN4nnop[0-9]*([a-z_]*)I([^ ]*)EEvNS[^ ]* .*VGPRs: ([0-9]*).*ScratchSize \[bytes\/lane\]: ([0-9]*).*Spill: ([0-9]*).*/\1 \2 vgpr=\3 scratch=\4 spill=\5/'
