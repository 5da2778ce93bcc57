#!/bin/bash
# timing of the ablated builds (results are wrong by construction; timing only)
cd /root/repo
CFG="bf16:64:16384:4:4:1:0"
echo "base  : $(python tools/perf.py $CFG 2>/dev/null | tail -1 | cut -c1-62)"
for a in 1 2 3 4 5 6 7 8; do
  echo "abl $a : $(NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_abl$a/libnnop_hip.so python tools/perf.py $CFG 2>/dev/null | tail -1 | cut -c1-62)"
done
