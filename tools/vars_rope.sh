#!/bin/bash
# dev: time the RoPE kernel variants built into nnop.jl_amd/lib_var{1..6} (U x NT)
cd /root/repo
for i in 1 2 3 4 5 6; do
  echo "var $i:"; NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_var$i/libnnop_hip.so python tools/perf_rope.py 2>/dev/null | cut -c1-140 | grep -o '"shape.*"us": [0-9.]*\|gbps": [0-9.]*' | paste - - 
done
