#!/bin/bash
# dev: time RoPE kernel variants built into nnop.jl_amd/lib_var{1..N} (e.g. heads per lane NNOP_ROPE_HG = 1, 2, 4, 8)
cd /root/repo
N=${1:-4}
for r in 1 2; do
for i in $(seq 1 $N); do
  echo "var $i: $(NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_var$i/libnnop_hip.so python tools/perf_rope.py 2>/dev/null | grep -o '"shape": "[a-z0-9-]*"\|"us": [0-9.]*\|"gbps": [0-9.]*' | paste - - - | sed 's/"shape": //; s/"us": //; s/"gbps": //' | tr '\n' '|')"
done
done
