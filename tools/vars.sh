#!/bin/bash
cd /root/repo
CFG="bf16:64:4096:4:4:4:0 bf16:64:16384:4:4:1:0"
echo "base : $(python tools/perf.py $CFG 2>/dev/null | tail -2 | cut -c30-52 | tr '\n' ' ')"
for i in 1 2 3 4 5; do
  echo "var $i: $(NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_var$i/libnnop_hip.so python tools/perf.py $CFG 2>/dev/null | tail -2 | cut -c30-52 | tr '\n' ' ')"
done
