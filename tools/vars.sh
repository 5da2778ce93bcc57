#!/bin/bash
# dev: A/B the default library against variant builds in nnop.jl_amd/lib_var{1..N} (interleaved, 2 rounds)
cd /root/repo
CFG="bf16:64:4096:4:4:4:0 f16:64:4096:4:4:4:0 bf16:64:16384:4:4:1:0 bf16:32:4096:8:8:4:0"
N=${1:-1}
for r in 1 2; do
  echo "base : $(python tools/perf.py $CFG 2>/dev/null | cut -c30-52 | tr '\n' ' ')"
  for i in $(seq 1 $N); do
    echo "var $i: $(NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_var$i/libnnop_hip.so python tools/perf.py $CFG 2>/dev/null | cut -c30-52 | tr '\n' ' ')"
  done
done
