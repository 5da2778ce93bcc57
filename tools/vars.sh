#!/bin/bash
# dev: A/B the default library against variant builds in nnop.jl_amd/lib_var{1..N} (interleaved, 2 rounds); fwd and bwd µs
cd /root/repo
CFG=${CFG:-"bf16:64:4096:4:4:4:0 bf16:64:16384:4:4:1:0 bf16:128:8192:8:8:2:1"}
N=${1:-1}
fmt() { awk '{printf "%s fwd %s bwd %s | ", $1, $3, $9}' ; }
for r in 1 2; do
  echo "base : $(python tools/perf.py $CFG 2>/dev/null | fmt)"
  for i in $(seq 1 $N); do
    echo "var $i: $(NNOP_LIB_PATH=/root/repo/nnop.jl_amd/lib_var$i/libnnop_hip.so python tools/perf.py $CFG 2>/dev/null | fmt)"
  done
done
