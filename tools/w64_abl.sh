#!/bin/bash
# dev: release library, then the timing-only ablation builds of the 64-row forward (make DEV=1 VAR=-DNNOP_W64_ABL=mask
# OUTDIR=../lib_var<n> BUILD=../build_var<n>); w64_check.py prints old-form and w64 times per shape
cd "$(dirname "$0")/.."
CFG=${CFG:-"bf16:64:4096:4:4:4:plain bf16:128:4096:16:16:4:plain bf16:128:8192:32:32:2:causal"}
echo "== release"; python tools/w64_check.py $CFG 2>/dev/null
for a in "$@"; do
  echo "== variant $a"; NNOP_LIB_PATH=$PWD/nnop.jl_amd/lib_var$a/libnnop_hip.so python tools/w64_check.py $CFG 2>/dev/null
done
