#!/bin/bash
cd "$(dirname "$0")/.."
CFG=${CFG:-"bf16:64:4096:4:4:4 bf16:128:4096:16:16:4"}
echo "== release"; python tools/w64_diag.py $CFG 2>/dev/null | grep -v "by "
for a in "$@"; do echo "== variant $a"; NNOP_LIB_PATH=$PWD/nnop.jl_amd/lib_var$a/libnnop_hip.so python tools/w64_diag.py $CFG 2>/dev/null | grep -v "by "; done
